from .cache_manager import MetaListPickleIO, CacheManager, MultiCacheManager, build_feature_cache  # noqa: F401
