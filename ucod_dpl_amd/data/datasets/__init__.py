from .feature_cache import MetaListPickleIO, CacheManager, MultiCacheManager, build_feature_cache  # noqa: F401
