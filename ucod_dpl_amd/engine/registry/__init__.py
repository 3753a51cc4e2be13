from .registry import Registry  # noqa: F401
from .root import BACKBONE_REGISTRY, MODULE_REGISTRY, DATASET_REGISTRY, HOOK_REGISTRY  # noqa: F401
