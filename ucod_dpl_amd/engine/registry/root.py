"""The four registries the reference declares (engine/registry/root.py:3-6)."""
from .registry import Registry

BACKBONE_REGISTRY = Registry("backbone")
MODULE_REGISTRY = Registry("module")
DATASET_REGISTRY = Registry("dataset")
HOOK_REGISTRY = Registry("hook")
