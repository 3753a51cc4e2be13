from .config import CfgNode  # noqa: F401
