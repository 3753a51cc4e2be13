"""The registries of the plugin surface (the reference declares these four names in engine/registry/root.py and registers
nothing in them; here every mirrored module registers itself, so ``REGISTRY.get(name)`` is a working lookup).

``REGISTRIES`` maps kind -> Registry for code that wants to walk them (``describe()`` prints what is registered)."""
from .registry import Registry

KINDS = ("backbone", "module", "dataset", "hook")
REGISTRIES = {kind: Registry(kind) for kind in KINDS}

# module-level names the reference's import sites use
globals().update({f"{kind.upper()}_REGISTRY": reg for kind, reg in REGISTRIES.items()})
__all__ = [f"{kind.upper()}_REGISTRY" for kind in KINDS] + ["REGISTRIES", "describe"]


def describe():
    """{kind: sorted names} of everything registered so far."""
    return {kind: sorted(name for name, _ in reg) for kind, reg in REGISTRIES.items()}
