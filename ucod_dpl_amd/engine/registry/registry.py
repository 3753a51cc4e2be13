"""name -> object registry: the reference's declared plugin surface (engine/registry/registry.py:10-92;
``register()`` as decorator or call, ``get()``, ``in``, iteration).  The reference never registers anything;
this build registers its modules so ``get(name)`` is a working lookup path."""
from tabulate import tabulate


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _add(self, name, obj):
        if name in self._obj_map:
            raise AssertionError("An object named '{}' was already registered in '{}' registry!".format(name, self._name))
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(target):
                self._add(target.__name__, target)
                return target
            return deco
        self._add(obj.__name__, obj)
        return None

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def __repr__(self):
        return "Registry of {}:\n".format(self._name) + tabulate(self._obj_map.items(), headers=["Names", "Objects"], tablefmt="fancy_grid")

    __str__ = __repr__
