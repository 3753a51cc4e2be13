"""Attribute-access configuration tree with ``_BASE_`` inheritance.

Host-side mirror of the reference's config API (engine/config/config.py:66-611, a yacs fork): the same
class name, ``CfgNode.load_with_base(path)`` (:140-191), attribute get/set (:193-218), ``dump()`` (:244-263),
``merge_from_file/other_cfg/list``, ``freeze/defrost/clone`` -- written from the behaviour, not the source:
a node is a dict whose nested dicts are nodes; ``.py`` config files export a ``cfg`` dict, ``.yaml`` files
hold the same tree; ``_BASE_`` (str or list, paths relative to the including file) is merged first, later
bases overriding earlier ones and the including file overriding all of them.
"""
import ast
import copy
import importlib.util
import os

import yaml

BASE_KEY = "_BASE_"
_LEAF_TYPES = (tuple, list, str, int, float, bool, type(None))


class CfgNode(dict):
    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        init_dict = {} if init_dict is None else init_dict
        key_list = [] if key_list is None else key_list
        tree = {}
        for k, v in copy.deepcopy(dict(init_dict)).items():
            if isinstance(v, dict):
                tree[k] = v if isinstance(v, CfgNode) else CfgNode(v, key_list + [k], new_allowed)
            else:
                if not isinstance(v, _LEAF_TYPES):
                    raise TypeError("Key {} with value {} is not a valid type".format(".".join(key_list + [str(k)]), type(v)))
                tree[k] = v
        super().__init__(tree)
        self.__dict__["_frozen"] = False
        self.__dict__["_new_allowed"] = new_allowed

    # ------------------------------------------------------------------ attribute protocol
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__["_frozen"]:
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        if name in self.__dict__:
            raise AttributeError("Invalid attempt to modify internal CfgNode state: {}".format(name))
        if isinstance(value, dict) and not isinstance(value, CfgNode):
            value = CfgNode(value)
        if not isinstance(value, _LEAF_TYPES + (CfgNode,)):
            raise TypeError("Invalid type {} for key {}".format(type(value), name))
        self[name] = value

    def __str__(self):
        lines = []
        for k, v in sorted(self.items()):
            if isinstance(v, CfgNode):
                body = str(v).split("\n")
                lines.append("{}:\n{}".format(k, "\n".join("  " + b for b in body)))
            else:
                lines.append("{}: {}".format(k, v))
        return "\n".join(lines)

    def __repr__(self):
        return "{}({})".format(type(self).__name__, dict.__repr__(self))

    # ------------------------------------------------------------------ state
    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return self.__dict__["_frozen"]

    def _set_frozen(self, flag):
        self.__dict__["_frozen"] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def is_new_allowed(self):
        return self.__dict__["_new_allowed"]

    def set_new_allowed(self, flag):
        self.__dict__["_new_allowed"] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v.set_new_allowed(flag)

    def clone(self):
        return copy.deepcopy(self)

    # ------------------------------------------------------------------ (de)serialisation
    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kwargs):
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, tuple):
                return [plain(x) for x in v]
            return v
        return yaml.safe_dump(plain(self.to_dict()), **kwargs)

    @classmethod
    def load_cfg(cls, src):
        """``src``: YAML string, or an open file object of a .yaml/.yml or .py (exporting ``cfg``) file."""
        if isinstance(src, str):
            return cls(yaml.safe_load(src) or {})
        ext = os.path.splitext(src.name)[1]
        if ext in ("", ".yaml", ".yml"):
            return cls(yaml.safe_load(src.read()) or {})
        if ext == ".py":
            spec = importlib.util.spec_from_file_location("ucod_cfg_source", src.name)
            module = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(module)
            if not isinstance(getattr(module, "cfg", None), dict):
                raise ValueError("Python config {} must define a dict named 'cfg'".format(src.name))
            return cls(module.cfg)
        raise ValueError("unsupported config file type: {}".format(src.name))

    def merge_from_file(self, filename):
        with open(filename, "r") as f:
            self.merge_from_other_cfg(self.load_cfg(f))

    def merge_from_other_cfg(self, other):
        _merge(other, self, self, [])

    def merge_from_list(self, cfg_list):
        if len(cfg_list) % 2:
            raise ValueError("Override list has odd length: {}".format(cfg_list))
        for full_key, raw in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            *parents, leaf = full_key.split(".")
            for p in parents:
                if p not in node:
                    raise KeyError("Non-existent key: {}".format(full_key))
                node = node[p]
            if leaf not in node:
                raise KeyError("Non-existent key: {}".format(full_key))
            node[leaf] = _coerce(_decode(raw), node[leaf], full_key)

    @classmethod
    def load_with_base(cls, filename, new_allowed=True):
        """Load ``filename`` and everything it inherits through ``_BASE_``; returns a plain nested tree (CfgNode)."""
        cfg = cls(new_allowed=new_allowed)
        cfg.merge_from_file(filename)

        def overlay(src, dst):
            for k, v in src.items():
                if isinstance(v, dict) and k in dst:
                    if not isinstance(dst[k], dict):
                        raise TypeError("Cannot inherit key '{}' from base!".format(k))
                    overlay(v, dst[k])
                else:
                    dst[k] = v

        if BASE_KEY not in cfg:
            return cfg
        bases = cfg[BASE_KEY]
        del cfg[BASE_KEY]
        merged = {}
        for b in (bases if isinstance(bases, (list, tuple)) else [bases]):
            b = os.path.expanduser(b) if b.startswith("~") else b
            if not b.startswith(("/", "http://", "https://")):
                b = os.path.join(os.path.dirname(filename), b)
            overlay(cls.load_with_base(b, new_allowed=new_allowed), merged)
        overlay(cfg, merged)
        return cls(merged, new_allowed=new_allowed)


def _decode(value):
    if isinstance(value, dict):
        return CfgNode(value)
    if not isinstance(value, str):
        return value
    try:
        return ast.literal_eval(value)
    except (ValueError, SyntaxError):
        return value


def _coerce(new, old, full_key):
    """Type-checked replacement: same type, or a safe list<->tuple / int->float cast."""
    if old is None or new is None or type(new) is type(old):
        return new
    for a, b in ((list, tuple), (tuple, list), (int, float)):
        if type(new) is a and type(old) is b:
            return b(new)
    raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(old), type(new), full_key))


def _merge(a, b, root, key_list):
    for k, raw in a.items():
        full_key = ".".join(key_list + [k])
        v = _decode(copy.deepcopy(raw))
        if k in b:
            if isinstance(v, CfgNode) and isinstance(b[k], CfgNode):
                _merge(v, b[k], root, key_list + [k])
            else:
                b[k] = v if isinstance(v, CfgNode) or isinstance(b[k], CfgNode) else _coerce(v, b[k], full_key)
        elif b.is_new_allowed() or root.is_new_allowed():
            b[k] = v
        else:
            raise KeyError("Non-existent config key: {}".format(full_key))
