"""Name -> class registries and the `configurable` constructor convention.

[D2-upstream] detectron2.utils.registry.Registry and detectron2.config.configurable: the
reference selects its ROI heads / box predictor by yaml strings through these
(ovr/modeling/roi_heads/roi_emb_heads.py:121,130,309; box_emb_head.py:68,239-249).  If
Detectron2 is importable, the classes of this package are ALSO registered into its
ROI_HEADS_REGISTRY (see locov_amd/roi_heads/__init__.py), which is what makes them a drop-in
for train_ovnet.py.
"""
from __future__ import annotations

import functools
import inspect


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._do_register(o.__name__, o)
                return o
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def _do_register(self, name, obj):
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map


def _is_cfg(x) -> bool:
    return hasattr(x, "MODEL") or (isinstance(x, dict) and "MODEL" in x)


def configurable(init_func):
    """Lets `Cls(cfg, *args)` expand to `Cls(**Cls.from_config(cfg, *args))`; explicit keyword
    construction keeps working."""
    assert inspect.isfunction(init_func) and init_func.__name__ == "__init__"

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        if (args and _is_cfg(args[0])) or _is_cfg(kwargs.get("cfg")):
            explicit = type(self).from_config(*args, **kwargs)
            init_func(self, **explicit)
        else:
            init_func(self, *args, **kwargs)

    return wrapped
