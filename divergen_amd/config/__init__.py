"""Config surface of the reference: a yacs-style CfgNode tree with `_BASE_` YAML inheritance,
`KEY VALUE` list overrides and `configurable`/`from_config` construction.

Mirrors D2/config/{config.py,defaults.py} + CN/config.py:3-88 (add_centernet_config) +
DG/divergen/config.py:4-209 (add_divergen_config).  The default VALUES are data
(divergen_amd/config/defaults.json, dumped from the reference's config tree by
tests/golden/make_default_config.py); DiverGen's configs/*.yaml load unchanged.
"""
import ast
import copy
import functools
import inspect
import json
import os

import yaml

_DEFAULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "defaults.json")
BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError("Attempted to set %s on a frozen CfgNode" % k)
        self[k] = v

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return object.__getattribute__(self, "_frozen")

    def _set_frozen(self, f):
        object.__setattr__(self, "_frozen", f)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(f)

    def clone(self):
        c = CfgNode(copy.deepcopy(_plain(self)))
        return c

    def dump(self, **kw):
        return yaml.safe_dump(_plain(self), **kw)

    # ---- merging
    @staticmethod
    def load_yaml_with_base(filename):
        with open(filename) as f:
            cfg = yaml.safe_load(f) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            merged = CfgNode.load_yaml_with_base(base)
            _merge_dict(cfg, merged, strict=False)
            return merged
        return cfg

    def merge_from_file(self, filename, allow_unsafe=False):
        loaded = CfgNode.load_yaml_with_base(filename)
        loaded.pop("VERSION", None) if "VERSION" not in self else None
        self.merge_from_other_cfg(loaded)

    def merge_from_other_cfg(self, other):
        _merge_node(other, self, [])

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0, "Override list has odd length: %s" % (lst,)
        for full_key, v in zip(lst[0::2], lst[1::2]):
            d = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in d:
                    raise KeyError("Non-existent config key: %s" % full_key)
                d = d[p]
            if parts[-1] not in d:
                raise KeyError("Non-existent config key: %s" % full_key)
            d[parts[-1]] = _coerce(_decode(v), d[parts[-1]], full_key)


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, tuple):
        return [_plain(v) for v in x]
    if isinstance(x, list):
        return [_plain(v) for v in x]
    return x


def _decode(v):
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, key):
    if old is None or new is None or type(new) == type(old):
        return new
    if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
        return type(old)(new) if isinstance(old, tuple) else list(new)
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, str) and not isinstance(new, str):
        raise ValueError("Type mismatch for %s: %r vs %r" % (key, new, old))
    if isinstance(old, bool) != isinstance(new, bool) or (isinstance(old, (int, float)) and isinstance(new, str)):
        raise ValueError("Type mismatch for %s: %r (%s) vs %r (%s)" % (key, new, type(new), old, type(old)))
    return new


def _merge_dict(src, dst, strict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_dict(v, dst[k], strict)
        else:
            dst[k] = v


def _merge_node(src, dst, path):
    for k, v in src.items():
        full = ".".join(path + [k])
        if k not in dst:
            raise KeyError("Non-existent config key: %s" % full)
        if isinstance(v, dict):
            if not isinstance(dst[k], dict):
                raise ValueError("Type mismatch for %s" % full)
            _merge_node(v, dst[k], path + [k])
        else:
            dst[k] = _coerce(_decode(v), dst[k], full)


def get_cfg():
    """Defaults of detectron2 + add_centernet_config + add_divergen_config (already merged)."""
    with open(_DEFAULTS) as f:
        cfg = CfgNode(json.load(f))
    # Keys of THIS build only (not in the reference tree; defaults keep the reference behaviour):
    #   INPUT.INST_POOL_SHARDS: directory of pool-*.dgxpool shards (divergen_amd/data/pool_store.py, built by
    #   tools/build_inst_pool.py); when set, pool instances are read from the shards instead of PIL-opened per sample.
    cfg.INPUT.INST_POOL_SHARDS = ""
    #   SOLVER.ALLREDUCE_DTYPE: "fp32" (the reference's DDP: gradients all-reduced as they are) or "bf16" (gradient buckets go over
    #   xGMI as bf16, half the bytes per step; engine/ddp.py ArenaReducer(wire_dtype=...)).
    cfg.SOLVER.ALLREDUCE_DTYPE = "fp32"
    return cfg


def add_centernet_config(cfg):
    """CN/config.py:3-88 -- keys are already part of get_cfg(); kept for call-site compatibility."""
    return cfg


def add_divergen_config(cfg):
    """DG/divergen/config.py:4-209 -- keys are already part of get_cfg()."""
    return cfg


def add_bsgal_config(cfg):
    """BS/bsgal/config.py:52-79,176-180: the keys BSGAL adds on top of the Detic / DiverGen tree (the gradient-based active
    selection of pasted batches); BS/configs/BSGAL/*.yaml load with these in place."""
    m, i = cfg.MODEL, cfg.INPUT
    for k, v in (("ACTIVE_MODE", "paste_or_zero"), ("ACTIVE_LOSS", "cls"), ("ACTIVE_LOSS_UPDATE", "all"), ("ACTIVE_SEED", 0),
                 ("ACTIVE_COMPARE", "default"), ("ACTIVE_TEST", "select"), ("ACTIVE_TEST_INS", "one"), ("ACTIVE_LR", 0.0001),
                 ("ACTIVE_OPTIMIZER", True), ("ACTIVE_OPTIMIZER_MODE", "sgd"), ("ACTIVE_PRED", False), ("ACTIVE_PRED_CHOOSE", ""),
                 ("ACTIVE_PRED_SUP", "all"), ("ACTIVE_ONLY_GT_TRAIN", False), ("ACTIVE_ONLY_GT_TEST", False),
                 ("ACTIVE_GRAD_COMPARE", False), ("ACTIVE_GRAD_NORM", True), ("ACTIVE_GRAD_SAVE", False),
                 ("ACTIVE_GRAD_UPDATE", "AVERAGE"), ("ONLY_PASTE_SUP", False), ("ACTIVE_FORWARD_ONCE", False),
                 ("ACTIVE_ONCE_MODE", "only_gt"), ("ACTIVE_EVAL", False), ("ACTIVE_DYNAMIC_THRES", 0.0), ("ACTIVE_TEST_BATCHSIZE", 4),
                 ("USE_XPASTE_BOX_LOSS", True), ("USE_XPASTE_MASK_LOSS", True)):
        if k not in m:
            m[k] = v
    for k, v in (("ACTIVE_SELECT", False), ("ACTIVE_SELECT_TYPE", "train"), ("SEPARATE_SYN", False), ("SEPERATE_SUP", False)):
        if k not in i:
            i[k] = v
    return cfg


def configurable(init_func=None, *, from_config=None):
    """D2/config/config.py `configurable`: allow `Cls(cfg, *args)` to build through
    `Cls.from_config(cfg, *args)` while still accepting explicit keyword construction."""
    def _called_with_cfg(*args, **kwargs):
        if len(args) and isinstance(args[0], CfgNode):
            return True
        return isinstance(kwargs.get("cfg"), CfgNode)

    if init_func is not None:
        @functools.wraps(init_func)
        def wrapped(self, *args, **kwargs):
            fc = type(self).from_config
            if _called_with_cfg(*args, **kwargs):
                explicit = fc(*args, **kwargs)
                init_func(self, **explicit)
            else:
                init_func(self, *args, **kwargs)
        return wrapped

    def wrapper(orig):
        @functools.wraps(orig)
        def wrapped(*args, **kwargs):
            if _called_with_cfg(*args, **kwargs):
                return orig(**from_config(*args, **kwargs))
            return orig(*args, **kwargs)
        return wrapped
    return wrapper
