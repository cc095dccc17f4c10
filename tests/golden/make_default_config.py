"""Dump the reference's default config tree (detectron2 defaults + add_centernet_config +
add_divergen_config) as plain data -> divergen_amd/config/defaults.json.  Authoring container only.
The key names and default values are the drop-in surface the shipped YAMLs rely on."""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa

R.install()


class CN(dict):
    def __init__(self, init=None, **kw):
        super().__init__()
        if init:
            for k, v in init.items():
                self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


cfgmod = types.ModuleType("detectron2.config.config")
cfgmod.CfgNode = CN
sys.modules["detectron2.config.config"] = cfgmod
sys.modules["detectron2.config"].CfgNode = CN
sys.modules["detectron2.config"].__path__ = [R.D2 + "/config"]
import importlib  # noqa
d = importlib.import_module("detectron2.config.defaults")
cfg = d._C
cnc = importlib.import_module("centernet.config")
cnc.add_centernet_config(cfg)
dgc = importlib.import_module("divergen.config")
dgc.add_divergen_config(cfg)


def plain(x):
    if isinstance(x, dict):
        return {k: plain(v) for k, v in x.items()}
    if isinstance(x, (tuple, list)):
        return [plain(v) for v in x]
    return x


out = os.path.join(HERE, "..", "..", "divergen_amd", "config", "defaults.json")
json.dump(plain(cfg), open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, len(json.dumps(plain(cfg))))
