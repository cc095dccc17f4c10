"""Load individual source files of the read-only reference (/root/reference) for
golden-vector generation.  AUTHORING CONTAINER ONLY: nothing here runs on the GPU box
and nothing from the reference is copied into the repo -- only inputs/outputs of the
reference functions are saved (tests/golden/*.npz) by make_golden.py.

The reference stack (detectron2, fvcore, timm, torchvision, cv2, ...) is not
installed, so the leaf files are imported one by one with:
  * bare namespace packages (``__path__`` pointing at the real directory) so that the
    heavy ``__init__`` files never execute;
  * permissive stub modules for third-party roots that are absent.
"""
import importlib
import importlib.abc
import importlib.machinery
import math
import sys
import types
from collections import namedtuple
from contextlib import contextmanager

import torch
from torch import nn

REF = "/root/reference"
D2 = REF + "/BSGAL/third_party/CenterNet2/detectron2"
CN = REF + "/BSGAL/third_party/CenterNet2/projects/CenterNet2/centernet"
DG = REF + "/DiverGen/divergen"


class _Permissive(types.ModuleType):
    """Module whose every attribute is another permissive object (never called for math)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        val = _PermissiveObj(self.__name__ + "." + name)
        setattr(self, name, val)
        return val


class _PermissiveObj:
    def __init__(self, name="stub"):
        self._name = name

    def __call__(self, *a, **k):
        # used as decorator -> return the decorated function unchanged
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _PermissiveObj(self._name + "()")

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _PermissiveObj(self._name + "." + name)

    def __mro_entries__(self, bases):
        return (object,)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("cv2", "albumentations", "fvcore", "torchvision", "pycocotools", "lvis",
             "iopath", "yacs", "timm", "termcolor", "clip")

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Permissive(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _ns(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    return m


class _Registry(dict):
    def register(self, obj=None):
        def deco(o):
            self[o.__name__] = o
            return o
        return deco(obj) if obj is not None else deco

    def get(self, name):
        return self[name]


def _configurable(init_func=None, *, from_config=None):
    """Stand-in for detectron2.config.configurable: explicit-kwargs construction only."""
    if init_func is not None:
        return init_func
    return lambda f: f


class _Storage:
    def __init__(self):
        self.scalars = {}

    def put_scalar(self, k, v, **kw):
        self.scalars[k] = float(v)

    @contextmanager
    def name_scope(self, name):
        yield

    iter = 0


_STORAGE = _Storage()


def drop_path(x, drop_prob: float = 0., training: bool = False):
    # timm==0.4.9 timm/models/layers/drop.py (published algorithm; timm is not vendored)
    if drop_prob == 0. or not training:
        return x
    keep_prob = 1 - drop_prob
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    random_tensor = keep_prob + torch.rand(shape, dtype=x.dtype, device=x.device)
    random_tensor.floor_()
    return x.div(keep_prob) * random_tensor


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)


def install():
    """Install stubs + namespace packages.  Idempotent."""
    if getattr(install, "_done", False):
        return
    install._done = True
    sys.meta_path.insert(0, _StubFinder())

    # ---- timm.models.layers (real arithmetic needed: DropPath / trunc_normal_) ----
    tl = _Permissive("timm.models.layers")
    tl.__path__ = []
    tl.DropPath = DropPath
    tl.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    t = _Permissive("timm"); t.__path__ = []
    tm = _Permissive("timm.models"); tm.__path__ = []
    sys.modules.update({"timm": t, "timm.models": tm, "timm.models.layers": tl})

    # ---- fvcore.nn (smooth_l1 / giou / weight_init) ----
    fv = _Permissive("fvcore"); fv.__path__ = []
    fnn = _Permissive("fvcore.nn"); fnn.__path__ = []

    def smooth_l1_loss(input, target, beta, reduction="none"):
        # fvcore.nn.smooth_l1_loss published semantics; beta < 1e-5 => L1
        if beta < 1e-5:
            loss = torch.abs(input - target)
        else:
            n = torch.abs(input - target)
            loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
        if reduction == "mean":
            loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
        elif reduction == "sum":
            loss = loss.sum()
        return loss

    fnn.smooth_l1_loss = smooth_l1_loss
    wi = types.ModuleType("fvcore.nn.weight_init")

    def c2_xavier_fill(module):
        nn.init.kaiming_uniform_(module.weight, a=1)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)

    def c2_msra_fill(module):
        nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)

    wi.c2_xavier_fill = c2_xavier_fill
    wi.c2_msra_fill = c2_msra_fill
    fnn.weight_init = wi
    sys.modules.update({"fvcore": fv, "fvcore.nn": fnn, "fvcore.nn.weight_init": wi})

    # ---- detectron2 namespace (real leaf files, stubbed packages) ----
    d2 = _ns("detectron2", D2)
    _ns("detectron2.structures", D2 + "/structures")
    boxes = importlib.import_module("detectron2.structures.boxes")
    inst = importlib.import_module("detectron2.structures.instances")
    st = sys.modules["detectron2.structures"]
    st.Boxes, st.pairwise_iou, st.BoxMode = boxes.Boxes, boxes.pairwise_iou, boxes.BoxMode
    st.pairwise_ioa = boxes.pairwise_ioa
    st.Instances = inst.Instances
    for nm in ("ImageList", "BitMasks", "ROIMasks", "PolygonMasks"):
        setattr(st, nm, _PermissiveObj(nm))

    lay = _ns("detectron2.layers", D2 + "/layers")
    wr = importlib.import_module("detectron2.layers.wrappers")
    ss = importlib.import_module("detectron2.layers.shape_spec")
    lay.ShapeSpec = ss.ShapeSpec
    for nm in ("cat", "nonzero_tuple", "Conv2d", "ConvTranspose2d", "cross_entropy", "shapes_to_tensor"):
        setattr(lay, nm, getattr(wr, nm))
    lay.batched_nms = _PermissiveObj("batched_nms")
    lay.get_norm = lambda norm, ch: None if norm == "" else nn.GroupNorm(32, ch)
    lay.ROIAlign = _PermissiveObj("ROIAlign")
    lay.ROIAlignRotated = _PermissiveObj("ROIAlignRotated")
    for nm in ("Linear", "interpolate", "FrozenBatchNorm2d", "CNNBlockBase"):
        setattr(lay, nm, getattr(wr, nm, _PermissiveObj(nm)))
    for nm in ("ciou_loss", "diou_loss"):
        setattr(lay, nm, _PermissiveObj(nm))

    cfgm = _ns("detectron2.config")
    cfgm.configurable = _configurable
    cfgm.CfgNode = dict
    ut = _ns("detectron2.utils")
    comm = _ns("detectron2.utils.comm")
    comm.get_world_size = lambda: 1
    comm.get_rank = lambda: 0
    comm.is_main_process = lambda: True
    ev = _ns("detectron2.utils.events")
    ev.get_event_storage = lambda: _STORAGE
    reg = _ns("detectron2.utils.registry")
    reg.Registry = lambda name: _Registry()
    for nm in ("visualizer", "memory", "env", "logger", "file_io"):
        sys.modules["detectron2.utils." + nm] = _Permissive("detectron2.utils." + nm)
    sys.modules["detectron2.utils.memory"].retry_if_cuda_oom = lambda f: f

    mod = _ns("detectron2.modeling", D2 + "/modeling")
    mod.detector_postprocess = _PermissiveObj("detector_postprocess")
    bb = _ns("detectron2.modeling.backbone", D2 + "/modeling/backbone")
    bbb = _ns("detectron2.modeling.backbone.backbone")

    class Backbone(nn.Module):
        def output_shape(self):
            return {n: ss.ShapeSpec(channels=self._out_feature_channels[n],
                                    stride=self._out_feature_strides[n])
                    for n in self._out_features}

        @property
        def size_divisibility(self):
            return 0

    bbb.Backbone = Backbone
    bb.Backbone = Backbone
    bld = _ns("detectron2.modeling.backbone.build")
    bld.BACKBONE_REGISTRY = _Registry()
    rn = _ns("detectron2.modeling.backbone.resnet")
    rn.build_resnet_backbone = _PermissiveObj("build_resnet_backbone")
    importlib.import_module("detectron2.modeling.backbone.fpn")  # real FPN
    bb.FPN = sys.modules["detectron2.modeling.backbone.fpn"].FPN
    pg = _ns("detectron2.modeling.proposal_generator", D2 + "/modeling/proposal_generator")
    pgb = _ns("detectron2.modeling.proposal_generator.build")
    pgb.PROPOSAL_GENERATOR_REGISTRY = _Registry()
    ma = _ns("detectron2.modeling.meta_arch")
    mab = _ns("detectron2.modeling.meta_arch.build")
    mab.META_ARCH_REGISTRY = _Registry()
    # real torch-only leaves
    for leaf in ("matcher", "sampling", "box_regression"):
        importlib.import_module("detectron2.modeling." + leaf)
    mod.Box2BoxTransform = sys.modules["detectron2.modeling.box_regression"].Box2BoxTransform
    _ns("detectron2.modeling.roi_heads", D2 + "/modeling/roi_heads")
    importlib.import_module("detectron2.modeling.roi_heads.fast_rcnn")
    _ns("detectron2.solver", D2 + "/solver")
    _ns("detectron2.data")
    sys.modules["detectron2.data.detection_utils"] = _Permissive("detectron2.data.detection_utils")
    sys.modules["detectron2.data.transforms"] = _Permissive("detectron2.data.transforms")
    sys.modules["detectron2.data.transforms"].Augmentation = object
    sys.modules["detectron2.data.transforms"].Transform = object
    for nm in ("samplers", "datasets", "datasets.coco", "dataset_mapper", "build", "common", "catalog"):
        sys.modules["detectron2.data." + nm] = _Permissive("detectron2.data." + nm)
    sys.modules["detectron2.data.dataset_mapper"].DatasetMapper = object

    # ---- centernet namespace ----
    _ns("centernet", CN)
    _ns("centernet.modeling", CN + "/modeling")
    _ns("centernet.modeling.layers", CN + "/modeling/layers")
    _ns("centernet.modeling.dense_heads", CN + "/modeling/dense_heads")
    _ns("centernet.modeling.backbone", CN + "/modeling/backbone")
    dbg = _Permissive("centernet.modeling.debug")
    sys.modules["centernet.modeling.debug"] = dbg
    dc = _Permissive("centernet.modeling.layers.deform_conv")
    sys.modules["centernet.modeling.layers.deform_conv"] = dc
    bif = _Permissive("centernet.modeling.backbone.bifpn")
    sys.modules["centernet.modeling.backbone.bifpn"] = bif

    # ---- divergen namespace ----
    _ns("divergen", DG)
    _ns("divergen.modeling", DG + "/modeling")
    _ns("divergen.modeling.backbone", DG + "/modeling/backbone")
    _ns("divergen.modeling.roi_heads", DG + "/modeling/roi_heads")
    _ns("divergen.data", DG + "/data")
    _ns("divergen.data.transforms", DG + "/data/transforms")
    sys.modules["divergen.modeling.debug"] = _Permissive("divergen.modeling.debug")
    sys.modules["divergen.data.transforms.possion_blending"] = _Permissive("pb")
    for nm in ("custom_augmentation_impl", "custom_copypaste", "custom_color_jitter"):
        sys.modules["divergen.data.transforms." + nm] = _Permissive(nm)


def ref(modname):
    install()
    return importlib.import_module(modname)
