"""FPN + P6/P7 top block.  Mirrors D2/modeling/backbone/fpn.py:17-162 and
CN/modeling/backbone/fpn_p5.py:15-33 (module names fpn_lateral{3,4,5}, fpn_output{3,4,5},
top_block.p6/p7).  Convolutions run channels-last so the RoI kernels read pixels contiguously."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ShapeSpec
from ...layers.conv_ops import Conv2d
from ...layers.norm_ops import upsample2x_add
from .swintransformer import Backbone


def c2_xavier_fill(m):
    nn.init.kaiming_uniform_(m.weight, a=1)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


def c2_msra_fill(m):
    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


class LastLevelP6P7_P5(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.num_levels, self.in_feature = 2, "p5"
        self.p6 = Conv2d(in_channels, out_channels, 3, 2, 1)
        self.p7 = Conv2d(out_channels, out_channels, 3, 2, 1)
        for m in (self.p6, self.p7):
            c2_xavier_fill(m)

    def forward(self, c5):
        p6 = self.p6(c5)
        return [p6, self.p7(F.relu(p6))]


class FPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        assert norm == "", "only the shipped configuration (FPN.NORM '') is built"
        shapes = bottom_up.output_shape()
        strides = [shapes[f].stride for f in in_features]
        chans = [shapes[f].channels for f in in_features]
        lateral, output = [], []
        for s, c in zip(strides, chans):
            stage = int(math.log2(s))
            lc = Conv2d(c, out_channels, 1)
            oc = Conv2d(out_channels, out_channels, 3, 1, 1)
            c2_xavier_fill(lc)
            c2_xavier_fill(oc)
            self.add_module("fpn_lateral%d" % stage, lc)
            self.add_module("fpn_output%d" % stage, oc)
            lateral.append(lc)
            output.append(oc)
        self.lateral_convs, self.output_convs = lateral[::-1], output[::-1]
        self.top_block, self.in_features, self.bottom_up = top_block, tuple(in_features), bottom_up
        self._out_feature_strides = {"p%d" % int(math.log2(s)): s for s in strides}
        if top_block is not None:
            for s in range(stage, stage + top_block.num_levels):
                self._out_feature_strides["p%d" % (s + 1)] = 2 ** (s + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]
        self._fuse_type = fuse_type

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def forward(self, x):
        feats = self.bottom_up(x)
        seg = self.__dict__.get("_segment")
        if seg is None:
            from ...utils.graphs import GraphedSegment
            seg = self.__dict__["_segment"] = GraphedSegment(_TopDown(self))
        ins = tuple(feats[f] for f in self.in_features)
        if self.training and seg.usable(ins):          # static shapes: replay the captured forward / backward graphs
            return dict(zip(self._out_features, seg(*ins)))
        return self._top_down(feats)

    def _top_down(self, feats):
        results = []
        prev = self.lateral_convs[0](feats[self.in_features[-1]])
        results.append(self.output_convs[0](prev))
        for idx, (lc, oc) in enumerate(zip(self.lateral_convs, self.output_convs)):
            if idx > 0:
                prev = upsample2x_add(lc(feats[self.in_features[-idx - 1]]), prev)       # lateral + nearest 2x of the level above, one launch
                if self._fuse_type == "avg":
                    prev = prev / 2
                results.insert(0, oc(prev))
        if self.top_block is not None:
            src = feats[self.top_block.in_feature] if self.top_block.in_feature in feats else \
                results[self._out_features.index(self.top_block.in_feature)]
            results.extend(self.top_block(src))
        return dict(zip(self._out_features, results))


class _TopDown(nn.Module):
    """FPN minus its bottom-up network, tensors in / tuple out: the unit captured as a hipGraph."""

    def __init__(self, fpn):
        super().__init__()
        self.__dict__["fpn"] = fpn          # not a registered child: the FPN owns the parameters
        self.amp = False

    def parameters(self, recurse=True):
        return (p for n, p in self.__dict__["fpn"].named_parameters() if not n.startswith("bottom_up."))

    def forward(self, *ins):
        fpn = self.__dict__["fpn"]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp, cache_enabled=False):
            out = fpn._top_down(dict(zip(fpn.in_features, ins)))
        return tuple(out[k] for k in fpn._out_features)
