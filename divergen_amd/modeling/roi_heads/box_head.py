"""FastRCNNConvFCHead (NUM_CONV 0, NUM_FC 2 in the shipped configs).  Mirrors
D2/modeling/roi_heads/box_head.py:26-98 (module names fc1, fc2)."""
import numpy as np
import torch
from torch import nn

from .. import ROI_BOX_HEAD_REGISTRY, ShapeSpec
from ...config import configurable
from ...layers.linear_ops import Linear
from ..backbone.fpn import c2_xavier_fill


@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Sequential):
    @configurable
    def __init__(self, input_shape, *, conv_dims, fc_dims, conv_norm=""):
        super().__init__()
        assert len(conv_dims) == 0, "ROI_BOX_HEAD.NUM_CONV is 0 in every shipped config"
        assert len(fc_dims) > 0
        self._output_size = (input_shape.channels, input_shape.height, input_shape.width)
        self.fcs = []
        for k, fc_dim in enumerate(fc_dims):
            if k == 0:
                self.add_module("flatten", nn.Flatten())
            fc = Linear(int(np.prod(self._output_size)), fc_dim)
            self.add_module("fc%d" % (k + 1), fc)
            self.add_module("fc_relu%d" % (k + 1), nn.ReLU())
            self.fcs.append(fc)
            self._output_size = fc_dim
        for l in self.fcs:
            c2_xavier_fill(l)

    @classmethod
    def from_config(cls, cfg, input_shape):
        h = cfg.MODEL.ROI_BOX_HEAD
        return dict(input_shape=input_shape, conv_dims=[h.CONV_DIM] * h.NUM_CONV, fc_dims=[h.FC_DIM] * h.NUM_FC,
                    conv_norm=h.NORM)

    def forward(self, x):
        for layer in self:
            x = layer(x)
        return x

    @property
    def output_shape(self):
        o = self._output_size
        return ShapeSpec(channels=o) if isinstance(o, int) else ShapeSpec(channels=o[0], height=o[1], width=o[2])


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)
