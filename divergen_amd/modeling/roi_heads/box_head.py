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
        # The first FC reads the pooled (R, C, S, S) features flattened.  The pooler's output is channels-last in memory, so the
        # parameter is STORED with its columns in (h, w, c) order: the flattening is then a view (the reference's (c, h, w) order cost
        # a transposing copy of the (1024, 256, 7, 7) tensor per cascade stage and pass).  State dicts keep the reference's column
        # order: the two hooks below permute on the way out and in, so checkpoints and goldens load unchanged.
        self._chw = None
        if input_shape.height and input_shape.width:
            self._chw = (int(input_shape.channels), int(input_shape.height), int(input_shape.width))
            self.fcs[0].weight._dgx_sd_perm = (self._cols_to_chw, self._cols_to_hwc)
            self._register_state_dict_hook(self._sd_out)
            self._register_load_state_dict_pre_hook(self._sd_in)

    def _cols_to_chw(self, w):          # stored (O, h*w*c) -> the reference's (O, c*h*w)
        c, h, w_ = self._chw
        return w.reshape(w.shape[0], h, w_, c).permute(0, 3, 1, 2).reshape(w.shape[0], -1).contiguous()

    def _cols_to_hwc(self, w):
        c, h, w_ = self._chw
        return w.reshape(w.shape[0], c, h, w_).permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()

    @staticmethod
    def _sd_out(module, state_dict, prefix, local_metadata):
        k = prefix + "fc1.weight"
        if k in state_dict:
            state_dict[k] = module._cols_to_chw(state_dict[k])

    def _sd_in(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        k = prefix + "fc1.weight"
        if k in state_dict and tuple(state_dict[k].shape) == tuple(self.fcs[0].weight.shape):
            state_dict[k] = self._cols_to_hwc(state_dict[k])

    def flatten_rows(self, x):
        """(R, C, S, S) logical -> (R, S*S*C): a view when the storage is channels-last (the pooler's output)."""
        return x.permute(0, 2, 3, 1).reshape(x.shape[0], -1) if (x.dim() == 4 and self._chw is not None) else x.flatten(1)

    @classmethod
    def from_config(cls, cfg, input_shape):
        h = cfg.MODEL.ROI_BOX_HEAD
        return dict(input_shape=input_shape, conv_dims=[h.CONV_DIM] * h.NUM_CONV, fc_dims=[h.FC_DIM] * h.NUM_FC,
                    conv_norm=h.NORM)

    def forward(self, x):
        for layer in self:
            x = self.flatten_rows(x) if isinstance(layer, nn.Flatten) else layer(x)
        return x

    @property
    def output_shape(self):
        o = self._output_size
        return ShapeSpec(channels=o) if isinstance(o, int) else ShapeSpec(channels=o[0], height=o[1], width=o[2])


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)
