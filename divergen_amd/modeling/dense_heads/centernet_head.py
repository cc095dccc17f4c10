"""CenterNet head (shared towers + agn_hm / bbox_pred).  Mirrors
CN/modeling/dense_heads/centernet_head.py:13-162; parameter names (bbox_tower.{0,1,3,4,...},
agn_hm, bbox_pred, scales.{l}.scale) follow the reference so checkpoints load."""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ...config import configurable
from ...layers.norm_ops import groupnorm_relu, groupnorm_relu_multi
from ...layers.conv_ops import Conv2d, conv3x3, conv3x3_group, conv3x3_group_usable, conv3x3_multi
from ...layers.linear_ops import group_parameters
from ...layers.dense_ops import centernet_head_outputs


class Scale(nn.Module):
    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.FloatTensor([init_value]))

    def forward(self, x):
        return x * self.scale


_HEAD_OUT = True      # the head's tail + flattening as one kernel each way
_GN_MULTI = True      # GroupNorm of a tower layer over all levels at once


class CenterNetHead(nn.Module):
    @configurable
    def __init__(self, in_channels, num_levels, *, num_classes=80, with_agn_hm=False, only_proposal=False,
                 norm="GN", num_cls_convs=4, num_box_convs=4, num_share_convs=0, use_deformable=False,
                 prior_prob=0.01):
        super().__init__()
        assert not use_deformable, "USE_DEFORMABLE is False in every shipped config"
        self.num_classes, self.with_agn_hm, self.only_proposal = num_classes, with_agn_hm, only_proposal
        for name, n in (("cls", 0 if only_proposal else num_cls_convs), ("bbox", num_box_convs), ("share", num_share_convs)):
            tower = []
            for _ in range(n):
                tower.append(Conv2d(in_channels, in_channels, 3, 1, 1, bias=True))
                if norm == "GN":
                    tower.append(nn.GroupNorm(32 if in_channels % 32 == 0 else 25, in_channels))
                elif norm != "":
                    raise NotImplementedError(norm)
                tower.append(nn.ReLU())
            self.add_module(name + "_tower", nn.Sequential(*tower))
        self.bbox_pred = Conv2d(in_channels, 4, 3, 1, 1)
        self.scales = nn.ModuleList([Scale(1.0) for _ in range(num_levels)])
        for mod in (self.cls_tower, self.bbox_tower, self.share_tower, self.bbox_pred):
            for l in mod.modules():
                if isinstance(l, nn.Conv2d):
                    nn.init.normal_(l.weight, std=0.01)
                    nn.init.constant_(l.bias, 0)
        nn.init.constant_(self.bbox_pred.bias, 8.0)
        bias_value = -math.log((1 - prior_prob) / prior_prob)
        if with_agn_hm:
            self.agn_hm = Conv2d(in_channels, 1, 3, 1, 1)
            nn.init.constant_(self.agn_hm.bias, bias_value)
            nn.init.normal_(self.agn_hm.weight, std=0.01)
            # heat-map row first, the four regression rows behind it, zero rows up to 64 (one whole K-tile of the input gradient)
            group_parameters(self.agn_hm.weight, self.bbox_pred.weight, pad_to=64)
            group_parameters(self.agn_hm.bias, self.bbox_pred.bias, pad_to=64)
        if not only_proposal:
            self.cls_logits = Conv2d(in_channels, num_classes, 3, 1, 1)
            nn.init.constant_(self.cls_logits.bias, bias_value)
            nn.init.normal_(self.cls_logits.weight, std=0.01)

    @classmethod
    def from_config(cls, cfg, input_shape):
        c = cfg.MODEL.CENTERNET
        return dict(in_channels=[s.channels for s in input_shape][0], num_levels=len(input_shape),
                    num_classes=c.NUM_CLASSES, with_agn_hm=c.WITH_AGN_HM, only_proposal=c.ONLY_PROPOSAL, norm=c.NORM,
                    num_cls_convs=c.NUM_CLS_CONVS, num_box_convs=c.NUM_BOX_CONVS, num_share_convs=c.NUM_SHARE_CONVS,
                    use_deformable=c.USE_DEFORMABLE, prior_prob=c.PRIOR_PROB)

    @staticmethod
    def _run_tower(tower, x):
        """nn.Sequential of (Conv2d, GroupNorm, ReLU) triples; GroupNorm + ReLU go out as one channels-last kernel
        pair when the activation is bf16 (autocast) -- no NCHW<->NHWC copies between the convolutions."""
        mods = list(tower)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.GroupNorm):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = groupnorm_relu(x, m.weight, m.bias, m.num_groups, m.eps, relu=relu)
                i += 2 if relu else 1
            else:
                x = m(x)
                i += 1
        return x

    @staticmethod
    def _run_tower_levels(tower, xs):
        """The tower over ALL levels, layer by layer: the convolutions stay one (implicit-GEMM) call per level, the GroupNorm + ReLU
        of a layer -- shared weights, one small latency-bound launch per level and pass in the level-by-level order -- runs over
        the levels together (layers.norm_ops.groupnorm_relu_multi: one launch per pass instead of five)."""
        mods = list(tower)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.GroupNorm):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                xs = groupnorm_relu_multi(xs, m.weight, m.bias, m.num_groups, m.eps, relu=relu)
                i += 2 if relu else 1
            else:
                ys = conv3x3_multi(xs, m.weight, m.bias) if isinstance(m, Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) else None
                xs = ys if ys is not None else [m(x) for x in xs]
                i += 1
        return xs

    def _multi(self, x):
        return (_GN_MULTI and len(x) > 1 and len(x) <= 8 and all(f.is_cuda for f in x) and len(self.share_tower) == 0 and len(self.cls_tower) == 0
                and all(not isinstance(m, nn.GroupNorm) or self.bbox_tower[0].out_channels == 8 * m.num_groups for m in self.bbox_tower))

    def forward_flat(self, x):
        """-> (reg (M, 4) f32, agn_hm logits (M,) f32), levels stacked in (level, image, y, x) order: the loss operands of
        centernet.py:179-235.  On the layer-major path the tail (slices, scale, ReLU, permutes, concatenations, casts) is one
        kernel each way (layers.dense_ops.centernet_head_outputs); otherwise forward() + the reference's flattening."""
        if _HEAD_OUT and self.only_proposal and self.with_agn_hm and torch.is_grad_enabled() and self._multi(x):
            towers = self._run_tower_levels(self.bbox_tower, list(x))
            boths = conv3x3_multi(towers, self.agn_hm.weight, self.agn_hm.bias)
            if boths is not None and all(b.dtype == torch.bfloat16 and b.permute(0, 2, 3, 1).is_contiguous() for b in boths):
                return centernet_head_outputs(boths, [s.scale for s in self.scales[:len(boths)]])
            _, reg, hm = self._tail(x, towers, boths)
        else:
            _, reg, hm = self.forward(x)
        # float(bf16) is exact, so casting after the cat gives the values of the reference's order (cast, then cat)
        reg_flat = torch.cat([r.permute(0, 2, 3, 1).reshape(-1, 4) for r in reg], dim=0).float()
        hm_flat = torch.cat([h.permute(0, 2, 3, 1).reshape(-1) for h in hm], dim=0).float()
        return reg_flat, hm_flat

    def forward(self, x):
        multi = self._multi(x)
        towers = self._run_tower_levels(self.bbox_tower, list(x)) if multi else None
        # the grouped predictor convolution (agn_hm | bbox_pred) over all levels with one padded-copy launch per pass
        boths = conv3x3_multi(towers, self.agn_hm.weight, self.agn_hm.bias) if multi and self.with_agn_hm else None
        return self._tail(x, towers, boths)

    def _tail(self, x, towers, boths):
        multi = towers is not None
        clss, bbox_reg, agn_hms = [], [], []
        for l, feature in enumerate(x):
            if multi:
                cls_tower, bbox_tower = feature, towers[l]
            else:
                feature = self._run_tower(self.share_tower, feature)
                cls_tower = self._run_tower(self.cls_tower, feature)
                bbox_tower = self._run_tower(self.bbox_tower, feature)
            clss.append(None if self.only_proposal else self.cls_logits(cls_tower))
            if boths is not None:
                agn_hms.append(boths[l][:, :1])
                reg = boths[l][:, 1:5]
            elif self.with_agn_hm and conv3x3_group_usable(bbox_tower, self.agn_hm.weight, self.agn_hm.bias):
                # agn_hm (1 ch) and bbox_pred (4 ch) read the same tower output and are one arena parameter group (5 rows + zero
                # rows up to 64): ONE implicit GEMM each way on the group's views
                both = conv3x3_group(bbox_tower, self.agn_hm.weight, self.agn_hm.bias)
                agn_hms.append(both[:, :1])
                reg = both[:, 1:5]
            elif self.with_agn_hm:
                # outside the arena (no optimizer built yet, eval): one GEMM over the concatenated weights
                both = conv3x3(bbox_tower, torch.cat([self.agn_hm.weight, self.bbox_pred.weight], 0),
                               torch.cat([self.agn_hm.bias, self.bbox_pred.bias], 0))
                agn_hms.append(both[:, :1])
                reg = both[:, 1:5]
            else:
                agn_hms.append(None)
                reg = self.bbox_pred(bbox_tower)
            bbox_reg.append(F.relu(self.scales[l](reg)))
        return clss, bbox_reg, agn_hms
