"""MaskRCNNConvUpsampleHead + mask_rcnn_loss.  Mirrors D2/modeling/roi_heads/mask_head.py:31-284
(module names mask_fcn{1..4}, deconv, predictor).  The GT target crop is the byte-tap HIP kernel
(no fp32 copy of the full-resolution masks)."""
import torch
from torch import nn
from torch.nn import functional as F

from .. import ROI_MASK_HEAD_REGISTRY
from ...config import configurable
from ...layers.conv_ops import Conv2d, ConvTranspose2d
from ...layers.mask_ops import mask_bce_with_stats
from ...utils.events import DeferredScalar, get_event_storage
from ..backbone.fpn import c2_msra_fill


def mask_rcnn_loss(pred_mask_logits, instances, vis_period=0):
    cls_agnostic = pred_mask_logits.size(1) == 1
    S = pred_mask_logits.size(2)
    gt_classes, gt_masks = [], []
    for inst in instances:
        if len(inst) == 0:
            continue
        if not cls_agnostic:
            gt_classes.append(inst.gt_classes.to(dtype=torch.int64))
        gt_masks.append(inst.gt_masks.crop_and_resize(inst.proposal_boxes.tensor, S))
    if len(gt_masks) == 0:
        return pred_mask_logits.sum() * 0
    gt_masks = torch.cat(gt_masks, dim=0)
    if cls_agnostic:
        pred = pred_mask_logits[:, 0]
    else:
        idx = torch.arange(pred_mask_logits.size(0), device=pred_mask_logits.device)
        pred = pred_mask_logits[idx, torch.cat(gt_classes, dim=0)]
    if pred.is_cuda and pred.dtype in (torch.float32, torch.bfloat16):
        # loss, gradient and the three statistics in one pass over the logits (dgx_mask_bce)
        loss, stats = mask_bce_with_stats(pred, gt_masks)
        # statistics: the raw counters stay on the device, the three ratios are formed when a writer reads them
        n = float(max(gt_masks.numel(), 1))
        st = get_event_storage()
        st.put_scalar("mask_rcnn/accuracy", DeferredScalar(lambda s, n=n: 1 - s[1] / n, stats))
        st.put_scalar("mask_rcnn/false_positive", DeferredScalar(lambda s, n=n: s[2] / max(n - s[4], 1.0), stats))
        st.put_scalar("mask_rcnn/false_negative", DeferredScalar(lambda s: s[3] / max(s[4], 1.0), stats))
        return loss
    gt_bool = gt_masks
    with torch.no_grad():
        incorrect = (pred > 0.0) != gt_bool
        npos = gt_bool.sum()
        st = get_event_storage()
        st.put_scalar("mask_rcnn/accuracy", 1 - incorrect.sum() / max(incorrect.numel(), 1.0))
        st.put_scalar("mask_rcnn/false_positive", (incorrect & ~gt_bool).sum() / (gt_bool.numel() - npos).clamp(min=1.0))
        st.put_scalar("mask_rcnn/false_negative", (incorrect & gt_bool).sum() / npos.clamp(min=1.0))
    return F.binary_cross_entropy_with_logits(pred.float(), gt_masks.to(dtype=torch.float32), reduction="mean")


def mask_rcnn_inference(pred_mask_logits, pred_instances):
    if pred_mask_logits.size(1) == 1:
        probs = pred_mask_logits.sigmoid()
    else:
        n = pred_mask_logits.shape[0]
        cls = torch.cat([i.pred_classes for i in pred_instances])
        probs = pred_mask_logits[torch.arange(n, device=cls.device), cls][:, None].sigmoid()
    for prob, inst in zip(probs.split([len(i) for i in pred_instances], dim=0), pred_instances):
        inst.pred_masks = prob


@ROI_MASK_HEAD_REGISTRY.register()
class MaskRCNNConvUpsampleHead(nn.Sequential):
    @configurable
    def __init__(self, input_shape, *, num_classes, conv_dims, conv_norm="", loss_weight=1.0, vis_period=0):
        super().__init__()
        assert conv_norm == "" and len(conv_dims) >= 1
        self.vis_period, self.loss_weight = vis_period, loss_weight
        cur = input_shape.channels
        self.conv_norm_relus = []
        for k, d in enumerate(conv_dims[:-1]):
            conv = Conv2d(cur, d, 3, 1, 1)
            self.add_module("mask_fcn%d" % (k + 1), conv)
            self.add_module("mask_fcn_relu%d" % (k + 1), nn.ReLU())
            self.conv_norm_relus.append(conv)
            cur = d
        self.deconv = ConvTranspose2d(cur, conv_dims[-1], 2, 2, 0)
        self.add_module("deconv_relu", nn.ReLU())
        self.predictor = Conv2d(conv_dims[-1], num_classes, 1, 1, 0)
        for l in self.conv_norm_relus + [self.deconv]:
            c2_msra_fill(l)
        nn.init.normal_(self.predictor.weight, std=0.001)
        nn.init.constant_(self.predictor.bias, 0)

    @classmethod
    def from_config(cls, cfg, input_shape):
        m = cfg.MODEL.ROI_MASK_HEAD
        return dict(input_shape=input_shape, conv_dims=[m.CONV_DIM] * (m.NUM_CONV + 1), conv_norm=m.NORM,
                    num_classes=1 if m.CLS_AGNOSTIC_MASK else cfg.MODEL.ROI_HEADS.NUM_CLASSES, vis_period=cfg.VIS_PERIOD)

    def layers(self, x):
        """conv / deconv layers take the ReLU that follows them into their GEMM epilogue."""
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, (Conv2d, ConvTranspose2d)) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                x = m(x, relu=True)
                i += 2
            else:
                x = m(x)
                i += 1
        return x

    def forward(self, x, instances):
        x = self.layers(x)[: sum(len(i) for i in instances)]   # drop the shape-padding RoIs
        if self.training:
            return {"loss_mask": mask_rcnn_loss(x, instances, self.vis_period) * self.loss_weight}
        mask_rcnn_inference(x, instances)
        return instances


def build_mask_head(cfg, input_shape):
    return ROI_MASK_HEAD_REGISTRY.get(cfg.MODEL.ROI_MASK_HEAD.NAME)(cfg, input_shape)
