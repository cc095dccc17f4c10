"""Oracle: the assembled training forward of CenterNet2 (Swin + FPN -> CenterNet proposals -> Detic cascade RoI heads +
mask head) on torch-CPU fp32, wired from the per-module restatements of this package.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Wiring follows
  DG/divergen/modeling/meta_arch/custom_rcnn.py:118-207          (CustomRCNN.forward, training branch)
  CN/modeling/dense_heads/centernet.py:179-235                   (CenterNet.forward: losses + proposals)
  DG/divergen/modeling/roi_heads/detic_roi_heads.py:192-338      (label_and_sample, cascade, mask branch)
  D2/modeling/proposal_generator/proposal_utils.py:126-196       (add_ground_truth_to_proposals)
The two random draws of the step (proposal sub-sampling, federated-loss class set) are INJECTED (`sample_fn`, `fed_fn`)
so that a test can run the product and the oracle on the same draw; likewise the proposals may be injected, because with
near-tied scores the top-k / NMS survivor SET is not stable under fp32 reordering between two implementations."""
import math

import torch

from . import centernet as C
from . import heads as H
from . import roi as R
from . import swin as S
from .quant import rb

BOX_W = ((10.0, 10.0, 5.0, 5.0), (20.0, 20.0, 10.0, 10.0), (30.0, 30.0, 15.0, 15.0))     # cascade_rcnn.py / Base yaml
IOUS = (0.6, 0.7, 0.8)


def backbone_and_dense(p, images, swin_size):
    c = S.SIZE2CONFIG[swin_size]
    feats = S.swin_forward(images, p, c["embed_dim"], c["depths"], c["num_heads"], c["ws"], prefix="backbone.bottom_up.")
    fp = H.fpn(feats, p, prefix="backbone.")
    levels = [fp[k] for k in ("p3", "p4", "p5", "p6", "p7")]
    regs, hms = H.centernet_head(levels, p, prefix="proposal_generator.centernet_head.")
    return fp, regs, hms


def centernet_losses(regs, hms, gt_boxes_list):
    shapes = [(int(r.shape[2]), int(r.shape[3])) for r in regs]
    pos, regt, hmt = C.ground_truth(gt_boxes_list, shapes)
    reg_pred = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, 4) for x in regs])
    agn = torch.cat([x.permute(0, 2, 3, 1).reshape(-1) for x in hms])
    return C.losses(pos, regt, hmt, reg_pred, agn)


def proposals_from_heatmaps(regs, hms, score_thresh, pre_topk, nms_thresh, post_topk):
    shapes = [(int(r.shape[2]), int(r.shape[3])) for r in regs]
    grids = C.compute_grids(shapes)
    B = regs[0].shape[0]
    out = []
    for i in range(B):
        bs, ss = [], []
        for l, s in enumerate(C.STRIDES):
            hm = hms[l][i, 0].reshape(-1).sigmoid()
            reg = (regs[l][i] * s).permute(1, 2, 0).reshape(-1, 4)
            b, sc = C.predict_level(grids[l], hm, reg, score_thresh, pre_topk)
            bs.append(b)
            ss.append(sc)
        out.append(C.nms_and_topk(torch.cat(bs), torch.cat(ss), nms_thresh, post_topk))
    return out


def _match(gt_boxes, gt_classes, boxes, thr, num_classes):
    if len(gt_boxes) == 0:
        return torch.zeros(len(boxes), dtype=torch.int64), torch.full((len(boxes),), num_classes, dtype=torch.int64)
    idx, lab = R.matcher(R.pairwise_iou(gt_boxes, boxes), [thr], [0, 1])
    cls = gt_classes[idx].clone()
    cls[lab == 0] = num_classes
    return idx, cls


class _ScaleGradient(torch.autograd.Function):
    """cascade_rcnn.py:20-28: identity forward, gradient x scale backward.  DeticCascadeROIHeads._run_stage
    (detic_roi_heads.py:396-414) applies it with 1 / num_cascade_stages to the pooled box features of every stage: the three stages'
    losses are summed, and the gradient they send into the SHARED feature maps is averaged."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


def roi_head_losses(p, fp, proposals, gts, image_sizes, num_classes, batch_per_image, pos_fraction, freq_weight, fed_num,
                    sample_fn, fed_fn, mask_weight=1.0, prefix="roi_heads.", stage_labels=None):
    """proposals: per image (boxes (n,4)).  gts: per image dict(boxes, classes, masks (n,H,W) bool).
    stage_labels (tests): {k: per image (keep mask (n,), classes (n_kept,), matched gt index (n_kept,))} for cascade stages
    k >= 1 -- the discrete outcome of `_match_and_label_boxes` handed in by the implementation under test, so that a refined
    box that sits within rounding of an IoU threshold cannot flip a label between the two sides; the boxes, features, logits
    and losses are still the oracle's own."""
    feats = [fp[k] for k in ("p3", "p4", "p5")]
    scales = (1 / 8, 1 / 16, 1 / 32)
    # label_and_sample_proposals (+ add_ground_truth_to_proposals)
    boxes, cls, midx = [], [], []
    for i, (pb, g) in enumerate(zip(proposals, gts)):
        b = torch.cat([pb, g["boxes"]])
        idx, c = _match(g["boxes"], g["classes"], b, IOUS[0], num_classes)
        fg_i, bg_i = sample_fn(i, c, batch_per_image, pos_fraction, num_classes)
        sel = torch.cat([fg_i, bg_i])
        boxes.append(b[sel])
        cls.append(c[sel])
        midx.append(idx[sel])
    losses = {}
    stage0 = [(b.clone(), c.clone(), m.clone()) for b, c, m in zip(boxes, cls, midx)]
    for k in range(3):
        if k > 0:
            nb, nc = [], []
            for i, g in enumerate(gts):
                b = R.apply_deltas(prev[i].detach(), boxes[i], BOX_W[k - 1])
                H_, W_ = image_sizes[i]
                b = torch.stack([b[:, 0].clamp(0, W_), b[:, 1].clamp(0, H_), b[:, 2].clamp(0, W_), b[:, 3].clamp(0, H_)], 1)
                keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
                if stage_labels is not None and k in stage_labels:
                    keep, c, idx = stage_labels[k][i]
                    b = b[keep]
                else:
                    b = b[keep]
                    idx, c = _match(g["boxes"], g["classes"], b, IOUS[k], num_classes)
                nb.append(b)
                nc.append(c)
                midx[i] = idx
            boxes, cls = nb, nc
        x = rb(R.roi_pooler(feats, boxes, 7, scales))         # (pooled features are stored in the feature maps' dtype)
        x = _ScaleGradient.apply(x, 1.0 / 3)                  # detic_roi_heads.py:403 (cascade_rcnn.py:20-28): features only
        x = H.box_head(x, p, "%sbox_head.%d." % (prefix, k))
        logits, deltas = H.box_predictor(x, p, "%sbox_predictor.%d." % (prefix, k))
        gtc = torch.cat(cls)
        gtb = torch.cat([g["boxes"][m] if len(g["boxes"]) else torch.zeros(len(m), 4) for g, m in zip(gts, midx)])
        appeared = fed_fn(k, gtc, fed_num, num_classes, freq_weight)
        losses["loss_cls_stage%d" % k] = H.sigmoid_ce_fed(logits, gtc, freq_weight, fed_num, appeared=appeared)
        losses["loss_box_reg_stage%d" % k] = H.box_reg_loss(torch.cat(boxes), gtb, deltas, gtc, num_classes, BOX_W[k])
        prev = list(deltas.split([len(b) for b in boxes]))
    # mask branch on the foreground of the stage-0 samples
    mb, mm = [], []
    for (b, c, m), g in zip(stage0, gts):
        fg = (c >= 0) & (c < num_classes)
        mb.append(b[fg])
        mm.append(g["masks"][m[fg]])
    if sum(len(b) for b in mb) == 0:
        losses["loss_mask"] = torch.zeros(())
    else:
        x = rb(R.roi_pooler(feats, mb, 14, scales))
        logits = H.mask_head(x, p, prefix + "mask_head.")
        losses["loss_mask"] = H.mask_loss(logits, mm, mb) * mask_weight
    return losses
