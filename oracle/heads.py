"""Oracle: FPN, CenterNet head, RoI box/mask heads and their losses on torch-CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional over reference-keyed state dicts.
  FPN.forward                      D2/modeling/backbone/fpn.py:113-154
  LastLevelP6P7_P5.forward         CN/modeling/backbone/fpn_p5.py:30-33
  CenterNetHead.forward            CN/modeling/dense_heads/centernet_head.py:141-162
  FastRCNNConvFCHead               D2/modeling/roi_heads/box_head.py:26-98
  DeticFastRCNNOutputLayers.forward DG/divergen/modeling/roi_heads/detic_fast_rcnn.py:437-466
  sigmoid_cross_entropy_loss       detic_fast_rcnn.py:203-235
  get_fed_loss_inds                DG/divergen/modeling/utils.py:16-28
  box_reg_loss                     detic_fast_rcnn.py:271-304 (smooth_l1, beta 0 => L1)
  MaskRCNNConvUpsampleHead.layers  D2/modeling/roi_heads/mask_head.py:209-284
  mask_rcnn_loss                   D2/modeling/roi_heads/mask_head.py:31-111
  fast_rcnn_inference_single_image D2/modeling/roi_heads/fast_rcnn.py:117-170 (pinned by tests/golden/fast_rcnn_inference.npz)
"""
import torch
import torch.nn.functional as F

from . import roi
from . import quant as Q
from .quant import rb


def fpn(feats, p, prefix="", in_features=("swin1", "swin2", "swin3")):
    """feats: dict name -> NCHW.  -> dict p3..p7.  Stages 3,4,5 for strides 8,16,32."""
    stages = [3, 4, 5]
    prev = None
    res = {}
    for name, s in reversed(list(zip(in_features, stages))):
        lat = rb(F.conv2d(feats[name], p["%sfpn_lateral%d.weight" % (prefix, s)], p["%sfpn_lateral%d.bias" % (prefix, s)]))
        if prev is not None:
            lat = rb(lat + F.interpolate(prev, scale_factor=2.0, mode="nearest"))
        prev = lat
        res["p%d" % s] = rb(F.conv2d(lat, p["%sfpn_output%d.weight" % (prefix, s)],
                                     p["%sfpn_output%d.bias" % (prefix, s)], padding=1))
    p6 = rb(F.conv2d(res["p5"], p[prefix + "top_block.p6.weight"], p[prefix + "top_block.p6.bias"], stride=2, padding=1))
    p7 = rb(F.conv2d(Q.relu(p6, "top_block.p6"), p[prefix + "top_block.p7.weight"], p[prefix + "top_block.p7.bias"], stride=2, padding=1))
    res["p6"], res["p7"] = p6, p7
    return {k: res[k] for k in ("p3", "p4", "p5", "p6", "p7")}


def centernet_head(xs, p, prefix="", num_box_convs=4):
    """xs: list of NCHW per level -> (reg list (relu'd), agn_hm logits list).  ONLY_PROPOSAL:
    cls tower has 0 convs; bbox tower = 4 x [conv3x3, GN(32), ReLU]."""
    regs, hms = [], []
    for l, x in enumerate(xs):
        t = x
        for i in range(num_box_convs):
            t = rb(F.conv2d(t, p["%sbbox_tower.%d.weight" % (prefix, 3 * i)], p["%sbbox_tower.%d.bias" % (prefix, 3 * i)], padding=1))
            t = F.group_norm(t, 32, p["%sbbox_tower.%d.weight" % (prefix, 3 * i + 1)], p["%sbbox_tower.%d.bias" % (prefix, 3 * i + 1)])
            t = rb(Q.relu(t, "tower.%d.%d" % (l, i)))
        hms.append(rb(F.conv2d(t, p[prefix + "agn_hm.weight"], p[prefix + "agn_hm.bias"], padding=1)))
        r = rb(F.conv2d(t, p[prefix + "bbox_pred.weight"], p[prefix + "bbox_pred.bias"], padding=1))
        regs.append(Q.relu(r * p["%sscales.%d.scale" % (prefix, l)], "reg.%d" % l))
    return regs, hms


def box_head(x, p, prefix):
    """(R,256,7,7) -> (R,1024): flatten, fc1, relu, fc2, relu."""
    x = x.flatten(1)
    x = Q.relu(rb(F.linear(x, p[prefix + "fc1.weight"], p[prefix + "fc1.bias"])), prefix + "fc1")
    return Q.relu(rb(F.linear(x, p[prefix + "fc2.weight"], p[prefix + "fc2.bias"])), prefix + "fc2")


def box_predictor(x, p, prefix):
    return (rb(F.linear(x, p[prefix + "cls_score.weight"], p[prefix + "cls_score.bias"])),
            rb(F.linear(x, p[prefix + "bbox_pred.weight"], p[prefix + "bbox_pred.bias"])))


def fed_loss_inds(gt_classes, num_sample_cats, C, weight):
    """utils.py:16-28 -- consumes one torch.multinomial draw from the global generator."""
    appeared = torch.unique(gt_classes)
    prob = torch.ones(C + 1, dtype=torch.float32)
    prob[-1] = 0
    if len(appeared) < num_sample_cats:
        prob[:C] = weight.float().clone()
        prob[appeared] = 0
        more = torch.multinomial(prob, num_sample_cats - len(appeared), replacement=False)
        appeared = torch.cat([appeared, more])
    return appeared


def sigmoid_ce_fed(logits, gt_classes, freq_weight, num_sample_cats=50, appeared=None):
    """detic_fast_rcnn.py:203-235 (USE_SIGMOID_CE + USE_FED_LOSS)."""
    if logits.numel() == 0:
        return logits.new_zeros([1])[0]
    B, C = logits.shape[0], logits.shape[1] - 1
    target = logits.new_zeros(B, C + 1)
    target[torch.arange(B), gt_classes] = 1
    target = target[:, :C]
    if appeared is None:
        appeared = fed_loss_inds(gt_classes, num_sample_cats, C, freq_weight)
    m = torch.zeros(C + 1)
    m[appeared] = 1
    w = m[:C].reshape(1, C)
    ce = F.binary_cross_entropy_with_logits(logits[:, :-1], target, reduction="none")
    return torch.sum(ce * w) / B


def box_reg_loss(prop_boxes, gt_boxes, pred_deltas, gt_classes, num_classes, weights):
    """detic_fast_rcnn.py:271-304 (class-agnostic, smooth_l1 beta=0 => |.|, mean over 4*n_fg)."""
    fg = torch.nonzero((gt_classes >= 0) & (gt_classes < num_classes)).squeeze(1)
    tgt = roi.get_deltas(prop_boxes[fg], gt_boxes[fg], weights)
    l = torch.abs(pred_deltas[fg] - tgt)
    return l.sum() / max(l.numel(), 1.0)


def mask_head(x, p, prefix, num_conv=4):
    for i in range(num_conv):
        x = Q.relu(rb(F.conv2d(x, p["%smask_fcn%d.weight" % (prefix, i + 1)], p["%smask_fcn%d.bias" % (prefix, i + 1)], padding=1)),
                   "%smask_fcn%d" % (prefix, i + 1))
    x = Q.relu(rb(F.conv_transpose2d(x, p[prefix + "deconv.weight"], p[prefix + "deconv.bias"], stride=2)), prefix + "deconv")
    return rb(F.conv2d(x, p[prefix + "predictor.weight"], p[prefix + "predictor.bias"]))


def mask_loss(mask_logits, gt_masks_list, prop_boxes_list):
    """mask_head.py:31-111, class-agnostic.  gt_masks_list: per image (n,H,W) bool for the fg
    proposals' matched GT; prop_boxes_list: per image (n,4)."""
    S = mask_logits.shape[2]
    tg = [roi.crop_and_resize(m, b, S) for m, b in zip(gt_masks_list, prop_boxes_list) if len(b)]
    if not tg:
        return mask_logits.sum() * 0
    tg = torch.cat(tg).to(torch.float32)
    return F.binary_cross_entropy_with_logits(mask_logits[:, 0], tg, reduction="mean")


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    """D2/modeling/roi_heads/fast_rcnn.py:117-170: rows with a non-finite box or score are dropped, the background column is cut,
    boxes are clipped to the image (structures/boxes.py:200-214: x to [0, w], y to [0, h]), (row, class) pairs above the score
    threshold go through per-class greedy NMS (kept in descending score order, stable), the best `topk_per_image` stay.
    Returns boxes (n, 4), scores (n,), classes (n,), proposal rows (n,) -- rows index the FILTERED (finite) proposal list, as
    in the reference."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    boxes, scores = boxes[valid], scores[valid][:, :-1]
    nreg = boxes.shape[1] // 4
    h, w = image_shape
    b = boxes.reshape(-1, 4).clone()
    b[:, 0].clamp_(min=0, max=w); b[:, 2].clamp_(min=0, max=w)
    b[:, 1].clamp_(min=0, max=h); b[:, 3].clamp_(min=0, max=h)
    b = b.view(-1, nreg, 4)
    rows, cls = torch.nonzero(scores > score_thresh, as_tuple=True)          # row-major order = the reference's nonzero()
    cand = b[rows, 0] if nreg == 1 else b[rows, cls]
    sc = scores[rows, cls]
    keep = roi.batched_nms(cand, sc, cls, nms_thresh)
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    return cand[keep], sc[keep], cls[keep], rows[keep]
