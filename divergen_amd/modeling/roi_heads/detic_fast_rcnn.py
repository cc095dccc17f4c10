"""DeticFastRCNNOutputLayers: cls_score / bbox_pred + sigmoid-CE with federated loss + L1 box loss.
Mirrors DG/divergen/modeling/roi_heads/detic_fast_rcnn.py:31-466 and
D2/modeling/roi_heads/fast_rcnn.py:45-460 for the configuration the shipped YAMLs select
(USE_SIGMOID_CE, USE_FED_LOSS, CLS_AGNOSTIC_BBOX_REG, smooth_l1 beta 0, no zero-shot classifier)."""
import json
import math

import torch
from torch import nn
from torch.nn import functional as F

from ...config import configurable
from ...layers import batched_nms
from ...layers.linear_ops import Linear, group_parameters, linear_padded
from ...structures import Boxes, Instances
from ...utils.events import get_event_storage
from ..box_regression import Box2BoxTransform


import os
_FUSED_LOSSES = True


def load_class_freq(path="datasets/metadata/lvis_v1_train_cat_info.json", freq_weight=1.0):
    """DG/divergen/modeling/utils.py:7-13."""
    cat_info = json.load(open(path, "r"))
    cat_info = torch.tensor([c["image_count"] for c in sorted(cat_info, key=lambda x: x["id"])])
    return cat_info.float() ** freq_weight


def get_fed_loss_inds(gt_classes, num_sample_cats, C, weight=None):
    """DG/divergen/modeling/utils.py:16-28 (torch.unique + torch.multinomial keep the RNG contract)."""
    appeared = torch.unique(gt_classes)
    prob = appeared.new_ones(C + 1).float()
    prob[-1] = 0
    if len(appeared) < num_sample_cats:
        if weight is not None:
            prob[:C] = weight.float().clone()
        prob[appeared] = 0
        more = torch.multinomial(prob, num_sample_cats - len(appeared), replacement=False)
        appeared = torch.cat([appeared, more])
    return appeared


_FED_CONSTS = {}


def fed_loss_class_mask(gt_classes, num_sample_cats, C, weight):
    """get_fed_loss_inds (DG/divergen/modeling/utils.py:16-28) as a (C+1,) 0/1 mask, without reading anything back
    to the host: the classes that appear, plus -- when fewer than num_sample_cats appear -- classes drawn without
    replacement with probability ~ weight among the others.  torch.multinomial(prob, k, replacement=False) IS
    `topk(prob / Exponential(1), k)`; the same draw is made here with k = num_sample_cats and only the first
    num_sample_cats - n_appeared of it kept, so the class SET is the one the reference code would obtain from the
    same generator state (when n_appeared >= num_sample_cats the reference draws nothing: the streams then differ)."""
    dev = gt_classes.device
    key = (C, str(dev), None if weight is None else (weight.data_ptr(), weight._version))
    if key not in _FED_CONSTS:       # the sampling weights do not change between calls
        prob0 = torch.ones(C + 1, dtype=torch.float32, device=dev) if weight is None else \
            torch.cat([weight.float(), weight.new_zeros(1).float()])
        prob0[C:].zero_()
        if len(_FED_CONSTS) > 16:
            _FED_CONSTS.clear()
        _FED_CONSTS[key] = prob0.contiguous()
    prob0 = _FED_CONSTS[key]
    expo = torch.empty_like(prob0).exponential_(1)          # torch's generator: the reference's position in the random stream
    if not gt_classes.is_cuda:
        app = torch.zeros(C + 1, dtype=torch.bool, device=dev)
        app[gt_classes.clamp(min=0)] = True
        q = prob0.masked_fill(app, 0) / expo
        vals, idx = torch.topk(q, min(num_sample_cats, C + 1))
        take = (torch.arange(idx.numel(), device=dev) < num_sample_cats - app.sum()) & (vals > 0)
        return app.index_put((idx,), app[idx] | take)
    from ... import _lib as L
    mask = torch.empty(C + 1, dtype=torch.uint8, device=dev)
    gt = gt_classes.contiguous()
    L.check(L.lib().dgx_fed_class_mask(L.ptr(gt) if gt.numel() else None, gt.numel(), L.ptr(prob0), L.ptr(expo), C, int(num_sample_cats),
                                       L.ptr(mask), L.stream()), "dgx_fed_class_mask")
    return mask.view(torch.bool)


class _DeticLosses(torch.autograd.Function):
    """loss_cls, loss_box_reg and the classification statistics of one cascade stage: libdgx dgx_detic_losses."""

    @staticmethod
    def forward(ctx, logits, deltas, gt_classes, class_w, prop, gtb, src, weights):
        from ... import _lib as L
        R, C1 = logits.shape
        logits, deltas = logits.contiguous(), deltas.contiguous()
        if deltas.dtype != logits.dtype:
            deltas = deltas.to(logits.dtype)
        dlogits = torch.empty_like(logits)
        dsign = torch.empty(R, 4, dtype=torch.float32, device=logits.device)
        out = torch.empty(16, dtype=torch.float32, device=logits.device)
        part = torch.empty(max(R, 1) * 8, dtype=torch.float32, device=logits.device)
        L.check(L.lib().dgx_detic_losses(L.ptr(logits), L.ptr(deltas), L.ptr(gt_classes.contiguous()),
                                         L.ptr(class_w.float().contiguous()) if class_w is not None else None,
                                         L.ptr(prop.float().contiguous()), L.ptr(gtb.float().contiguous()),
                                         L.ptr(src.contiguous()) if src is not None else None, R, C1 - 1,
                                         float(weights[0]), float(weights[1]), float(weights[2]), float(weights[3]),
                                         L.ptr(dlogits), L.ptr(dsign), L.ptr(out), L.ptr(part), L.dtype_code(logits), L.stream()),
                "dgx_detic_losses")
        ctx.save_for_backward(dlogits, dsign, out)
        ctx.ddt = deltas.dtype
        ctx.mark_non_differentiable(out)
        return out[8], out[9], out

    @staticmethod
    def backward(ctx, g_cls, g_box, _):
        dlogits, dsign, out = ctx.saved_tensors
        return dlogits * (g_cls * out[14]).to(dlogits.dtype), (dsign * (g_box * out[10])).to(ctx.ddt), None, None, None, None, None, None


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    """D2/modeling/roi_heads/fast_rcnn.py:117-170."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not valid.all():
        boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]
    nreg = boxes.shape[1] // 4
    b = Boxes(boxes.reshape(-1, 4))
    b.clip(image_shape)
    boxes = b.tensor.view(-1, nreg, 4)
    filter_mask = scores > score_thresh
    filter_inds = filter_mask.nonzero()
    boxes = boxes[filter_inds[:, 0], 0] if nreg == 1 else boxes[filter_mask]
    scores = scores[filter_mask]
    keep = batched_nms(boxes, scores, filter_inds[:, 1], nms_thresh)
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    boxes, scores, filter_inds = boxes[keep], scores[keep], filter_inds[keep]
    result = Instances(image_shape)
    result.pred_boxes = Boxes(boxes)
    result.scores = scores
    result.pred_classes = filter_inds[:, 1]
    return result, filter_inds[:, 0]


def fast_rcnn_inference(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image):
    res = [fast_rcnn_inference_single_image(b, s, sh, score_thresh, nms_thresh, topk_per_image)
           for s, b, sh in zip(scores, boxes, image_shapes)]
    return [x[0] for x in res], [x[1] for x in res]


def _log_classification_stats(pred_logits, gt_classes, prefix="fast_rcnn"):
    """fast_rcnn.py:88-114, kept on the device (writers convert lazily)."""
    n = gt_classes.numel()
    if n == 0:
        return
    with torch.no_grad():
        pred = pred_logits.argmax(dim=1)
        bg = pred_logits.shape[1] - 1
        fg = (gt_classes >= 0) & (gt_classes < bg)
        num_fg = fg.sum().clamp(min=1)
        st = get_event_storage()
        st.put_scalar(prefix + "/cls_accuracy", (pred == gt_classes).sum() / n)
        st.put_scalar(prefix + "/fg_cls_accuracy", ((pred == gt_classes) & fg).sum() / num_fg)
        st.put_scalar(prefix + "/false_negative", ((pred == bg) & fg).sum() / num_fg)


class DeticFastRCNNOutputLayers(nn.Module):
    @configurable
    def __init__(self, input_shape, *, box2box_transform, num_classes, test_score_thresh=0.0, test_nms_thresh=0.5,
                 test_topk_per_image=100, cls_agnostic_bbox_reg=False, smooth_l1_beta=0.0, box_reg_loss_type="smooth_l1",
                 loss_weight=1.0, mult_proposal_score=False, use_sigmoid_ce=False, use_fed_loss=False,
                 ignore_zero_cats=False, fed_loss_num_cat=50, prior_prob=0.01, cat_freq_path="",
                 fed_loss_freq_weight=0.5, use_zeroshot_cls=False, divergen_box_loss=True, only_paste_sup=False, **unused):
        super().__init__()
        if use_zeroshot_cls or box_reg_loss_type != "smooth_l1":
            raise NotImplementedError("USE_ZEROSHOT_CLS / non-smooth_l1 box losses are outside the shipped configs")
        self.num_classes = num_classes
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.cls_score = Linear(input_size, num_classes + 1)
        self.bbox_pred = Linear(input_size, (1 if cls_agnostic_bbox_reg else num_classes) * 4)
        # cls_score and bbox_pred read the same features: their rows sit back to back in the parameter arena (1454 + 4 -> 1464
        # rows), one GEMM each way serves both (forward, input gradient, weight gradient)
        group_parameters(self.cls_score.weight, self.bbox_pred.weight)
        group_parameters(self.cls_score.bias, self.bbox_pred.bias)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in (self.cls_score, self.bbox_pred):
            nn.init.constant_(l.bias, 0)
        self.box2box_transform, self.smooth_l1_beta = box2box_transform, smooth_l1_beta
        self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image = test_score_thresh, test_nms_thresh, test_topk_per_image
        self.mult_proposal_score, self.use_sigmoid_ce, self.use_fed_loss = mult_proposal_score, use_sigmoid_ce, use_fed_loss
        self.ignore_zero_cats, self.fed_loss_num_cat, self.divergen_box_loss = ignore_zero_cats, fed_loss_num_cat, divergen_box_loss
        # BSGAL (BS/bsgal/modeling/roi_heads/detic_fast_rcnn.py:222-247): also report the classification loss of the rows matched
        # to pasted / to original instances (`loss_paste_ins`, `loss_nopaste_ins`), the terms its gradient comparison differentiates
        self.only_paste_sup = only_paste_sup
        if use_sigmoid_ce:
            nn.init.constant_(self.cls_score.bias, -math.log((1 - prior_prob) / prior_prob))
        if use_fed_loss or ignore_zero_cats:
            fw = load_class_freq(cat_freq_path, fed_loss_freq_weight)
            if use_fed_loss and len(fw) < num_classes:
                fw = torch.cat([fw, fw.new_zeros(num_classes - len(fw))])
            self.register_buffer("freq_weight", fw)
        else:
            self.freq_weight = None

    @classmethod
    def from_config(cls, cfg, input_shape, box2box_transform=None):
        h = cfg.MODEL.ROI_BOX_HEAD
        return dict(input_shape=input_shape,
                    box2box_transform=box2box_transform or Box2BoxTransform(weights=h.BBOX_REG_WEIGHTS),
                    num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES, cls_agnostic_bbox_reg=h.CLS_AGNOSTIC_BBOX_REG,
                    smooth_l1_beta=h.SMOOTH_L1_BETA, test_score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                    test_nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
                    box_reg_loss_type=h.BBOX_REG_LOSS_TYPE, mult_proposal_score=h.MULT_PROPOSAL_SCORE,
                    use_sigmoid_ce=h.USE_SIGMOID_CE, use_fed_loss=h.USE_FED_LOSS, ignore_zero_cats=h.IGNORE_ZERO_CATS,
                    fed_loss_num_cat=h.FED_LOSS_NUM_CAT, prior_prob=h.PRIOR_PROB, cat_freq_path=h.CAT_FREQ_PATH,
                    fed_loss_freq_weight=h.FED_LOSS_FREQ_WEIGHT, use_zeroshot_cls=h.USE_ZEROSHOT_CLS,
                    divergen_box_loss=cfg.MODEL.USE_DIVERGEN_BOX_LOSS and cfg.MODEL.get("USE_XPASTE_BOX_LOSS", True),
                    only_paste_sup=cfg.MODEL.get("ONLY_PASTE_SUP", False))

    def forward(self, x, classifier_info=(None, None, None)):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        y = self.forward_joint(x)
        if y is not None:
            c1, nb = self.cls_score.out_features, self.bbox_pred.out_features
            return y[:, :c1], y[:, c1:c1 + nb]
        return self.cls_score(x), self.bbox_pred(x)

    def forward_joint(self, x):
        """(R, pad8(C + 1 + 4)) logits | box deltas | zero columns from ONE GEMM over the arena group, or None when the parameters
        are not arena resident (module used on its own)."""
        w = self.cls_score.weight
        if getattr(w, "_dgx16g", None) is None:
            return None
        return linear_padded(x, w, self.cls_score.bias)

    @property
    def fused_supported(self):
        """The one-pass loss / cascade kernels cover the shipped recipe: sigmoid CE, class-agnostic L1 box regression."""
        return _FUSED_LOSSES and self.use_sigmoid_ce and self.bbox_pred.out_features == 4 and self.smooth_l1_beta < 1e-5

    def fused_ok(self, scores, deltas):
        return (_FUSED_LOSSES and scores.is_cuda and scores.shape[0] > 0 and self.use_sigmoid_ce and deltas.shape[1] == 4
                and self.smooth_l1_beta < 1e-5)

    def _fused_losses(self, predictions, proposals):
        scores, deltas = predictions
        gt_classes = torch.cat([p.gt_classes for p in proposals], dim=0)
        prop = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        gtb = torch.cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals], dim=0)
        has_src = all(p.has("instance_source") for p in proposals)
        src = torch.cat([p.instance_source for p in proposals if len(p)], dim=0) if has_src else None
        return self.losses_from_tensors(scores, deltas, gt_classes, prop, gtb, src)

    def _class_weight(self, gt_classes, C):
        """The (C,) weight of the sigmoid CE: federated class sample x zero-frequency mask (None = all ones)."""
        w = None
        if self.use_fed_loss and self.freq_weight is not None:
            w = fed_loss_class_mask(gt_classes, self.fed_loss_num_cat, C, self.freq_weight)[:C].float()
        if self.ignore_zero_cats and self.freq_weight is not None:
            z = (self.freq_weight.view(-1) > 1e-4).float()
            w = z if w is None else w * z
        return w

    def paste_split(self, scores, gt_classes, w, src):
        """sigmoid_cross_entropy_loss_with_fed (:431-470): the weighted BCE summed over the rows of pasted (source >= 1) and of
        original (source == 0) instances, both divided by the number of rows B (rows labelled -1 = padding are not rows)."""
        C = scores.shape[1] - 1
        valid = gt_classes >= 0
        B = valid.sum().clamp(min=1).float()
        target = torch.nn.functional.one_hot(gt_classes.clamp(min=0), C + 1)[:, :C].to(torch.float32)
        ce = F.binary_cross_entropy_with_logits(scores[:, :C].float(), target, reduction="none")
        if w is not None:
            ce = ce * w.view(1, C)
        row = ce.sum(1) * valid
        return (row * (src >= 1)).sum() / B, (row * (src == 0)).sum() / B

    def losses_from_tensors(self, scores, deltas, gt_classes, prop, gtb, src):
        """One kernel pair for loss_cls + loss_box_reg + logging statistics (sigmoid CE, class-agnostic L1);
        rows with gt_classes < 0 are ignored.  `src` (instance_source) is used only when the DiverGen box loss is off."""
        paste_src = src
        if self.divergen_box_loss:
            src = None
        C = scores.shape[1] - 1
        w = self._class_weight(gt_classes, C)
        with torch.autocast("cuda", enabled=False):
            loss_cls, loss_box, out = _DeticLosses.apply(scores, deltas, gt_classes, w, prop, gtb, src, self.box2box_transform.weights)
        st = get_event_storage()
        st.put_scalar("fast_rcnn/cls_accuracy", out[11])
        st.put_scalar("fast_rcnn/fg_cls_accuracy", out[12])
        st.put_scalar("fast_rcnn/false_negative", out[13])
        losses = {"loss_cls": loss_cls, "loss_box_reg": loss_box}
        if self.only_paste_sup and paste_src is not None:
            losses["loss_paste_ins"], losses["loss_nopaste_ins"] = self.paste_split(scores, gt_classes, w, paste_src)
        return losses

    def losses(self, predictions, proposals, classifier_info=(None, None, None)):
        if len(proposals) and self.fused_ok(predictions[0], predictions[1]):
            return self._fused_losses(predictions, proposals)
        scores, deltas = predictions[0].float(), predictions[1].float()
        gt_classes = torch.cat([p.gt_classes for p in proposals], dim=0) if len(proposals) else torch.empty(0)
        _log_classification_stats(scores, gt_classes)
        if len(proposals):
            prop = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
            gtb = torch.cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals], dim=0)
        else:
            prop = gtb = torch.empty((0, 4), device=deltas.device)
        w = self._class_weight(gt_classes, scores.shape[1] - 1) if (self.use_sigmoid_ce and scores.numel()) else None
        loss_cls = self.sigmoid_cross_entropy_loss(scores, gt_classes, w) if self.use_sigmoid_ce else \
            F.cross_entropy(scores, gt_classes, reduction="mean")
        has_src = len(proposals) > 0 and all(p.has("instance_source") for p in proposals)
        paste_src = torch.cat([p.instance_source for p in proposals if len(p)], dim=0) if has_src else None
        src = None if self.divergen_box_loss else paste_src
        losses = {"loss_cls": loss_cls, "loss_box_reg": self.box_reg_loss(prop, gtb, deltas, gt_classes, src)}
        if self.only_paste_sup and paste_src is not None and self.use_sigmoid_ce and scores.numel():
            losses["loss_paste_ins"], losses["loss_nopaste_ins"] = self.paste_split(scores, gt_classes, w, paste_src)
        return losses

    def no_grad_losses(self, predictions, proposals, classifier_info=(None, None, None)):
        """BS detic_fast_rcnn.py:268-352 for the sigmoid-CE recipe: the losses of the held-out pass over ground-truth
        proposals -- classification WITHOUT the federated / zero-frequency class weights (:393-430), box regression as usual.
        (The per-paste loss matrix it can also return belongs to ACTIVE_ONLY_GT_TRAIN, which no shipped configuration sets.)"""
        assert self.use_sigmoid_ce
        scores, deltas = predictions[0].float(), predictions[1].float()
        gt_classes = torch.cat([p.gt_classes for p in proposals], dim=0) if len(proposals) else torch.empty(0)
        _log_classification_stats(scores, gt_classes)
        prop = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        gtb = torch.cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals], dim=0)
        loss_cls = self.sigmoid_cross_entropy_loss(scores, gt_classes, "none")
        has_src = all(p.has("instance_source") for p in proposals)
        src = None if (self.divergen_box_loss or not has_src) else torch.cat([p.instance_source for p in proposals if len(p)], dim=0)
        return {"loss_cls": loss_cls, "loss_box_reg": self.box_reg_loss(prop, gtb, deltas, gt_classes, src)}

    def sigmoid_cross_entropy_loss(self, logits, gt_classes, weight=None):
        """weight: a (C,) class weight computed by the caller (`_class_weight`), "none" = unweighted (BSGAL's no-fed form),
        None = draw it here (the reference's own call pattern)."""
        if logits.numel() == 0:
            return logits.new_zeros([1])[0]
        B, C = logits.shape[0], logits.shape[1] - 1
        target = logits.new_zeros(B, C + 1)
        target[torch.arange(B, device=logits.device), gt_classes] = 1
        target = target[:, :C]
        if isinstance(weight, str):
            w = 1
        elif weight is not None:
            w = weight.view(1, C)
        else:
            w = 1
            if self.use_fed_loss and self.freq_weight is not None:
                appeared = get_fed_loss_inds(gt_classes, self.fed_loss_num_cat, C, self.freq_weight)
                m = appeared.new_zeros(C + 1)
                m[appeared] = 1
                w = w * m[:C].view(1, C).float()
            if self.ignore_zero_cats and self.freq_weight is not None:
                w = w * (self.freq_weight.view(-1) > 1e-4).float().view(1, C)
        ce = F.binary_cross_entropy_with_logits(logits[:, :-1], target, reduction="none")
        return torch.sum(ce * w) / B

    def box_reg_loss(self, prop, gtb, deltas, gt_classes, instance_source=None):
        fg = ((gt_classes >= 0) & (gt_classes < self.num_classes)).nonzero().squeeze(1)
        fgd = deltas[fg] if deltas.shape[1] == 4 else deltas.view(-1, self.num_classes, 4)[fg, gt_classes[fg]]
        tgt = self.box2box_transform.get_deltas(prop[fg], gtb[fg])
        l = torch.abs(fgd - tgt) if self.smooth_l1_beta < 1e-5 else \
            torch.where((fgd - tgt).abs() < self.smooth_l1_beta, 0.5 * (fgd - tgt) ** 2 / self.smooth_l1_beta,
                        (fgd - tgt).abs() - 0.5 * self.smooth_l1_beta)
        if instance_source is not None:
            l = l[instance_source[fg] == 0]
        return l.sum() / max(l.numel(), 1.0)

    def predict_boxes(self, predictions, proposals):
        if not len(proposals):
            return []
        deltas = predictions[1]
        prop = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        return self.box2box_transform.apply_deltas(deltas, prop).split([len(p) for p in proposals])

    def predict_probs(self, predictions, proposals):
        s = predictions[0].float()
        probs = s.sigmoid() if self.use_sigmoid_ce else F.softmax(s, dim=-1)
        return probs.split([len(p) for p in proposals], dim=0)
