"""Oracle: ROIAlign / ROIPooler / mask crop / NMS / box utilities on CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The arithmetic of roi_align and nms lives in
oracle/roi_nms.c (restating torchvision, which the reference calls but does not vendor).
Reference call sites followed here (D2 = BSGAL/third_party/CenterNet2/detectron2):
  ROIAlign.forward            D2/layers/roi_align.py:49-65
  assign_boxes_to_levels      D2/modeling/poolers.py:22-58
  convert_boxes_to_pooler_format D2/modeling/poolers.py:61-90
  ROIPooler.forward           D2/modeling/poolers.py:185-245
  BitMasks.crop_and_resize    D2/structures/masks.py:189-220
  batched_nms                 D2/layers/nms.py:9-20 -> torchvision.ops.boxes.batched_nms
  pairwise_iou                D2/structures/boxes.py:310-357
  Matcher.__call__            D2/modeling/matcher.py:62-104
  subsample_labels            D2/modeling/sampling.py:9-54
  Box2BoxTransform            D2/modeling/box_regression.py:43-118
"""
import ctypes
import math
import os

import numpy as np
import torch

from .build import build_oracle

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_oracle())
        _LIB.oracle_nms.restype = ctypes.c_int64
    return _LIB


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _RoIAlignFn(torch.autograd.Function):
    """ROIAlign as the reference's autograd sees it (D2/layers/roi_align.py:49-65 -> torchvision.ops.roi_align: differentiable with
    respect to the feature map, not to the boxes): forward and backward are the two C functions of oracle/roi_nms.c, both pinned on
    the reference's known-answer test and on ROIAlignRotated_cpu.cpp compiled in place (tests/test_oracle_roi.py)."""

    @staticmethod
    def forward(ctx, inp, rois, scale, out_size, sampling_ratio, aligned):
        ctx.save_for_backward(rois)
        ctx.meta = (scale, tuple(inp.shape), sampling_ratio, aligned)
        return _roi_align_values(inp, rois, scale, out_size, sampling_ratio, aligned)

    @staticmethod
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        scale, in_shape, sampling_ratio, aligned = ctx.meta
        return roi_align_backward(grad_out.contiguous(), rois, scale, in_shape, sampling_ratio, aligned), None, None, None, None, None


def roi_align(inp, rois, scale, out_size, sampling_ratio=0, aligned=True):
    """inp (N,C,H,W) float32 tensor, rois (R,5) -> (R,C,ph,pw); differentiable w.r.t. inp when it requires grad."""
    if torch.is_grad_enabled() and inp.requires_grad:
        return _RoIAlignFn.apply(inp, rois, scale, out_size, sampling_ratio, aligned)
    return _roi_align_values(inp, rois, scale, out_size, sampling_ratio, aligned)


def _roi_align_values(inp, rois, scale, out_size, sampling_ratio=0, aligned=True):
    x = np.ascontiguousarray(inp.detach().numpy(), dtype=np.float32)
    r = np.ascontiguousarray(rois.detach().numpy(), dtype=np.float32)
    ph, pw = (out_size, out_size) if isinstance(out_size, int) else out_size
    N, C, H, W = x.shape
    out = np.zeros((r.shape[0], C, ph, pw), np.float32)
    lib().oracle_roi_align_forward(_fp(x), N, C, H, W, _fp(r), r.shape[0], ctypes.c_float(scale),
                                   ph, pw, sampling_ratio, int(aligned), _fp(out))
    return torch.from_numpy(out)


def roi_align_backward(grad_out, rois, scale, in_shape, sampling_ratio=0, aligned=True):
    g = np.ascontiguousarray(grad_out.detach().numpy(), dtype=np.float32)
    r = np.ascontiguousarray(rois.detach().numpy(), dtype=np.float32)
    N, C, H, W = in_shape
    ph, pw = g.shape[2], g.shape[3]
    gi = np.zeros(in_shape, np.float32)
    lib().oracle_roi_align_backward(_fp(g), N, C, H, W, _fp(r), r.shape[0], ctypes.c_float(scale),
                                    ph, pw, sampling_ratio, int(aligned), _fp(gi))
    return torch.from_numpy(gi)


def assign_boxes_to_levels(box_lists, min_level, max_level, canonical_box_size=224, canonical_level=4):
    """poolers.py:22-58.  box_lists: list of (n,4) tensors."""
    areas = torch.cat([(b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) for b in box_lists])
    sizes = torch.sqrt(areas)
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
    return torch.clamp(lv, min=min_level, max=max_level).to(torch.int64) - min_level


def pooler_format(box_lists):
    """poolers.py:61-90."""
    boxes = torch.cat(box_lists, 0)
    idx = torch.cat([torch.full((len(b),), i, dtype=boxes.dtype) for i, b in enumerate(box_lists)])
    return torch.cat([idx[:, None], boxes], 1)


def roi_pooler(feats, box_lists, out_size, scales, sampling_ratio=0):
    """poolers.py:185-245 (ROIAlignV2).  feats: list of NCHW per level."""
    rois = pooler_format(box_lists)
    if len(feats) == 1:
        return roi_align(feats[0], rois, scales[0], out_size, sampling_ratio, True)
    min_level = int(-math.log2(scales[0]))
    max_level = int(-math.log2(scales[-1]))
    lv = assign_boxes_to_levels(box_lists, min_level, max_level)
    out = torch.zeros(rois.shape[0], feats[0].shape[1], out_size, out_size)
    for l, (f, s) in enumerate(zip(feats, scales)):
        inds = torch.nonzero(lv == l).squeeze(1)
        out[inds] = roi_align(f, rois[inds], s, out_size, sampling_ratio, True)
    return out


def crop_and_resize(masks, boxes, mask_size):
    """masks (n,H,W) bool/uint8, boxes (n,4) -> (n,S,S) bool.  masks.py:189-220."""
    n = len(boxes)
    rois = torch.cat([torch.arange(n, dtype=boxes.dtype)[:, None], boxes], 1)
    out = roi_align(masks.to(torch.float32)[:, None], rois, 1.0, mask_size, 0, True).squeeze(1)
    return out >= 0.5


def nms(boxes, scores, thr):
    """Greedy NMS, keep sorted by descending score (torchvision.ops.nms semantics)."""
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    b = np.ascontiguousarray(boxes.detach().float().numpy())
    order = np.ascontiguousarray(torch.sort(scores.detach(), descending=True, stable=True)[1].numpy())
    keep = np.zeros(len(order), np.int64)
    n = lib().oracle_nms(_fp(b), _fp(order), ctypes.c_int64(len(order)), ctypes.c_float(thr), _fp(keep))
    return torch.from_numpy(keep[:n].copy())


def batched_nms(boxes, scores, idxs, thr):
    """D2/layers/nms.py:9-20.  torchvision picks the coordinate-offset trick or a per-class loop
    by a size heuristic; both give the same keep set.  Per-class loop restated here (exact IoU)."""
    boxes = boxes.float()
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for c in torch.unique(idxs):
        ci = torch.nonzero(idxs == c).squeeze(1)
        keep_mask[ci[nms(boxes[ci], scores[ci], thr)]] = True
    ki = torch.nonzero(keep_mask).squeeze(1)
    return ki[torch.sort(scores[ki], descending=True, stable=True)[1]]


# ------------------------------------------------------------------ box utilities
def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def pairwise_iou(b1, b2):
    """boxes.py:310-357."""
    wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    a1, a2 = box_area(b1), box_area(b2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype))


def matcher(iou, thresholds, labels):
    """matcher.py:62-104 (no low-quality matches).  iou (M gt, N pred)."""
    if iou.numel() == 0:
        n = iou.shape[1]
        return torch.zeros(n, dtype=torch.int64), torch.full((n,), labels[0], dtype=torch.int8)
    vals, idx = iou.max(dim=0)
    th = [-float("inf")] + list(thresholds) + [float("inf")]
    lab = torch.ones_like(idx, dtype=torch.int8)
    for l, lo, hi in zip(labels, th[:-1], th[1:]):
        lab[(vals >= lo) & (vals < hi)] = l
    return idx, lab


def subsample_labels(labels, num_samples, positive_fraction, bg_label):
    """sampling.py:9-54 -- consumes two torch.randperm draws from the global CPU generator."""
    pos = torch.nonzero((labels != -1) & (labels != bg_label)).squeeze(1)
    neg = torch.nonzero(labels == bg_label).squeeze(1)
    num_pos = min(pos.numel(), int(num_samples * positive_fraction))
    num_neg = min(neg.numel(), num_samples - num_pos)
    p1 = torch.randperm(pos.numel())[:num_pos]
    p2 = torch.randperm(neg.numel())[:num_neg]
    return pos[p1], neg[p2]


SCALE_CLAMP = math.log(1000.0 / 16)


def get_deltas(src, tgt, weights):
    """box_regression.py:43-76."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack([wx * (tx - sx) / sw, wy * (ty - sy) / sh,
                        ww * torch.log(tw / sw), wh * torch.log(th / sh)], 1)


def apply_deltas(deltas, boxes, weights):
    """box_regression.py:78-118 (class-agnostic: deltas (N,4))."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx, dy = deltas[:, 0] / wx, deltas[:, 1] / wy
    dw = torch.clamp(deltas[:, 2] / ww, max=SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3] / wh, max=SCALE_CLAMP)
    pcx, pcy = dx * w + cx, dy * h + cy
    pw, phh = torch.exp(dw) * w, torch.exp(dh) * h
    return torch.stack([pcx - 0.5 * pw, pcy - 0.5 * phh, pcx + 0.5 * pw, pcy + 0.5 * phh], 1)
