"""NMS / batched NMS / IoU+Matcher.  Reference: D2/layers/nms.py:9-20 (torchvision semantics),
D2/structures/boxes.py:310-357, D2/modeling/matcher.py:62-104."""
import torch

from .. import _lib as L


def nms(boxes, scores, iou_threshold):
    """Greedy NMS; returns kept indices sorted by descending score (torchvision.ops.nms)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True)[1]
    sb = boxes.float()[order].contiguous()
    words = L.lib().dgx_nms_workspace_words(n)
    ws = torch.empty(words, dtype=torch.int64, device=boxes.device)
    keep = torch.empty(n, dtype=torch.uint8, device=boxes.device)
    cnt = torch.empty(1, dtype=torch.int32, device=boxes.device)
    L.check(L.lib().dgx_nms_sorted(L.ptr(sb), n, float(iou_threshold), L.ptr(ws), L.ptr(keep), L.ptr(cnt),
                                   L.stream()), "dgx_nms_sorted")
    return order[keep.bool()]


def nms_batched_sorted(boxes, scores, n_valid, iou_threshold, max_keep=0, cap=None):
    """Greedy NMS for B images at once without a host round trip.
    boxes (B,K,4) f32 sorted by descending score per image, scores (B,K) f32 (sorted) or None,
    n_valid (B) int32 on the device.  Returns keep_idx (B,cap) int32 (-1 padded, score order) and
    num_keep (B) int32.  max_keep > 0 stops at that many kept boxes (+ score ties, as the reference's
    ">= k-th score" rule)."""
    B, K, _ = boxes.shape
    cap = cap or (max_keep + 64 if max_keep > 0 else K)
    boxes = boxes.float().contiguous()
    lib = L.lib()
    ws = torch.empty(max(int(lib.dgx_nms_batched_workspace_words(B, K)), 1), dtype=torch.int64, device=boxes.device)
    keep_idx = torch.empty(B, cap, dtype=torch.int32, device=boxes.device)
    num_keep = torch.empty(B, dtype=torch.int32, device=boxes.device)
    sc = scores.float().contiguous() if scores is not None else None
    L.check(lib.dgx_nms_batched(L.ptr(boxes), L.ptr(sc), L.ptr(n_valid.to(torch.int32).contiguous()), B, K, float(iou_threshold),
                                int(max_keep), L.ptr(ws), L.ptr(keep_idx), cap, L.ptr(num_keep), L.stream()), "dgx_nms_batched")
    return keep_idx, num_keep


def batched_nms(boxes, scores, idxs, iou_threshold):
    """Per-class NMS via the coordinate-offset strategy (one of torchvision's two, same keep set)."""
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    boxes = boxes.float()
    if idxs is None or bool((idxs == idxs[0]).all()):
        return nms(boxes, scores, iou_threshold)
    off = idxs.to(boxes) * (boxes.max() + 1)
    return nms(boxes + off[:, None], scores, iou_threshold)


def iou_match(gt_boxes, proposal_boxes, threshold, return_iou=False):
    """-> matched_idx int64 (N), matched_label int8 (N) [, max_iou f32 (N)]."""
    gt = gt_boxes.float().contiguous()
    pr = proposal_boxes.float().contiguous()
    N = pr.shape[0]
    idx = torch.empty(N, dtype=torch.int64, device=pr.device)
    lab = torch.empty(N, dtype=torch.int8, device=pr.device)
    miou = torch.empty(N, dtype=torch.float32, device=pr.device) if return_iou else None
    L.check(L.lib().dgx_iou_match(L.ptr(gt) if gt.shape[0] else None, gt.shape[0], L.ptr(pr), N, float(threshold),
                                  L.ptr(idx), L.ptr(lab), L.ptr(miou), L.stream()), "dgx_iou_match")
    return (idx, lab, miou) if return_iou else (idx, lab)
