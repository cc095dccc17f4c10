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
