"""CenterNet dense target assignment.  Reference: CN/modeling/dense_heads/centernet.py:338-436."""
import ctypes

import torch

from .. import _lib as L
from ..utils.h2d import upload_i32


def centernet_targets(gt_boxes_list, level_hw, strides, soi, hm_min_overlap=0.8, min_radius=4, label_inds=False):
    """gt_boxes_list: per-image (n_i,4) GPU tensors.  -> reg_targets (M*B,4), heatmap (M*B,1),
    level-major layout (level, image, y, x) exactly as the reference's _get_ground_truth.
    label_inds=True: also ((sum n_i * L,) i64 indices, bool cared) of `_get_label_inds` (dgx_centernet_label_inds)."""
    B = len(gt_boxes_list)
    dev = gt_boxes_list[0].device
    counts = [int(b.shape[0]) for b in gt_boxes_list]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    gt = torch.cat([b.float().reshape(-1, 4) for b in gt_boxes_list]).contiguous() if offs[-1] else \
        torch.zeros(1, 4, dtype=torch.float32, device=dev)
    offs_t = upload_i32(offs, dev)          # pinned + async: no stream stall for B+1 integers
    Lv = len(strides)
    M = sum(h * w for h, w in level_hw)
    reg = torch.empty(M * B, 4, dtype=torch.float32, device=dev)
    hm = torch.empty(M * B, 1, dtype=torch.float32, device=dev)
    hw = (ctypes.c_int32 * (2 * Lv))(*[v for p in level_hw for v in p])
    st = (ctypes.c_int32 * Lv)(*strides)
    so = (ctypes.c_float * (2 * Lv))(*[float(v) for p in soi for v in p])
    delta = (1 - hm_min_overlap) / (1 + hm_min_overlap)
    L.check(L.lib().dgx_centernet_targets(L.ptr(gt), L.ptr(offs_t), B, hw, st, so, Lv, delta, float(min_radius),
                                          L.ptr(reg), L.ptr(hm), L.stream()), "dgx_centernet_targets")
    if not label_inds:
        return reg, hm
    tot = offs[-1]
    ind = torch.empty(tot * Lv, dtype=torch.int64, device=dev)
    cared = torch.empty(tot * Lv, dtype=torch.uint8, device=dev)
    L.check(L.lib().dgx_centernet_label_inds(L.ptr(gt), L.ptr(offs_t), B, tot, hw, st, so, Lv, L.ptr(ind), L.ptr(cared), L.stream()),
            "dgx_centernet_label_inds")
    return reg, hm, (ind, cared.view(torch.bool))
