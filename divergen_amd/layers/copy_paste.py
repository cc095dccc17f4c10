"""Instance copy-paste compositor ('basic' blend) on the GPU.
Reference: DG/divergen/data/custom_build_copypaste_mapper.py:488-566, :79-92."""
import numpy as np
import torch

from .. import _lib as L
from ..utils.h2d import upload_i32


def copy_paste(image, masks, boxes, labels, pastes):
    """image uint8 (3,H,W), masks uint8 (n,H,W), boxes f32 (n,4), labels i64 (n) -- GPU tensors.
    pastes: list of (rgba uint8 numpy/tensor (h,w,4), x0, y0, label) applied in order.
    Returns dict(image, masks, boxes, labels, source) exactly like the sequential reference."""
    K = len(pastes)
    dev = image.device
    n0, H, W = masks.shape[0], image.shape[1], image.shape[2]
    if K == 0:
        return dict(image=image, masks=masks, boxes=boxes, labels=labels,
                    source=torch.zeros(n0, dtype=torch.int64, device=dev))
    desc, chunks, off = [], [], 0
    for rgba, x0, y0, _ in pastes:
        a = rgba if isinstance(rgba, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(rgba))
        h, w = int(a.shape[0]), int(a.shape[1])
        a = a.to(dev, non_blocking=True).reshape(-1)
        desc.append([off, h, w, int(x0), int(y0)])
        chunks.append(a)
        pad = (-a.numel()) % 4
        if pad:
            chunks.append(torch.zeros(pad, dtype=torch.uint8, device=dev))
        off += a.numel() + pad
    flat = torch.cat(chunks)
    desc_t = upload_i32(desc, dev).view(-1, 5)
    image = image.contiguous().clone()
    masks = masks.contiguous()
    boxes0 = boxes.float().contiguous()
    nobj = n0 + K
    out_masks = torch.empty(nobj, H, W, dtype=torch.uint8, device=dev)
    out_boxes = torch.empty(nobj, 4, dtype=torch.float32, device=dev)
    out_valid = torch.empty(nobj, dtype=torch.uint8, device=dev)
    stats = torch.empty(nobj * (K + 1) * 5 + 3 + H * W, dtype=torch.int32, device=dev)
    L.check(L.lib().dgx_copy_paste(L.ptr(image), L.ptr(masks) if n0 else None, L.ptr(boxes0) if n0 else None, n0, H, W,
                                   L.ptr(flat), L.ptr(desc_t), K, L.ptr(out_masks), L.ptr(out_boxes), L.ptr(out_valid),
                                   L.ptr(stats), L.stream()), "dgx_copy_paste")
    valid = out_valid.bool()
    all_labels = torch.cat([labels.to(torch.int64), upload_i32([int(np.asarray(p[3]).reshape(-1)[0]) for p in pastes], dev).long()])
    source = torch.cat([torch.zeros(n0, dtype=torch.int64, device=dev), torch.ones(K, dtype=torch.int64, device=dev)])
    return dict(image=image, masks=out_masks[valid], boxes=out_boxes[valid], labels=all_labels[valid],
                source=source[valid])
