"""Instance copy-paste compositor ('basic' blend) on the GPU.
Reference: DG/divergen/data/custom_build_copypaste_mapper.py:488-566, :79-92."""
import numpy as np
import torch

from .. import _lib as L
from ..utils.h2d import upload_i32


class PackedPastes:
    """The K paste patches of one image as the kernel takes them: ONE flat uint8 buffer (each RGBA patch padded to 4 bytes) + the
    (K, 5) int32 descriptors (byte offset, h, w, x0, y0) + the K labels, all on the device."""

    def __init__(self, flat, desc, labels, K):
        self.flat, self.desc, self.labels, self.K = flat, desc, labels, K

    def __len__(self):
        return self.K


def pack_pastes_host(pastes):
    """list of (rgba uint8 (h, w, 4) numpy, x0, y0, label) -> (flat uint8, desc int32 (K, 5), labels int64 (K)) as CPU tensors:
    what a LOADER WORKER hands the training process for one image (no device, no libdgx).  K = 0 gives a 4-byte flat buffer."""
    desc, off = [], 0
    for rgba, x0, y0, _ in pastes:
        h, w = int(rgba.shape[0]), int(rgba.shape[1])
        n = h * w * 4
        desc.append([off, h, w, int(x0), int(y0)])
        off += n + ((-n) % 4)
    host = np.zeros(max(off, 4), dtype=np.uint8)
    for (rgba, _, _, _), d in zip(pastes, desc):
        n = d[1] * d[2] * 4
        a = rgba.detach().cpu().numpy() if isinstance(rgba, torch.Tensor) else np.asarray(rgba)
        host[d[0]:d[0] + n] = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
    labels = np.array([int(np.asarray(p[3]).reshape(-1)[0]) for p in pastes], dtype=np.int64)
    return torch.from_numpy(host), torch.from_numpy(np.asarray(desc, dtype=np.int32).reshape(-1, 5)), torch.from_numpy(labels)


def pack_pastes(pastes, device):
    """list of (rgba uint8 (h, w, 4) numpy | tensor, x0, y0, label) -> PackedPastes.  Host arrays (the loader's case: patches come
    out of the instance pool in host memory) are laid out in one host buffer and go up in ONE copy; patches that already live on
    the device are gathered with one concatenation.  Round 2 concatenated 2 K device chunks per image inside every step, which
    torch executes as one hipMemcpyAsync per chunk: 38 blit launches of ~10 us per image (0.8 ms per step on the loader stream)."""
    K = len(pastes)
    on_dev = [isinstance(r, torch.Tensor) and r.is_cuda for r, _, _, _ in pastes]
    if K and all(on_dev):
        desc, off, chunks = [], 0, []
        for rgba, x0, y0, _ in pastes:
            h, w = int(rgba.shape[0]), int(rgba.shape[1])
            n = h * w * 4
            desc.append([off, h, w, int(x0), int(y0)])
            off += n + ((-n) % 4)
            chunks.append(rgba.reshape(-1))
            if (-n) % 4:
                chunks.append(torch.zeros((-n) % 4, dtype=torch.uint8, device=device))
        flat = torch.cat(chunks)
        desc_t = upload_i32(desc, device).view(-1, 5)
        labels = upload_i32([int(np.asarray(p[3]).reshape(-1)[0]) for p in pastes], device).long()
        return PackedPastes(flat, desc_t, labels, K)
    flat, desc, labels = pack_pastes_host(pastes)
    if torch.device(device).type != "cuda":
        return PackedPastes(flat, desc, labels, K)
    flat = flat.pin_memory().to(device, non_blocking=True)
    desc_t = upload_i32(desc.numpy(), device).view(-1, 5) if K else torch.zeros(0, 5, dtype=torch.int32, device=device)
    labels = upload_i32(labels.numpy(), device).long() if K else torch.zeros(0, dtype=torch.int64, device=device)
    return PackedPastes(flat, desc_t, labels, K)


def copy_paste(image, masks, boxes, labels, pastes, lazy_masks=False):
    """image uint8 (3,H,W), masks uint8 (n,H,W), boxes f32 (n,4), labels i64 (n) -- GPU tensors.
    pastes: list of (rgba uint8 numpy/tensor (h,w,4), x0, y0, label) applied in order, or a PackedPastes (pack_pastes).
    Returns dict(image, masks, boxes, labels, source) exactly like the sequential reference.  Mask bytes pass through (0/1 in,
    0/1 out).  lazy_masks: `masks` holds ALL n + K objects' rows and `keep` (i64) the rows of the surviving ones, in order --
    what structures.BitMasks(masks, index=keep) takes: the full-resolution rows of the survivors are then never gathered
    (the only consumer, crop_and_resize, reads a few rows through the index)."""
    K = len(pastes)
    dev = image.device
    n0, H, W = masks.shape[0], image.shape[1], image.shape[2]
    if K == 0:
        out = dict(image=image, masks=masks, boxes=boxes, labels=labels, source=torch.zeros(n0, dtype=torch.int64, device=dev))
        if lazy_masks:
            out["keep"] = torch.arange(n0, dtype=torch.int64, device=dev)
        return out
    pk = pastes if isinstance(pastes, PackedPastes) else pack_pastes(pastes, dev)
    flat, desc_t = pk.flat, pk.desc
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
    keep = out_valid.nonzero().squeeze(1)          # ONE compaction (and one device->host count) for the four per-object tensors
    all_labels = torch.cat([labels.to(torch.int64), pk.labels])
    source = torch.cat([torch.zeros(n0, dtype=torch.int64, device=dev), torch.ones(K, dtype=torch.int64, device=dev)])
    out = dict(image=image, boxes=out_boxes.index_select(0, keep), labels=all_labels.index_select(0, keep), source=source.index_select(0, keep))
    if lazy_masks:
        out["masks"], out["keep"] = out_masks, keep
    else:
        out["masks"] = out_masks.index_select(0, keep)
    return out
