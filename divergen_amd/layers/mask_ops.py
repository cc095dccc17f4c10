"""Evaluation post-processing on the GPU: `paste_masks_in_image` (D2/layers/mask_ops.py:73-150) and the
run-length encoding the results writer needs (D2/evaluation/coco_evaluation.py:380-420 instances_to_coco_json), through
libdgx's dgx_paste_masks / dgx_paste_rle.  No torch fallback: a missing extension raises in `_lib.lib()`."""
import ctypes

import numpy as np
import torch

from .. import _lib


def _prep(masks, boxes):
    assert masks.dim() == 3 and masks.shape[-1] == masks.shape[-2], "Only square mask predictions are supported"
    assert boxes.shape == (masks.shape[0], 4), boxes.shape
    return masks.float().contiguous(), boxes.float().contiguous()


def paste_masks_in_image(masks, boxes, image_shape, threshold=0.5):
    """(N,S,S) probabilities + (N,4) boxes -> (N,H,W) bool, the reference's signature and result."""
    H, W = int(image_shape[0]), int(image_shape[1])
    N = masks.shape[0]
    if N == 0:
        return masks.new_empty((0, H, W), dtype=torch.bool)
    if hasattr(boxes, "tensor"):
        boxes = boxes.tensor
    m, b = _prep(masks, boxes)
    out = torch.empty((N, H, W), dtype=torch.uint8, device=m.device)
    _lib.check(_lib.lib().dgx_paste_masks(_lib.ptr(m), _lib.ptr(b), _lib.ptr(out), N, m.shape[-1], H, W, float(threshold),
                                          _lib.stream()), "dgx_paste_masks")
    return out.view(torch.bool)


def paste_masks_rle_counts(masks, boxes, image_shape, threshold=0.5, cap=4096):
    """Run lengths of the pasted masks without materialising them: (counts i32 (N,cap) on the device, nruns i32 (N))."""
    H, W = int(image_shape[0]), int(image_shape[1])
    N = masks.shape[0]
    if hasattr(boxes, "tensor"):
        boxes = boxes.tensor
    m, b = _prep(masks, boxes)
    L = _lib.lib()
    return _grow_until_fits(
        lambda c, r, cp: _lib.check(L.dgx_paste_rle(_lib.ptr(m), _lib.ptr(b), _lib.ptr(c), _lib.ptr(r), N, m.shape[-1], H, W,
                                                    float(threshold), cp, _lib.stream()), "dgx_paste_rle"), N, cap, m.device)


def _grow_until_fits(launch, N, cap, device):
    while True:
        counts = torch.empty((N, cap), dtype=torch.int32, device=device)
        nruns = torch.empty((N,), dtype=torch.int32, device=device)
        if N:
            launch(counts, nruns, cap)
        need = int(-nruns.min()) if N else 0          # results writer: a host sync here is the point of the call
        if need <= 0:
            return counts, nruns
        cap = 1 << (need - 1).bit_length()


def _to_rle_dicts(counts, nruns, H, W):
    nr = nruns.cpu().numpy()
    width = int(nr.max()) if len(nr) else 0
    host = np.ascontiguousarray(counts[:, :width].cpu().numpy())
    L = _lib.lib()
    out = []
    buf = ctypes.create_string_buffer(max(16, 7 * width))
    bp = ctypes.cast(buf, ctypes.c_void_p)
    for n in range(len(nr)):
        ln = L.dgx_rle_to_string(host[n].ctypes.data, int(nr[n]), bp, len(buf))
        assert ln >= 0
        out.append({"size": [H, W], "counts": buf.raw[:ln]})
    return out


def paste_masks_rle(masks, boxes, image_shape, threshold=0.5):
    """List of COCO RLE dicts {"size": [H, W], "counts": bytes}, what `mask_util.encode` returns per detection."""
    H, W = int(image_shape[0]), int(image_shape[1])
    counts, nruns = paste_masks_rle_counts(masks, boxes, image_shape, threshold)
    return _to_rle_dicts(counts, nruns, H, W)


def rle_encode_bitmasks(bits, cap=4096):
    """(N,H,W) bool/uint8 device bitmasks -> list of COCO RLE dicts (dgx_rle_encode)."""
    N, H, W = bits.shape
    b = (bits.view(torch.uint8) if bits.dtype == torch.bool else bits.to(torch.uint8)).contiguous()
    L = _lib.lib()
    counts, nruns = _grow_until_fits(
        lambda c, r, cp: _lib.check(L.dgx_rle_encode(_lib.ptr(b), _lib.ptr(c), _lib.ptr(r), N, H, W, cp, _lib.stream()),
                                    "dgx_rle_encode"), N, cap, b.device)
    return _to_rle_dicts(counts, nruns, H, W)
