"""Evaluation post-processing on the GPU: `paste_masks_in_image` (D2/layers/mask_ops.py:73-150) and the
run-length encoding the results writer needs (D2/evaluation/coco_evaluation.py:380-420 instances_to_coco_json), through
libdgx's dgx_paste_masks / dgx_paste_rle.  No torch fallback: a missing extension raises in `_lib.lib()`."""
import ctypes

import numpy as np
import torch

from .. import _lib
from .. import _lib as L


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


class _MaskBCE(torch.autograd.Function):
    """mean BCE-with-logits over (R, S, S) mask logits + the three mask statistics, one dgx_mask_bce call
    (mask_head.py:35-110); the gradient is produced by the same pass."""

    @staticmethod
    def forward(ctx, pred, gt_u8):
        R = pred.shape[0]
        inner = pred[0].numel() if R else 1
        if pred.dim() > 1 and R and not pred[0].is_contiguous():
            pred = pred.contiguous()
        n = pred.numel()
        row_stride = pred.stride(0) if R else inner
        out = torch.empty(5, dtype=torch.float32, device=pred.device)
        ws = torch.empty(max(int(L.lib().dgx_mask_bce_workspace_floats(n)), 8), dtype=torch.float32, device=pred.device)
        grad = torch.empty(pred.shape, dtype=pred.dtype, device=pred.device) if ctx.needs_input_grad[0] else None
        L.check(L.lib().dgx_mask_bce(pred.data_ptr() if n else None, row_stride, inner, L.ptr(gt_u8), n, L.ptr(grad), L.ptr(out), L.ptr(ws),
                                     L.dtype_code(pred), L.stream()), "dgx_mask_bce")
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gloss, _gstats):
        (grad,) = ctx.saved_tensors
        return (grad * gloss.to(grad.dtype)) if grad is not None else None, None


def mask_bce_with_stats(pred, gt_bool):
    """pred (R, S, S) f32 / bf16 logits (dim 0 may be strided), gt_bool (R, S, S) bool / uint8 ->
    (mean BCE loss, stats f32 (5) = [loss, #incorrect, #false positive, #false negative, #positive])."""
    gt = gt_bool.view(torch.uint8) if gt_bool.dtype == torch.bool else gt_bool
    return _MaskBCE.apply(pred, gt.contiguous())
