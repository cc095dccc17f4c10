"""ROIAlign / ROIPooler / GT-mask crop over channels-last features.
Reference: D2/layers/roi_align.py:7-65, D2/modeling/poolers.py:22-245, D2/structures/masks.py:189-220."""
import ctypes
import math
import os

import torch

from .. import _lib as L


# the atomic-scatter backward stays for the parity test of the two forms (tests/test_gpu_kernels.py sets _GATHER)
_GATHER = True


def _nhwc(feat):
    """(N,C,H,W) logical tensor -> physical (N,H,W,C) contiguous view for the kernels."""
    return feat.permute(0, 2, 3, 1).contiguous()


class FeatureGradients:
    """Gradient maps of a list of feature tensors that their consumers' backward passes fill IN PLACE, instead of returning one
    map each for autograd to add up: `maps[i]` is None or the (N,C,H,W) gradient of feature i so far.  The FPN levels have up to
    five consumers per step (CenterNet head, three cascade stages, mask head); at 1024^2 one addition of a P3-sized pair of maps
    moves 50 MB.  Whoever owns the features joins the maps back into autograd after the last consumer ran
    (modeling/meta_arch/custom_rcnn.py _JoinGradients); a feature tensor announces its slot as `_dgx_grad_sink = (self, i)`."""

    def __init__(self, n):
        self.maps = [None] * n

    def add(self, i, g):
        """The FIRST gradient of a slot is kept WITHOUT a copy and later ones (and the poolers' gather kernels) add into it in place.
        Invariant this relies on: the tensor autograd hands over is owned by this pass alone.  It holds for an eager node's
        fresh output and for the graphed CenterNet head's static grad-input buffer -- the next replay rewrites that buffer and nothing else
        reads it in between (utils/graphs.py ALIAS_STATIC hand-over buffers are OUTPUTS, not gradient inputs).  Autograd itself never
        hands the same gradient tensor to two nodes without the second one being allowed to modify it only if it owns it: nodes that
        keep a gradient they were given must not write to it -- this class is the one exception and is only fed by nodes of this package
        (_CaptureGradient, the RoI poolers).  (Cloning views defensively would copy every NHWC-stored map: 50 MB per level and consumer.)"""
        m = self.maps[i]
        if m is None:
            self.maps[i] = g
        else:
            m.add_(g)

    def take(self, i):
        m, self.maps[i] = self.maps[i], None
        return m


def _sink_of(feats):
    """(FeatureGradients, slots) when every feature announces a slot of the SAME object, else None."""
    sinks = [getattr(f, "_dgx_grad_sink", None) for f in feats]
    if any(s is None for s in sinks) or any(s[0] is not sinks[0][0] for s in sinks):
        return None
    return sinks[0][0], [s[1] for s in sinks]


class _ROIPooler(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, out_size, min_level, sampling_ratio, out_nhwc, aligned, scale0, sink, grad_scale, *feats):
        ctx.sink, ctx.grad_scale = sink, float(grad_scale)
        nl = len(feats)
        f0 = feats[0]
        N, C = f0.shape[0], f0.shape[1]
        phys = [_nhwc(f) for f in feats]
        R = rois.shape[0]
        rois = rois.float().contiguous()
        ph = pw = out_size
        # the kernel always writes channels-last (8 channels per lane, 16-byte taps); a (C, ph, pw)-ordered result
        # (what the box head's first FC expects) is one transposing copy of the small pooled tensor
        vec = C % 8 == 0
        if out_nhwc or vec:
            out = torch.empty(R, ph, pw, C, dtype=f0.dtype, device=f0.device).permute(0, 3, 1, 2)
            out_phys = out.permute(0, 2, 3, 1)
        else:
            out = torch.empty(R, C, ph, pw, dtype=f0.dtype, device=f0.device)
            out_phys = out
        k_nhwc = out_nhwc or vec
        Hs = (ctypes.c_int * nl)(*[f.shape[2] for f in feats])
        Ws = (ctypes.c_int * nl)(*[f.shape[3] for f in feats])
        if nl == 1:
            L.check(L.lib().dgx_roi_align_fwd(L.ptr(phys[0]), L.ptr(rois), L.ptr(out_phys), N, Hs[0], Ws[0], C, R,
                                              scale0, ph, pw, sampling_ratio, int(aligned), int(k_nhwc),
                                              L.dtype_code(f0), L.stream()), "dgx_roi_align_fwd")
        else:
            ptrs = (ctypes.c_void_p * nl)(*[L.ptr(p) for p in phys])
            L.check(L.lib().dgx_roi_pooler_fwd(ptrs, Hs, Ws, nl, min_level, L.ptr(rois), L.ptr(out_phys), None, N, C,
                                               R, ph, pw, sampling_ratio, int(k_nhwc), L.dtype_code(f0), L.stream()),
                    "dgx_roi_pooler_fwd")
        if k_nhwc and not out_nhwc:
            out = out.contiguous()
        ctx.save_for_backward(rois)
        ctx.cfg = (out_size, min_level, sampling_ratio, k_nhwc, aligned, scale0, N, C,
                   [tuple(f.shape) for f in feats], f0.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        (rois,) = ctx.saved_tensors
        out_size, min_level, sampling_ratio, out_nhwc, aligned, scale0, N, C, shapes, dt = ctx.cfg
        nl = len(shapes)
        R = rois.shape[0]
        g_phys = gout.permute(0, 2, 3, 1).contiguous() if out_nhwc else gout.contiguous()
        Hs = (ctypes.c_int * nl)(*[s[2] for s in shapes])
        Ws = (ctypes.c_int * nl)(*[s[3] for s in shapes])
        if out_nhwc and _GATHER and C % 8 == 0 and C >= 64 and 256 % (C // 8) == 0 and out_size <= 16:
            # output-stationary gather: every gradient pixel written once, in the feature dtype (no fp32 staging maps)
            accumulate = 0
            if ctx.sink is not None:
                # the maps other consumers of these features already wrote (channels-last storage, this dtype) take the sum in place
                fg, slots = ctx.sink
                grads = []
                for s, i in zip(shapes, slots):
                    m = fg.maps[i]
                    if m is not None and not (m.dtype == g_phys.dtype and tuple(m.shape) == tuple(s) and m.permute(0, 2, 3, 1).is_contiguous()
                                              and m.data_ptr() % 16 == 0):
                        m = m.to(g_phys.dtype).contiguous(memory_format=torch.channels_last)
                    grads.append(None if m is None else m.permute(0, 2, 3, 1))
                accumulate = int(any(g is not None for g in grads))
                mk = torch.zeros if accumulate else torch.empty
                grads = [mk(s[0], s[2], s[3], s[1], dtype=g_phys.dtype, device=gout.device) if g is None else g for g, s in zip(grads, shapes)]
            else:
                grads = [torch.empty(s[0], s[2], s[3], s[1], dtype=g_phys.dtype, device=gout.device) for s in shapes]
            ptrs = (ctypes.c_void_p * nl)(*[L.ptr(g) for g in grads])
            L.check(L.lib().dgx_roi_pooler_bwd_gather_accum(L.ptr(g_phys), ptrs, Hs, Ws, nl, min_level, scale0, int(aligned),
                                                            L.ptr(rois), N, C, R, out_size, out_size, sampling_ratio, accumulate,
                                                            ctx.grad_scale, L.dtype_code(g_phys), L.stream()), "dgx_roi_pooler_bwd_gather_accum")
            if ctx.sink is not None:
                for g, i in zip(grads, slots):
                    fg.maps[i] = g.permute(0, 3, 1, 2)
                return (None,) * (9 + nl)
            return (None,) * 9 + tuple(g.permute(0, 3, 1, 2).to(dt) for g in grads)
        if ctx.grad_scale != 1.0:
            g_phys = g_phys * ctx.grad_scale
        grads = [torch.zeros(s[0], s[2], s[3], s[1], dtype=torch.float32, device=gout.device) for s in shapes]
        if nl == 1:
            L.check(L.lib().dgx_roi_align_bwd(L.ptr(g_phys), L.ptr(rois), L.ptr(grads[0]), N, Hs[0], Ws[0], C, R,
                                              scale0, out_size, out_size, sampling_ratio, int(aligned), int(out_nhwc),
                                              L.dtype_code(g_phys), L.stream()), "dgx_roi_align_bwd")
        else:
            ptrs = (ctypes.c_void_p * nl)(*[L.ptr(g) for g in grads])
            L.check(L.lib().dgx_roi_pooler_bwd(L.ptr(g_phys), ptrs, Hs, Ws, nl, min_level, L.ptr(rois), N, C, R,
                                               out_size, out_size, sampling_ratio, int(out_nhwc),
                                               L.dtype_code(g_phys), L.stream()), "dgx_roi_pooler_bwd")
        outs = [g.permute(0, 3, 1, 2).to(dt) for g in grads]
        if ctx.sink is not None:
            for o, i in zip(outs, ctx.sink[1]):
                ctx.sink[0].add(i, o)
            return (None,) * (9 + nl)
        return (None,) * 9 + tuple(outs)


def roi_align(feat, rois, spatial_scale, out_size, sampling_ratio=0, aligned=True, out_nhwc=False, grad_scale=1.0):
    """feat (N,C,H,W) (any memory format), rois (R,5) -> (R,C,S,S)."""
    return _ROIPooler.apply(rois, out_size, 0, sampling_ratio, out_nhwc, aligned, float(spatial_scale), _sink_of([feat]), grad_scale, feat)


def roi_pooler(feats, rois, out_size, scales, sampling_ratio=0, out_nhwc=False, grad_scale=1.0):
    """Multi-level ROIAlignV2 (poolers.py:185-245).  feats: list of (N,C,H,W); rois (R,5) with
    batch index in column 0; scales: per-level spatial scales (powers of two).
    grad_scale: the factor of a _ScaleGradient (cascade_rcnn.py:20-28) placed on the pooled features, applied inside the backward."""
    min_level = int(round(-math.log2(scales[0])))
    if len(feats) == 1:
        return roi_align(feats[0], rois, scales[0], out_size, sampling_ratio, True, out_nhwc, grad_scale)
    return _ROIPooler.apply(rois, out_size, min_level, sampling_ratio, out_nhwc, True, float(scales[0]), _sink_of(feats), grad_scale, *feats)


def mask_crop(masks, boxes, mask_idx, size):
    """masks uint8/bool (M,H,W) on GPU, boxes (R,4), mask_idx (R) -> bool (R,size,size).
    BitMasks.crop_and_resize (masks.py:189-220) without the fp32 mask copy."""
    m = masks.view(torch.uint8) if masks.dtype == torch.bool else masks
    m = m.contiguous()
    boxes = boxes.float().contiguous()
    idx = mask_idx.to(torch.int32).contiguous()
    R = boxes.shape[0]
    out = torch.empty(R, size, size, dtype=torch.uint8, device=boxes.device)
    L.check(L.lib().dgx_mask_crop(L.ptr(m), L.ptr(boxes), L.ptr(idx), L.ptr(out), m.shape[0], m.shape[1], m.shape[2],
                                  R, size, L.stream()), "dgx_mask_crop")
    return out.view(torch.bool)
