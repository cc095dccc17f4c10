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


class _HeadOutputs(torch.autograd.Function):
    """Grouped predictor outputs of all levels (channels-last bf16, channel 0 = heat-map logit, 1..4 = regression) + the per-level
    scale parameters -> (reg (M, 4) f32, hm (M,) f32): dgx_centernet_head_outputs / _bwd, one launch each way."""

    @staticmethod
    def forward(ctx, n, *args):
        xs, scales = args[:n], args[n:]
        lib = L.lib()
        C = xs[0].shape[-1]
        rows = [x.shape[0] * x.shape[1] * x.shape[2] for x in xs]
        M = sum(rows)
        dev = xs[0].device
        reg = torch.empty(M, 4, dtype=torch.float32, device=dev)
        hm = torch.empty(M, dtype=torch.float32, device=dev)
        lv = (L.HeadLevel * n)()
        for i, (x, s) in enumerate(zip(xs, scales)):
            lv[i].x, lv[i].dx, lv[i].scale, lv[i].rows = L.ptr(x), None, L.ptr(s), rows[i]
        L.check(lib.dgx_centernet_head_outputs(lv, n, C, L.ptr(reg), L.ptr(hm), L.stream()), "dgx_centernet_head_outputs")
        ctx.save_for_backward(*xs, *scales)
        ctx.n = n
        return reg, hm

    @staticmethod
    def backward(ctx, g_reg, g_hm):
        n = ctx.n
        saved = ctx.saved_tensors
        xs, scales = saved[:n], saved[n:]
        lib = L.lib()
        C = xs[0].shape[-1]
        dev = xs[0].device
        g_reg = (g_reg if g_reg is not None else torch.zeros(sum(x.shape[0] * x.shape[1] * x.shape[2] for x in xs), 4, device=dev)).float().contiguous()
        g_hm = (g_hm if g_hm is not None else torch.zeros(g_reg.shape[0], device=dev)).float().contiguous()
        dxs = [torch.empty_like(x) for x in xs]
        lv = (L.HeadLevel * n)()
        for i, (x, dx, s) in enumerate(zip(xs, dxs, scales)):
            lv[i].x, lv[i].dx, lv[i].scale, lv[i].rows = L.ptr(x), L.ptr(dx), L.ptr(s), x.shape[0] * x.shape[1] * x.shape[2]
        ws = torch.empty(max(int(lib.dgx_centernet_head_outputs_bwd_workspace_floats(lv, n, C)), 1), dtype=torch.float32, device=dev)
        d_scale = torch.empty(n, dtype=torch.float32, device=dev)
        L.check(lib.dgx_centernet_head_outputs_bwd(lv, n, C, L.ptr(g_reg), L.ptr(g_hm), L.ptr(d_scale), L.ptr(ws), L.stream()),
                "dgx_centernet_head_outputs_bwd")
        # the scale gradients go straight into the parameters' arena views (linear_ops.accumulate_grad: None when done in place):
        # no AccumulateGrad node runs for them -- inside a replayed segment those nodes were born on the capture warm-up's side
        # stream, and autograd would run (and warn about) their accumulation there, outside the stream the reducer orders itself behind
        from .linear_ops import accumulate_grad
        return (None,) + tuple(dxs) + tuple(accumulate_grad(s, lambda i=i: d_scale[i:i + 1]) if ctx.needs_input_grad[1 + n + i] else None
                                            for i, s in enumerate(scales))


def centernet_head_outputs(boths, scales):
    """boths: per level the grouped predictor output, logical (B, C, h, w) on channels-last bf16 storage (C >= 8, channel 0 = agn_hm,
    1..4 = bbox_pred); scales: per level the (1,) f32 Scale parameter.  -> (reg_flat (M, 4) f32, hm_flat (M,) f32): what
    centernet_head.py:113-131 + centernet.py:179-235 compute level by level (slices, scale, ReLU, permutes, cats, casts)."""
    xs = []
    for b in boths:
        x = b.permute(0, 2, 3, 1)
        assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[-1] % 8 == 0
        xs.append(x)
    return _HeadOutputs.apply(len(xs), *xs, *scales)
