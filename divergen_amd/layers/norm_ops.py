"""Fused LayerNorm (+bf16 cast, + window gather) over libdgx.  Reference: nn.LayerNorm call sites
swintransformer.py:213 (norm1, followed by pad/roll/partition :216-233) and :255 (norm2)."""
import torch

from .. import _lib as L
from .linear_ops import accumulate_grad, notify_ready


class _LayerNormBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, B, H, W, ws, shift):
        assert x.dtype in (torch.float32, torch.bfloat16)
        x = x.contiguous()
        C = x.shape[-1]
        T = x.numel() // C
        if ws > 0:
            assert T == B * H * W
            nW = (-(-H // ws)) * (-(-W // ws))
            y = torch.empty(B * nW, ws * ws, C, dtype=torch.bfloat16, device=x.device)
        else:
            y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        mean = torch.empty(T, dtype=torch.float32, device=x.device)
        rstd = torch.empty(T, dtype=torch.float32, device=x.device)
        L.check(L.lib().dgx_layernorm_fwd(L.ptr(x), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(mean), L.ptr(rstd), T, C,
                                          float(eps), B, H, W, ws, shift, L.dtype_code(x), L.stream()), "dgx_layernorm_fwd")
        ctx.save_for_backward(x, mean, rstd)
        ctx.weight, ctx.bias, ctx.cfg = weight, bias, (T, C, B, H, W, ws, shift)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        T, C, B, H, W, ws, shift = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        nblk = L.lib().dgx_layernorm_bwd_blocks(T)
        part = torch.empty(nblk * 2 * C, dtype=torch.float32, device=x.device)
        in_arena = (weight.is_leaf and bias.is_leaf and weight.grad is not None and bias.grad is not None
                    and getattr(weight, "_dgx16", None) is not None and getattr(bias, "_dgx16", None) is not None)
        dg = weight.grad if in_arena else torch.zeros(C, dtype=torch.float32, device=x.device)
        db = bias.grad if in_arena else torch.zeros(C, dtype=torch.float32, device=x.device)
        L.check(L.lib().dgx_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(weight), None, L.ptr(dx), L.ptr(dg),
                                          L.ptr(db), L.ptr(part), T, C, B, H, W, ws, shift, L.dtype_code(x), L.stream()), "dgx_layernorm_bwd")
        if in_arena:
            for p in (weight, bias):
                notify_ready(p)
            return dx, None, None, None, None, None, None, None, None
        return dx, dg.to(weight.dtype), db.to(bias.dtype), None, None, None, None, None, None


class _LayerNormF32Out(torch.autograd.Function):
    """LayerNorm with an fp32 result (PatchEmbed.norm under autocast): x f32|bf16 -> y f32."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        C = x.shape[-1]
        T = x.numel() // C
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        mean = torch.empty(T, dtype=torch.float32, device=x.device)
        rstd = torch.empty(T, dtype=torch.float32, device=x.device)
        L.check(L.lib().dgx_layernorm_f32out_fwd(L.ptr(x), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(mean), L.ptr(rstd), T, C,
                                                 float(eps), L.dtype_code(x), L.stream()), "dgx_layernorm_f32out_fwd")
        ctx.save_for_backward(x, mean, rstd)
        ctx.weight, ctx.bias, ctx.cfg = weight, bias, (T, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        T, C = ctx.cfg
        dy = dy.float().contiguous()
        dx = torch.empty_like(x)
        part = torch.empty(L.lib().dgx_layernorm_bwd_blocks(T) * 2 * C, dtype=torch.float32, device=x.device)
        in_arena = (weight.is_leaf and bias.is_leaf and weight.grad is not None and bias.grad is not None
                    and getattr(weight, "_dgx16", None) is not None and getattr(bias, "_dgx16", None) is not None)
        dg = weight.grad if in_arena else torch.zeros(C, dtype=torch.float32, device=x.device)
        db = bias.grad if in_arena else torch.zeros(C, dtype=torch.float32, device=x.device)
        L.check(L.lib().dgx_layernorm_f32out_bwd(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(weight), L.ptr(dx), L.ptr(dg),
                                                 L.ptr(db), L.ptr(part), T, C, L.dtype_code(x), L.stream()), "dgx_layernorm_f32out_bwd")
        if in_arena:
            for p in (weight, bias):
                notify_ready(p)
            return dx, None, None, None
        return dx, dg.to(weight.dtype), db.to(bias.dtype), None


def layernorm_f32out(x, weight, bias, eps=1e-5):
    """f32|bf16 (..., C) -> LayerNorm -> f32 (..., C)."""
    return _LayerNormF32Out.apply(x, weight, bias, eps)


class _PatchMergeLN(torch.autograd.Function):
    """PatchMerging's pad + 2x2 gather + LayerNorm(4C) (swintransformer.py:272-298) as one kernel each way."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, B, H, W):
        assert x.dtype in (torch.float32, torch.bfloat16)
        x = x.contiguous()
        C0 = x.shape[-1]
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        T2 = B * H2 * W2
        y = torch.empty(B, H2 * W2, 4 * C0, dtype=torch.bfloat16, device=x.device)
        mean = torch.empty(T2, dtype=torch.float32, device=x.device)
        rstd = torch.empty(T2, dtype=torch.float32, device=x.device)
        L.check(L.lib().dgx_patch_merge_ln_fwd(L.ptr(x), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(mean), L.ptr(rstd), B, H, W, C0,
                                               float(eps), L.dtype_code(x), L.stream()), "dgx_patch_merge_ln_fwd")
        ctx.save_for_backward(x, mean, rstd)
        ctx.weight, ctx.bias, ctx.cfg = weight, bias, (B, H, W, C0, T2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        B, H, W, C0, T2 = ctx.cfg
        C = 4 * C0
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        nblk = L.lib().dgx_layernorm_bwd_blocks(T2)
        part = torch.empty(nblk * 2 * C, dtype=torch.float32, device=x.device)
        in_arena = (weight.is_leaf and bias.is_leaf and weight.grad is not None and bias.grad is not None
                    and getattr(weight, "_dgx16", None) is not None and getattr(bias, "_dgx16", None) is not None)
        dg = weight.grad if in_arena else torch.zeros(C, dtype=torch.float32, device=x.device)
        db = bias.grad if in_arena else torch.zeros(C, dtype=torch.float32, device=x.device)
        L.check(L.lib().dgx_patch_merge_ln_bwd(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(weight), L.ptr(dx), L.ptr(dg),
                                               L.ptr(db), L.ptr(part), B, H, W, C0, L.dtype_code(x), L.stream()),
                "dgx_patch_merge_ln_bwd")
        if in_arena:
            for p in (weight, bias):
                notify_ready(p)
            return dx, None, None, None, None, None, None
        return dx, dg.to(weight.dtype), db.to(bias.dtype), None, None, None, None


def patch_merge_layernorm(x, weight, bias, eps, B, H, W):
    """x (B, H*W, C0) f32|bf16 -> bf16 (B, ceil(H/2)*ceil(W/2), 4*C0): pad, 2x2 gather, LayerNorm."""
    return _PatchMergeLN.apply(x, weight, bias, eps, B, H, W)


def layernorm_bf16(x, weight, bias, eps=1e-5):
    """fp32 (..., C) -> LayerNorm -> bf16 (..., C)."""
    return _LayerNormBF16.apply(x, weight, bias, eps, 0, 0, 0, 0, 0)


def layernorm_window_gather(x, weight, bias, eps, B, H, W, ws, shift):
    """fp32 (B, H*W, C) -> LayerNorm -> zero-pad, roll(-shift), partition -> bf16 (B*nW, ws*ws, C)."""
    return _LayerNormBF16.apply(x, weight, bias, eps, B, H, W, ws, shift)


class _ResidualAdd(torch.autograd.Function):
    """out = x + scale[b] * y;  y bf16 in token order (ws == 0) or window order (ws > 0)."""

    @staticmethod
    def forward(ctx, x, y, scale, B, H, W, ws, shift):
        x = x.contiguous()
        y = y.contiguous()
        C = x.shape[-1]
        out = torch.empty_like(x)
        L.check(L.lib().dgx_residual_fwd(L.ptr(x), L.ptr(y), L.ptr(scale), L.ptr(out), B, H, W, C, ws, shift,
                                         L.dtype_code(x), L.stream()), "dgx_residual_fwd")
        ctx.scale, ctx.cfg, ctx.yshape = scale, (B, H, W, C, ws, shift), y.shape
        return out

    @staticmethod
    def backward(ctx, g):
        B, H, W, C, ws, shift = ctx.cfg
        g = g.contiguous()
        dy = torch.empty(ctx.yshape, dtype=torch.bfloat16, device=g.device)
        L.check(L.lib().dgx_residual_bwd(L.ptr(g), L.ptr(ctx.scale), L.ptr(dy), B, H, W, C, ws, shift, L.dtype_code(g),
                                         L.stream()), "dgx_residual_bwd")
        return g, dy, None, None, None, None, None, None


class _Upsample2xAdd(torch.autograd.Function):
    """lat + nearest-upsampled top on channels-last maps (FPN top-down step): one launch each way."""

    @staticmethod
    def forward(ctx, lat, top):
        N, C, H, W = lat.shape
        latc = lat.contiguous(memory_format=torch.channels_last)
        topc = top.to(lat.dtype).contiguous(memory_format=torch.channels_last)
        out = torch.empty_like(latc)
        nhwc = lambda t: t.permute(0, 2, 3, 1)          # the channels-last storage as a contiguous (N, H, W, C) view
        L.check(L.lib().dgx_upsample2x_add_fwd(L.ptr(nhwc(latc)), L.ptr(nhwc(topc)), L.ptr(nhwc(out)), N, H, W, C, L.dtype_code(latc),
                                               L.stream()), "dgx_upsample2x_add_fwd")
        ctx.cfg, ctx.top_dtype = (N, C, H, W), top.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.cfg
        g = g.contiguous(memory_format=torch.channels_last)
        gtop = torch.empty((N, C, H // 2, W // 2), dtype=g.dtype, device=g.device).contiguous(memory_format=torch.channels_last)
        L.check(L.lib().dgx_upsample2x_add_bwd(L.ptr(g.permute(0, 2, 3, 1)), L.ptr(gtop.permute(0, 2, 3, 1)), N, H, W, C, L.dtype_code(g),
                                               L.stream()), "dgx_upsample2x_add_bwd")
        return g, gtop.to(ctx.top_dtype)


def upsample2x_add(lat, top):
    """FPN top-down step, fpn.py:139-145: lat (N,C,H,W) + F.interpolate(top (N,C,H/2,W/2), scale_factor=2, mode='nearest').  GPU maps with
    C % 8 == 0 and even H, W run dgx_upsample2x_add_fwd/bwd (anything else on a GPU raises); host tensors take torch's two ops (host
    logic tests)."""
    if not lat.is_cuda:
        return lat + torch.nn.functional.interpolate(top, scale_factor=2.0, mode="nearest")
    if tuple(top.shape) != (lat.shape[0], lat.shape[1], lat.shape[2] // 2, lat.shape[3] // 2) or lat.shape[2] % 2 or lat.shape[3] % 2:
        raise L.DgxError("upsample2x_add: lateral %s against top-down %s -- the kernel takes exact 2x pairs" % (tuple(lat.shape), tuple(top.shape)))
    return _Upsample2xAdd.apply(lat, top)


def residual_add(x, y, scale, B, H, W, ws=0, shift=0):
    """x (B, H*W, C) fp32|bf16, y bf16 ((B, H*W, C) or windows (B*nW, ws*ws, C)), scale (B,) f32 or None."""
    return _ResidualAdd.apply(x, y, scale, B, H, W, ws, shift)


class _GroupNormReLU(torch.autograd.Function):
    """GroupNorm (8 channels per group) + optional ReLU on channels-last bf16 activations (libdgx dgx_groupnorm_*).
    x: physical (N, H, W, C) bf16 contiguous."""

    @staticmethod
    def forward(ctx, x, weight, bias, G, eps, relu):
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(N * G, dtype=torch.float32, device=x.device)
        rstd = torch.empty(N * G, dtype=torch.float32, device=x.device)
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        scratch = torch.empty(int(L.lib().dgx_groupnorm_scratch_floats(N, H * W, G)), dtype=torch.float32, device=x.device)
        L.check(L.lib().dgx_groupnorm_fwd(L.ptr(x), L.ptr(w32), L.ptr(b32), L.ptr(y), L.ptr(mean), L.ptr(rstd), L.ptr(scratch), N, H * W, C, G,
                                          float(eps), int(relu), L.stream()), "dgx_groupnorm_fwd")
        ctx.save_for_backward(x, mean, rstd, w32, b32)
        ctx.weight, ctx.bias, ctx.cfg = weight, bias, (N, H, W, C, G, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, w32, b32 = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        N, H, W, C, G, relu = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = torch.empty_like(x)
        part = torch.empty(int(L.lib().dgx_groupnorm_scratch_floats(N, H * W, G)), dtype=torch.float32, device=x.device)
        in_arena = (weight.is_leaf and bias.is_leaf and weight.grad is not None and bias.grad is not None
                    and weight.grad.dtype == torch.float32 and getattr(weight, "_dgx16", None) is not None
                    and getattr(bias, "_dgx16", None) is not None)
        dg = weight.grad if in_arena else torch.zeros(C, dtype=torch.float32, device=x.device)
        db = bias.grad if in_arena else torch.zeros(C, dtype=torch.float32, device=x.device)
        L.check(L.lib().dgx_groupnorm_bwd(L.ptr(x), L.ptr(dy), L.ptr(mean), L.ptr(rstd), L.ptr(w32), L.ptr(b32), L.ptr(dx), L.ptr(dg),
                                          L.ptr(db), L.ptr(part), N, H * W, C, G, int(relu), L.stream()), "dgx_groupnorm_bwd")
        if in_arena:
            notify_ready(weight)
            notify_ready(bias)
            return dx, None, None, None, None, None
        return dx, dg.to(weight.dtype), db.to(bias.dtype), None, None, None


class _GroupNormReLUMulti(torch.autograd.Function):
    """The same GroupNorm (+ ReLU) over several channels-last bf16 tensors -- the FPN levels of one CenterNet tower layer, which
    share its weights -- in ONE launch per pass (dgx_groupnorm_fwd_multi / _bwd_multi; <= 8 tensors).  xs: (N, H, W, C) bf16."""

    @staticmethod
    def forward(ctx, weight, bias, G, eps, relu, *xs):
        n = len(xs)
        C = xs[0].shape[-1]
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        items = (L.GnItem * n)()
        ys, saved, keep = [], [], []
        lib = L.lib()
        for i, x in enumerate(xs):
            N, H, W, _ = x.shape
            y = torch.empty_like(x)
            mean = torch.empty(N * G, dtype=torch.float32, device=x.device)
            rstd = torch.empty(N * G, dtype=torch.float32, device=x.device)
            scratch = torch.empty(int(lib.dgx_groupnorm_scratch_floats(N, H * W, G)), dtype=torch.float32, device=x.device)
            it = items[i]
            it.x, it.dy, it.out, it.mean, it.rstd, it.scratch = L.ptr(x), None, L.ptr(y), L.ptr(mean), L.ptr(rstd), L.ptr(scratch)
            it.N, it.HW = N, H * W
            ys.append(y)
            saved += [x, mean, rstd]
            keep.append(scratch)                       # alive until the launches below are enqueued
        L.check(lib.dgx_groupnorm_fwd_multi(items, n, L.ptr(w32), L.ptr(b32), C, G, float(eps), int(relu), L.stream()), "dgx_groupnorm_fwd_multi")
        ctx.save_for_backward(w32, b32, *saved)
        ctx.weight, ctx.bias, ctx.cfg = weight, bias, (G, relu, n, C)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        w32, b32 = ctx.saved_tensors[:2]
        saved = ctx.saved_tensors[2:]
        weight, bias = ctx.weight, ctx.bias
        G, relu, n, C = ctx.cfg
        dev = w32.device
        in_arena = (weight.is_leaf and bias.is_leaf and weight.grad is not None and bias.grad is not None
                    and weight.grad.dtype == torch.float32 and getattr(weight, "_dgx16", None) is not None
                    and getattr(bias, "_dgx16", None) is not None)
        dg = weight.grad if in_arena else torch.zeros(C, dtype=torch.float32, device=dev)
        db = bias.grad if in_arena else torch.zeros(C, dtype=torch.float32, device=dev)
        items = (L.GnItem * n)()
        dxs, keep = [], []
        lib = L.lib()
        for i in range(n):
            x, mean, rstd = saved[3 * i:3 * i + 3]
            N, H, W, _ = x.shape
            dy = dys[i]
            if dy is None:
                dy = torch.zeros_like(x)
            dy = dy.contiguous()
            if dy.dtype != torch.bfloat16:
                dy = dy.to(torch.bfloat16)
            dx = torch.empty_like(x)
            part = torch.empty(int(lib.dgx_groupnorm_scratch_floats(N, H * W, G)), dtype=torch.float32, device=dev)
            it = items[i]
            it.x, it.dy, it.out, it.mean, it.rstd, it.scratch = L.ptr(x), L.ptr(dy), L.ptr(dx), L.ptr(mean), L.ptr(rstd), L.ptr(part)
            it.N, it.HW = N, H * W
            dxs.append(dx)
            keep += [dy, part]
        L.check(lib.dgx_groupnorm_bwd_multi(items, n, L.ptr(w32), L.ptr(b32), L.ptr(dg), L.ptr(db), C, G, int(relu), L.stream()),
                "dgx_groupnorm_bwd_multi")
        if in_arena:
            notify_ready(weight)
            notify_ready(bias)
            return (None, None, None, None, None) + tuple(dxs)
        return (dg.to(weight.dtype), db.to(bias.dtype), None, None, None) + tuple(dxs)


def groupnorm_relu_multi(xs, weight, bias, num_groups, eps=1e-5, relu=True):
    """list of logical (N, C, H, W) tensors with channels-last storage -> list of the same, bf16: one GroupNorm (+ ReLU) over all
    of them in one launch per pass (the levels of one tower layer)."""
    C = xs[0].shape[1]
    if not (all(x.is_cuda and x.shape[1] == C for x in xs) and C == 8 * num_groups and len(xs) <= 8):
        raise L.DgxError("groupnorm_relu_multi: <= 8 GPU inputs with the same C = 8 * groups required")
    xps = [x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous() for x in xs]
    with torch.autocast("cuda", enabled=False):
        ys = _GroupNormReLUMulti.apply(weight, bias, num_groups, eps, relu, *xps)
    return [y.permute(0, 3, 1, 2) for y in ys]


def groupnorm_relu(x, weight, bias, num_groups, eps=1e-5, relu=True):
    """x logical (N, C, H, W) with channels-last storage -> same, bf16 (the only form built: 8 channels per group)."""
    N, C, H, W = x.shape
    if not (x.is_cuda and C == 8 * num_groups):
        raise L.DgxError("groupnorm_relu: GPU input with 8 channels per group required (C %d, groups %d, %s)" % (C, num_groups, x.device))
    xp = x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()
    with torch.autocast("cuda", enabled=False):
        y = _GroupNormReLU.apply(xp, weight, bias, num_groups, eps, relu)
    return y.permute(0, 3, 1, 2)
