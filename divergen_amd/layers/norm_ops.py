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


def groupnorm_relu(x, weight, bias, num_groups, eps=1e-5, relu=True):
    """x logical (N, C, H, W) with channels-last storage -> same, bf16 (the only form built: 8 channels per group)."""
    N, C, H, W = x.shape
    if not (x.is_cuda and C == 8 * num_groups):
        raise L.DgxError("groupnorm_relu: GPU input with 8 channels per group required (C %d, groups %d, %s)" % (C, num_groups, x.device))
    xp = x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()
    with torch.autocast("cuda", enabled=False):
        y = _GroupNormReLU.apply(xp, weight, bias, num_groups, eps, relu)
    return y.permute(0, 3, 1, 2)
