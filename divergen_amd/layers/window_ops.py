"""Swin window ops: attention core (fwd/bwd), window gather/scatter, shift-region ids.
Reference: DG/divergen/modeling/backbone/swintransformer.py:126-157, :216-251, :368-387."""
import math

import torch

from .. import _lib as L


def shift_regions(H, W, ws):
    """(nW, ws*ws) int8 region id of every token of every window position for SW-MSA; the
    reference's additive mask (swintransformer.py:368-387) is (region_i != region_j) ? -100 : 0.
    Host-side integer arithmetic, cached by the caller per (H, W, ws)."""
    shift = ws // 2
    Hp = int(math.ceil(H / ws)) * ws
    Wp = int(math.ceil(W / ws)) * ws
    hr = torch.zeros(Hp, dtype=torch.int8)
    hr[Hp - ws:Hp - shift] = 1
    hr[Hp - shift:] = 2
    wr = torch.zeros(Wp, dtype=torch.int8)
    wr[Wp - ws:Wp - shift] = 1
    wr[Wp - shift:] = 2
    reg = hr[:, None] * 3 + wr[None, :]
    reg = reg.reshape(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    return reg.contiguous()


class _WindowAttentionCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, table, region, nW, nH, ws, scale):
        if qkv.dtype != torch.bfloat16:
            raise L.DgxError("window_attention_core: qkv must be bfloat16")
        B_, N, C3 = qkv.shape
        assert N == ws * ws and C3 == 3 * nH * 32, (qkv.shape, ws, nH)
        qkv = qkv.contiguous()
        table = table.float().t().contiguous()          # (nH, T): one contiguous row per head
        out = torch.empty(B_, N, nH * 32, dtype=torch.bfloat16, device=qkv.device)
        lse = torch.empty(B_, nH, N, dtype=torch.float32, device=qkv.device)
        L.check(L.lib().dgx_window_attention_fwd(L.ptr(qkv), L.ptr(table), table.shape[1], 1, L.ptr(region), L.ptr(out), L.ptr(lse),
                                                 B_, nW, nH, ws, scale, L.stream()), "dgx_window_attention_fwd")
        ctx.save_for_backward(qkv, table, region, out, lse)
        ctx.cfg = (nW, nH, ws, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, table, region, out, lse = ctx.saved_tensors
        nW, nH, ws, scale = ctx.cfg
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dtable = torch.zeros_like(table)
        L.check(L.lib().dgx_window_attention_bwd(L.ptr(qkv), L.ptr(table), L.ptr(region), L.ptr(out), L.ptr(lse),
                                                 L.ptr(dout), L.ptr(dqkv), L.ptr(dtable), dtable.shape[1], 1, qkv.shape[0], nW, nH, ws,
                                                 scale, L.stream()), "dgx_window_attention_bwd")
        return dqkv, dtable.t(), None, None, None, None, None


def window_attention_core(qkv, table, region, nW, nH, ws, scale):
    """qkv bf16 (B_, N, 3*nH*32) -> bf16 (B_, N, nH*32).  region: int8 (nW, N) or None."""
    return _WindowAttentionCore.apply(qkv, table, region, nW, nH, ws, scale)


def _shuffle(gather, src, B, H, W, C, ws, shift):
    nWh, nWw = -(-H // ws), -(-W // ws)
    src = src.contiguous()
    if gather:
        dst = torch.empty(B * nWh * nWw, ws * ws, C, dtype=src.dtype, device=src.device)
        fn = L.lib().dgx_window_gather
    else:
        dst = torch.empty(B, H * W, C, dtype=src.dtype, device=src.device)
        fn = L.lib().dgx_window_scatter
    L.check(fn(L.ptr(src), L.ptr(dst), B, H, W, C, ws, shift, L.dtype_code(src), L.stream()), "dgx_window_shuffle")
    return dst


class _WindowGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W, ws, shift):
        B, Ltok, C = x.shape
        assert Ltok == H * W
        ctx.cfg = (B, H, W, C, ws, shift)
        return _shuffle(True, x, B, H, W, C, ws, shift)

    @staticmethod
    def backward(ctx, g):
        B, H, W, C, ws, shift = ctx.cfg
        return _shuffle(False, g, B, H, W, C, ws, shift), None, None, None, None


class _WindowScatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xw, B, H, W, ws, shift):
        C = xw.shape[-1]
        ctx.cfg = (B, H, W, C, ws, shift)
        return _shuffle(False, xw, B, H, W, C, ws, shift)

    @staticmethod
    def backward(ctx, g):
        B, H, W, C, ws, shift = ctx.cfg
        return _shuffle(True, g, B, H, W, C, ws, shift), None, None, None, None, None


def window_gather(x, H, W, ws, shift):
    """x (B, H*W, C) -> (B*nW, ws*ws, C): zero-pad, roll(-shift), partition."""
    return _WindowGather.apply(x, H, W, ws, shift)


def window_scatter(xw, B, H, W, ws, shift):
    """(B*nW, ws*ws, C) -> (B, H*W, C): reverse, roll(+shift), crop."""
    return _WindowScatter.apply(xw, B, H, W, ws, shift)
