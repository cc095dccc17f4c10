"""One Swin block as ONE autograd node with a hand-written backward.

Reference: SwinTransformerBlock.forward (DG/divergen/modeling/backbone/swintransformer.py:201-257) =
LN -> pad/roll/partition -> WindowAttention (:126-157) -> reverse/roll/crop -> DropPath + residual ->
LN -> Mlp (:40-46) -> DropPath + residual.

The eager composition of the same kernels costs ~25 autograd nodes per block (x24 blocks) and the step is
bound by the host issuing them, not by the GPU.  Here forward and backward are straight-line sequences of
libdgx kernels and library GEMMs: no autograd bookkeeping between them, weight gradients accumulate in
fp32 directly in the gradient arena (and signal the data-parallel reducer per parameter), bias gradients
come from the column-sum kernel, the residual-branch gradient is folded into the LayerNorm backward, and
the bias-table gradient is scattered by the attention kernel straight into the parameter's gradient.
Used when every parameter of the block lives in a FlatArena (training); anything else takes the
composed path in swintransformer.py, which runs the same kernels in the same order.
"""
import torch

from .. import _lib as L
from . import gemm_ops as G
from .linear_ops import BF16, notify_ready, shadow, shadow_t, wgrad_group_form, wgrad_grouped


def arena_resident(params):
    return all(p.is_leaf and getattr(p, "_dgx16", None) is not None and p.grad is not None
               and p.grad.dtype == torch.float32 for p in params)


def _ready(*params):
    for p in params:
        notify_ready(p)


def colsum_into(gb, dy2, beta=1.0):
    """gb fp32 (N,) = beta*gb + dy2.sum(0); dy2 bf16 (M, N)."""
    M, N = dy2.shape
    lib = L.lib()
    ws = torch.empty(max(int(lib.dgx_colsum_workspace_bytes(M, N)), 4), dtype=torch.uint8, device=dy2.device)
    L.check(lib.dgx_colsum_bf16(L.ptr(dy2), L.ptr(gb), M, N, float(beta), L.ptr(ws), L.stream()), "dgx_colsum_bf16")


def colsum_grouped(problems, beta=1.0):
    """[(gb fp32 (N,), dy2 bf16 (M, N)), ...] (<= 8): gb = beta*gb + dy2.sum(0) for all of them in two launches."""
    n = len(problems)
    arr = (L.ColsumProblem * n)()
    for i, (gb, dy2) in enumerate(problems):
        arr[i].dy, arr[i].out, arr[i].M, arr[i].N = dy2.data_ptr(), gb.data_ptr(), dy2.shape[0], dy2.shape[1]
    lib = L.lib()
    ws = torch.empty(max(int(lib.dgx_colsum_grouped_workspace_bytes(arr, n)), 4), dtype=torch.uint8, device=problems[0][1].device)
    L.check(lib.dgx_colsum_grouped(arr, n, float(beta), L.ptr(ws), L.stream()), "dgx_colsum_grouped")


import os
# Weight / bias gradients of consecutive blocks share grouped launches: a Swin-L stage-2 block alone has 108 output tiles of
# 256x256, so its grouped weight-gradient GEMM would be split in two M-slabs (216 workgroups) whose fp32 partial tiles go through
# a workspace and a reduce kernel; two blocks together are 216 tiles = one round of the chip with NO split: every workgroup runs
# the whole contraction and folds straight into the gradient arena (round 2).  Round 3 packs by PROBLEM instead of by block: a
# launch takes queued problems in order while they fit 256 tiles, i.e. two blocks and the fc1 (or qkv + proj) of a third = 252
# tiles instead of 216 on the 256 CUs, and 18 stage-2 blocks need 8 launches instead of 9 (a launch lasts one tile's contraction
# whatever its tile count).  Operands stay alive until their problem has been launched; whatever is still pending when the backward
# pass ends is flushed by an autograd end-of-backward callback.  (_PAIR = 1 / 2: one launch per block / per block pair, the rounds-2 forms.)
# Round 4: blocks with many output tiles and a long contraction (Swin-L stage 2: 144 tiles of 256x192 per block, M = 8 192 / 10 368) are
# queued until ~4 rounds of the chip are together (7 blocks = 1 008 tiles, 28 problems) and run as ONE launch of the persistent
# loader-wave kernel (csrc/wgrad_lw.hip: every tile contracts its whole M -- no split, no workspace, no reduce launch); the library says
# whether a group qualifies (dgx_wgrad_grouped_form), otherwise the group goes the round-3 way.
_PAIR = 3
_PENDING = []          # [problem (g, dy, x, bias, weight), tiles] in arrival order
_PENDING_LW = []       # problems queued for a loader-wave launch
_CB_QUEUED = [False]
_ROUND, _MAXP = 256, 12
_LW_MIN_M = 4096      # shortest contraction queued for the loader-wave form (stage 3, M = 2 048 / 2 592, is faster split: profiles/r04_wgrad_lw_probe.txt)
_LW_ITEMS, _LW_MAXP = 1000, 32


def _lw_items(g):
    return (-(-g.shape[0] // 256)) * (-(-g.shape[1] // 192))


def _tiles(g):
    return (-(-g.shape[0] // 256)) * (-(-g.shape[1] // 256))


def _launch(items):
    # weight AND bias gradients in one grouped launch: the bias gradient is dY^T 1 on the fragments the weight-gradient kernel
    # holds anyway (round 3; the separate column-sum kernels cost 95 launches / 1.1 ms per step)
    # every problem of the launch the FIRST writer of its gradient segments in this pass (solver.FlatArena.claim_first_write): the
    # launch overwrites (beta = 0) -- nothing read, nothing that had to be zeroed
    members = [q for (g, d, x_, b, w), _ in items for q in ((w,) if b is None else (w, b))]
    slot = getattr(members[0], "_dgx_arena_slot", None)
    first = slot is not None and all(getattr(q, "_dgx_arena_slot", (None,))[0] is slot[0] for q in members) and slot[0].claim_first_write(members)
    wgrad_grouped([(g, d, x_, b.grad if b is not None else None) for (g, d, x_, b, w), _ in items], beta=0.0 if first else 1.0)
    for (g, d, x_, b, w), _ in items:
        _ready(w) if b is None else _ready(w, b)


def flush_wgrads(final=True):
    """Launch pending weight / bias gradients (<= 12 problems and, when several fit, <= 256 tiles per grouped launch) and signal
    their parameters.  final=False (called when a block's problems arrive) keeps the tail that does not yet fill a round."""
    if final:
        _CB_QUEUED[0] = False
        _flush_ln()
        _flush_lw()
    while _PENDING:
        if not final and sum(t for _, t in _PENDING) < _ROUND:      # not enough queued to choose a full round from
            break
        pick, rest, s = [], [], 0
        for it in _PENDING:                          # first fit in arrival order (the order of the launches is free)
            if len(pick) < _MAXP and (s + it[1] <= _ROUND or not pick):
                pick.append(it)
                s += it[1]
            else:
                rest.append(it)
        _PENDING[:] = rest
        _launch(pick)


def reset_pending():
    """Drop problems whose backward pass never completed (an exception unwound it): called when the gradients are cleared for a
    new step, so that a stale problem can never be launched into the gradients of the next step."""
    if _PENDING or _PENDING_LW or _PENDING_LN:
        import warnings
        warnings.warn("divergen_amd: %d pending weight gradients dropped (LayerNorm parameter folds included; an earlier backward pass did not finish)"
                      % (len(_PENDING) + len(_PENDING_LW) + len(_PENDING_LN)))
        del _PENDING[:]
        del _PENDING_LW[:]
        del _PENDING_LN[:]
    _CB_QUEUED[0] = False


_PENDING_LN = []       # [(partial rows, weight, bias)] of LayerNorm backward calls whose second stage has not run; _PENDING_LN_KEY = their (T, C)
_PENDING_LN_KEY = [None]


def _flush_ln():
    """Second stage of the queued LayerNorm parameter gradients (<= 16 norms, one launch: dgx_layernorm_param_reduce_n), then their
    gradient-ready signals."""
    if not _PENDING_LN:
        return
    import ctypes
    n = len(_PENDING_LN)
    T, C = _PENDING_LN_KEY[0]
    arr = lambda vals: (ctypes.c_void_p * n)(*vals)
    L.check(L.lib().dgx_layernorm_param_reduce_n(arr([p.data_ptr() for p, _, _ in _PENDING_LN]), arr([w.grad.data_ptr() for _, w, _ in _PENDING_LN]),
                                                 arr([b.grad.data_ptr() for _, _, b in _PENDING_LN]), n, T, C, L.stream()),
            "dgx_layernorm_param_reduce_n")
    for _, w, b in _PENDING_LN:
        _ready(w, b)
    del _PENDING_LN[:]


def _defer_ln_reduce(part, norms, T, C):
    n2w, n2b, n1w, n1b = norms
    if _PENDING_LN and _PENDING_LN_KEY[0] != (T, C):
        _flush_ln()
    if any(w is n2w or w is n1w for _, w, _ in _PENDING_LN):       # a norm queued twice (shared / re-entered block): two slices of one
        _flush_ln()                                                 # launch would race on its dgamma / dbeta -- serialise as before
    _PENDING_LN_KEY[0] = (T, C)
    _PENDING_LN.append((part[0], n2w, n2b))
    _PENDING_LN.append((part[1], n1w, n1b))
    if len(_PENDING_LN) >= 16:
        _flush_ln()


def _flush_lw():
    """Launch the queued loader-wave group -- in one launch if the library takes it in that form, else the round-3 way."""
    if not _PENDING_LW:
        return
    items = [[p, _tiles(p[0])] for p in _PENDING_LW]
    del _PENDING_LW[:]
    if wgrad_group_form([(g, d, x_, b.grad if b is not None else None) for (g, d, x_, b, w), _ in items]):
        _launch(items)
    else:
        _flush_all()
        _PENDING.extend(items)
        _flush_all()


def _queue_callback():
    if (_PENDING or _PENDING_LW or _PENDING_LN) and not _CB_QUEUED[0]:
        _CB_QUEUED[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(flush_wgrads)


def _defer_wgrads(wgrads, params):
    blk = sum(_tiles(p[0]) for p in wgrads)
    M = wgrads[0][1].shape[0]
    if _PENDING_LW and _PENDING_LW[0][1].shape[0] != M:   # a new stage: the queued group goes out
        _flush_lw()
    if _LW_MIN_M and min(p[1].shape[0] for p in wgrads) >= _LW_MIN_M and sum(_lw_items(p[0]) for p in wgrads) >= 128 and wgrads[0][1].is_cuda:
        if len(_PENDING_LW) + len(wgrads) > _LW_MAXP:
            _flush_lw()
        _PENDING_LW.extend(wgrads)
        if sum(_lw_items(p[0]) for p in _PENDING_LW) >= _LW_ITEMS or len(_PENDING_LW) + len(wgrads) > _LW_MAXP:
            _flush_lw()
        _queue_callback()
        return
    if _PENDING and _PENDING[0][0][1].shape[0] != M:      # a new stage (other token count): its problems do not share launches
        _flush_all()
    if _PAIR >= 3 and 2 * blk <= _ROUND:                  # at least two blocks fit a round: pack by problem
        _PENDING.extend([p, _tiles(p[0])] for p in wgrads)
        flush_wgrads(final=False)
    else:                                                 # one launch per block / per pair of blocks
        _PENDING.extend([p, 0] for p in wgrads)
        if len(_PENDING) // 4 >= min(_PAIR, 2):
            _flush_all()
    _queue_callback()


def _flush_all():
    while _PENDING:
        items = _PENDING[:_MAXP]
        del _PENDING[:_MAXP]
        _launch(items)


def _linear_bwd(dy2, x2, weight, bias, wgrads, gelu_of=None):
    """Weight / bias gradient of y = x W^T + b queued for the block's grouped weight-gradient / bias-gradient launches;
    returns dx = dy W (bf16) from the MFMA GEMM on the transposed weight image -- times GELU'(gelu_of) when given (the
    input gradient of fc2 carried through the activation in the GEMM's epilogue)."""
    wgrads.append((weight.grad.view(weight.shape[0], -1), dy2, x2, bias, weight))
    wt = shadow_t(weight)
    return G.gemm_gelu_grad(dy2, wt, gelu_of) if gelu_of is not None else G.gemm_nt(dy2, wt)


# Hand-over between consecutive blocks of a stage in the backward pass: block i+1's LayerNorm-1 backward writes block i's incoming
# gradient g AND its DropPath-scaled bf16 copy (the operand of block i's fc2 gradients) in the same pass; block i picks the copy up
# here instead of running dgx_residual_bwd over g.  The slot holds g itself, so its storage cannot be recycled while the entry is
# live; a gradient that autograd accumulated from several consumers is another tensor and simply does not match.
_HANDOVER = [None]        # (g tensor, scale tensor or None, scaled bf16 copy)


def _take_handover(g, s2):
    h, _HANDOVER[0] = _HANDOVER[0], None
    if h is None or h[0].data_ptr() != g.data_ptr() or h[0].shape != g.shape or h[0].dtype != g.dtype:
        return None
    same_scale = (h[1] is None and s2 is None) or (h[1] is not None and s2 is not None and h[1].data_ptr() == s2.data_ptr())
    return h[2] if same_scale else None


# Rows between LayerNorm-1 and the proj GEMM in COMPACT window order (csrc/winmap.h): the reference pads the token grid to multiples of
# the window AFTER norm1 (swintransformer.py:216-221) and crops after proj (:248-251), so the padding tokens' qkv rows are the qkv bias
# and their proj rows are thrown away -- 26.6 % of the rows of Swin stages 2 / 3 at 1024^2 (72^2 for 64^2), 65 % of stage 3 at 896^2.
# With COMPACT the qkv / proj forward GEMMs, their input-gradient GEMMs and the proj weight gradient run over the B*H*W real rows only;
# the attention kernels take the bias for a padding token's q / k / v and write its (0, dk, dv) behind the real rows of dqkv, which the
# qkv weight / bias gradient still sums over.  Exact: every real token's value and every parameter gradient is what the padded form
# computes (tests/test_gpu_parity_modules.py::test_swin_block_compact_equals_padded).  False = the rounds 1-5 form.
COMPACT = True


def compact_ok(H, W, ws, shift):
    nWh, nWw = -(-H // ws), -(-W // ws)
    return COMPACT and (nWh * ws != H or nWw * ws != W) and H >= shift and W >= shift


class _SwinBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, region, s1, s2, cfg, prev_scale, n1w, n1b, qw, qb, table, pw, pb, n2w, n2b, w1, b1, w2, b2):
        ctx.prev_scale = prev_scale                # (scale,) when x came straight out of another block of this kind (swin_block below)
        B, H, W, ws, shift, nH, scale, eps1, eps2 = cfg
        lib, st, dev = L.lib(), L.stream(), x.device
        x = x.contiguous()
        C = x.shape[-1]
        T = B * H * W
        code = L.dtype_code(x)
        nW = (-(-H // ws)) * (-(-W // ws))
        B_, N = B * nW, ws * ws
        Tw = B_ * N
        f32 = torch.float32
        compact = compact_ok(H, W, ws, shift)
        wsm = -ws if compact else ws               # the map argument of the kernels: negative = compact window order
        Tr = T if compact else Tw                  # rows the GEMMs around the attention run over
        # LN1 + bf16 + pad + roll + partition (compact: the real tokens first, the padding tokens' zero rows behind them)
        xw = torch.empty(Tw, C, dtype=BF16, device=dev)
        mean1 = torch.empty(T, dtype=f32, device=dev)
        rstd1 = torch.empty(T, dtype=f32, device=dev)
        L.check(lib.dgx_layernorm_fwd(x.data_ptr(), n1w.data_ptr(), n1b.data_ptr(), xw.data_ptr(), mean1.data_ptr(),
                                      rstd1.data_ptr(), T, C, eps1, B, H, W, wsm, shift, code, st), "dgx_layernorm_fwd")
        # attention
        qkv = G.gemm_nt(xw[:Tr], shadow(qw), shadow(qb))
        tbl = table.detach()                       # ((2ws-1)^2, nH) as stored: the kernels take its strides
        o = torch.empty(Tr, C, dtype=BF16, device=dev)
        lse = torch.empty(B_, nH, N, dtype=f32, device=dev)
        if compact:
            L.check(lib.dgx_window_attention_fwd_compact(qkv.data_ptr(), shadow(qb).data_ptr(), tbl.data_ptr(), tbl.stride(1), tbl.stride(0),
                                                         L.ptr(region), o.data_ptr(), lse.data_ptr(), B, H, W, nH, ws, shift, scale, st),
                    "dgx_window_attention_fwd_compact")
        else:
            L.check(lib.dgx_window_attention_fwd(qkv.data_ptr(), tbl.data_ptr(), tbl.stride(1), tbl.stride(0), L.ptr(region),
                                                 o.data_ptr(), lse.data_ptr(), B_, nW, nH, ws, scale, st), "dgx_window_attention_fwd")
        # proj + (reverse + roll + crop + DropPath + residual) in the GEMM's epilogue
        x1 = G.gemm_bias_residual(o, shadow(pw), shadow(pb), x.view(B, H * W, C), s1, B, H, W, wsm, shift).view(x.shape)
        # LN2 + MLP + residual
        h2 = torch.empty(T, C, dtype=BF16, device=dev)
        mean2 = torch.empty(T, dtype=f32, device=dev)
        rstd2 = torch.empty(T, dtype=f32, device=dev)
        L.check(lib.dgx_layernorm_fwd(x1.data_ptr(), n2w.data_ptr(), n2b.data_ptr(), h2.data_ptr(), mean2.data_ptr(),
                                      rstd2.data_ptr(), T, C, eps2, 0, 0, 0, 0, 0, code, st), "dgx_layernorm_fwd")
        f1, a = G.gemm_bias_gelu(h2, shadow(w1), shadow(b1))        # fc1 + exact GELU: both tensors from one epilogue
        out = G.gemm_bias_residual(a, shadow(w2), shadow(b2), x1.view(B, H * W, C), s2, B, H, W, 0, 0).view(x.shape)
        ctx.save_for_backward(x, mean1, rstd1, xw, qkv, region, o, lse, x1, mean2, rstd2, h2, f1, a, s1, s2)
        ctx.params = (n1w, n1b, qw, qb, table, pw, pb, n2w, n2b, w1, b1, w2, b2)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, g):
        x, mean1, rstd1, xw, qkv, region, o, lse, x1, mean2, rstd2, h2, f1, a, s1, s2 = ctx.saved_tensors
        n1w, n1b, qw, qb, table, pw, pb, n2w, n2b, w1, b1, w2, b2 = ctx.params
        B, H, W, ws, shift, nH, scale, eps1, eps2 = ctx.cfg
        lib, st, dev = L.lib(), L.stream(), g.device
        g = g.contiguous()
        C = x.shape[-1]
        T = B * H * W
        code = L.dtype_code(x)
        nW = (-(-H // ws)) * (-(-W // ws))
        B_, N = B * nW, ws * ws
        Tw = B_ * N
        # MLP branch: df2 = s2 * g as bf16 -- handed over by the next block's LayerNorm-1 backward, g itself when there is nothing
        # to scale or convert, a pass over g otherwise
        df2 = _take_handover(g, s2)
        if df2 is None:
            if s2 is None and g.dtype == BF16:
                df2 = g.view(T, C)
            else:
                df2 = torch.empty(T, C, dtype=BF16, device=dev)
                L.check(lib.dgx_residual_bwd(g.data_ptr(), L.ptr(s2), df2.data_ptr(), B, H, W, C, 0, 0, code, st), "dgx_residual_bwd")
        wgrads = []
        df1 = _linear_bwd(df2, a, w2, b2, wgrads, gelu_of=f1)       # (df2 W2) * GELU'(f1)
        dh2 = _linear_bwd(df1, h2, w1, b1, wgrads)
        # LN2 backward + the residual-branch gradient g -> dx1
        nblk = lib.dgx_layernorm_bwd_blocks(T)
        part = torch.empty(2, nblk * 2 * C, dtype=torch.float32, device=dev)     # partial (dgamma | dbeta) rows of norm2, norm1
        # ... and, from the same pass, the attention branch's operand dpr = s1 * dx1 in window order (zero rows for the padding;
        # compact: the real rows only)
        compact = compact_ok(H, W, ws, shift)
        wsm = -ws if compact else ws
        Tr = T if compact else Tw
        dx1 = torch.empty_like(x)
        dpr = torch.empty(Tr, C, dtype=BF16, device=dev)
        L.check(lib.dgx_layernorm_bwd_emit(dh2.data_ptr(), x1.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), n2w.data_ptr(),
                                           g.data_ptr(), dx1.data_ptr(), None, None, part[0].data_ptr(),
                                           T, C, 0, 0, 0, 0, 0, code, dpr.data_ptr(), L.ptr(s1), B, H, W, wsm, shift, st),
                "dgx_layernorm_bwd_emit")
        do = _linear_bwd(dpr, o, pw, pb, wgrads)
        dqkv = torch.empty(Tw, qkv.shape[1], dtype=BF16, device=dev)
        assert table.grad.stride() == table.stride()          # one pair of strides serves the table and its gradient
        if compact:
            L.check(lib.dgx_window_attention_bwd_compact(qkv.data_ptr(), shadow(qb).data_ptr(), table.data_ptr(), L.ptr(region), o.data_ptr(),
                                                         lse.data_ptr(), do.data_ptr(), dqkv.data_ptr(), table.grad.data_ptr(), table.stride(1),
                                                         table.stride(0), B, H, W, nH, ws, shift, scale, st), "dgx_window_attention_bwd_compact")
            _ready(table)
            # qkv weight / bias gradient over ALL rows of dqkv (the padding tokens' dk / dv reach the bias; their xw rows are zero), the
            # input gradient over the real rows only
            wgrads.append((qw.grad.view(qw.shape[0], -1), dqkv, xw, qb, qw))
            dxw = G.gemm_nt(dqkv[:T], shadow_t(qw))
        else:
            L.check(lib.dgx_window_attention_bwd(qkv.data_ptr(), table.data_ptr(), L.ptr(region), o.data_ptr(), lse.data_ptr(),
                                                 do.data_ptr(), dqkv.data_ptr(), table.grad.data_ptr(), table.stride(1), table.stride(0),
                                                 B_, nW, nH, ws, scale, st), "dgx_window_attention_bwd")
            _ready(table)
            dxw = _linear_bwd(dqkv, xw, qw, qb, wgrads)
        # LN1 backward through the window map, accumulated onto dx1 in place
        prev = ctx.prev_scale                      # (scale,) when x came straight out of another block of this kind
        if prev and (prev[0] is not None or dx1.dtype != BF16):
            scaled = torch.empty(T, C, dtype=BF16, device=dev)      # the previous block's fc2-gradient operand, from this pass
            L.check(lib.dgx_layernorm_bwd_emit(dxw.data_ptr(), x.data_ptr(), mean1.data_ptr(), rstd1.data_ptr(), n1w.data_ptr(),
                                               dx1.data_ptr(), dx1.data_ptr(), None, None, part[1].data_ptr(),
                                               T, C, B, H, W, wsm, shift, code, scaled.data_ptr(), L.ptr(prev[0]), B, H, W, 0, 0, st),
                    "dgx_layernorm_bwd_emit")
            _HANDOVER[0] = (dx1, prev[0], scaled)
        else:
            L.check(lib.dgx_layernorm_bwd(dxw.data_ptr(), x.data_ptr(), mean1.data_ptr(), rstd1.data_ptr(), n1w.data_ptr(),
                                          dx1.data_ptr(), dx1.data_ptr(), None, None, part[1].data_ptr(),
                                          T, C, B, H, W, wsm, shift, code, st), "dgx_layernorm_bwd")
        # the second stage of both norms' parameter gradients: queued, the norms of up to eight blocks of a stage share one launch
        _defer_ln_reduce(part, (n2w, n2b, n1w, n1b), T, C)
        # the four weight gradients of the block: one grouped launch (256x256 tiles, small M-split)
        # the four weight gradients of the block (256x256 tiles): launched together with the next block's (flush_wgrads)
        _defer_wgrads(wgrads, (w2, w1, pw, qw, b2, b1, pb, qb))
        return (dx1,) + (None,) * 18


def swin_block(x, region, s1, s2, cfg, params):
    """x (B, H*W, C) fp32|bf16 -> same.  cfg = (B, H, W, ws, shift, nH, scale, eps1, eps2); params in the
    order norm1.{w,b}, qkv.{w,b}, bias table, proj.{w,b}, norm2.{w,b}, fc1.{w,b}, fc2.{w,b}."""
    prev = getattr(x, "_dgx_next_scale", None)     # left on x by the block that produced it (below)
    with torch.autocast("cuda", enabled=False):
        out = _SwinBlockFn.apply(x, region, s1, s2, cfg, prev, *params)
    out._dgx_next_scale = (s2,)                    # a tuple: None is a valid scale
    return out
