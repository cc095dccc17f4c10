"""Linear layers over the flat parameter arenas.

A parameter that lives in a FlatArena carries `_dgx16` (its bf16 shadow, refreshed by the optimizer
kernel) and a persistent fp32 `.grad` view.  The GEMMs then run bf16 x bf16 -> bf16 with no per-step
weight casts, and the weight gradient is ONE library GEMM that accumulates in fp32 straight into the
gradient arena (beta = 1) -- no bf16 gradient tensor, no cast, no AccumulateGrad add.  That removes
~6 tiny launches per Linear per step (~1000 launches for Swin-L CenterNet2).
Reference: nn.Linear call sites in swintransformer.py (qkv/proj/fc1/fc2/reduction), box_head.py, fast_rcnn.py."""
import torch
import torch.nn as nn

from .. import _lib as L

BF16 = torch.bfloat16
import os
WGRAD_MIN_M = int(os.environ.get("DGX_WGRAD_MIN_M", 4096))   # measured crossover vs the library GEMM (tools/wgrad_probe2.py): long-M shapes only


def wgrad_into(g2, dy2, x2, beta=1.0):
    """g2 fp32 (Nn,Kk) = beta*g2 + dy2^T x2: the 256x256 split-M MFMA kernel (dgx_linear_wgrad_grouped, here a group
    of one) for the long contractions of this model, the library GEMM for short ones."""
    M, Nn = dy2.shape
    Kk = x2.shape[1]
    if (dy2.is_cuda and M >= WGRAD_MIN_M and Nn % 8 == 0 and Kk % 8 == 0 and g2.is_contiguous() and dy2.is_contiguous()
            and x2.is_contiguous() and dy2.dtype == BF16 and x2.dtype == BF16 and M * max(Nn, Kk) * 2 < (1 << 31)):
        wgrad_grouped([(g2, dy2, x2)], beta)
    elif beta == 0.0:
        torch.mm(dy2.t(), x2, out_dtype=torch.float32, out=g2)
    else:
        torch.addmm(g2, dy2.t(), x2, out_dtype=torch.float32, out=g2)


def wgrad_grouped(problems, beta=1.0):
    """[(g2 fp32 (Nn,Kk), dy2 bf16 (M,Nn), x2 bf16 (M,Kk)), ...] (<= 8): g2 = beta*g2 + dy2^T x2, ONE launch
    (dgx_linear_wgrad_grouped: 256x256 MFMA tiles, M-split sized for the whole group)."""
    n = len(problems)
    arr = (L.WgradProblem * n)()
    for i, (g2, dy2, x2) in enumerate(problems):
        assert g2.is_contiguous() and dy2.is_contiguous() and x2.is_contiguous()
        assert dy2.dtype == BF16 and x2.dtype == BF16 and g2.dtype == torch.float32
        arr[i].dy, arr[i].x, arr[i].gw = dy2.data_ptr(), x2.data_ptr(), g2.data_ptr()
        arr[i].M, arr[i].Nn, arr[i].Kk = dy2.shape[0], dy2.shape[1], x2.shape[1]
    lib = L.lib()
    nbytes = int(lib.dgx_wgrad_grouped_workspace_bytes(arr, n))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=problems[0][1].device)
    L.check(lib.dgx_linear_wgrad_grouped(arr, n, float(beta), L.ptr(ws), L.stream()), "dgx_linear_wgrad_grouped")


_READY_SUSPENDED = [False]


class suspend_ready:
    """Context: gradient-ready callbacks (data-parallel reducer) are not fired -- used while a segment is being
    captured into a hipGraph, whose warm-up backward passes are not part of any training step."""

    def __enter__(self):
        self.prev = _READY_SUSPENDED[0]
        _READY_SUSPENDED[0] = True

    def __exit__(self, *a):
        _READY_SUSPENDED[0] = self.prev


def notify_ready(p):
    r = getattr(p, "_dgx_ready", None)
    if r is not None and not _READY_SUSPENDED[0] and not (p.is_cuda and torch.cuda.is_current_stream_capturing()):
        r()


def shadow(p):
    """bf16 view of a parameter: the arena shadow if present, else a cast (CPU tests, pre-arena)."""
    s = getattr(p, "_dgx16", None)
    return s if s is not None else p.detach().to(BF16)


def shadow_t(p):
    """bf16 TRANSPOSE of a matrix parameter, (numel / shape[0], shape[0]): the arena's transposed twin if present (refreshed
    after every optimizer step, solver.FlatArena.refresh_transposes), else a transposed copy (pre-arena use)."""
    s = getattr(p, "_dgx16t", None)
    if s is not None:
        return s
    w = shadow(p)
    return w.reshape(w.shape[0], -1).t().contiguous()


def _own_gemm_ok(x2, n, k):
    return x2.is_cuda and n % 8 == 0 and k % 8 == 0


def accumulate_grad(p, make_grad_fp32, gemm_into=None):
    """Write a gradient into p's arena view.  Returns None if done in place (and signals the
    data-parallel reducer), else the gradient tensor for autograd to accumulate."""
    g = p.grad if p.is_leaf else None
    if g is not None and g.dtype == torch.float32 and getattr(p, "_dgx16", None) is not None:
        if gemm_into is not None:
            gemm_into(g)
        else:
            g.add_(make_grad_fp32())
        notify_ready(p)
        return None
    return make_grad_fp32().to(p.dtype)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        w16 = shadow(weight)
        w16 = w16.reshape(w16.shape[0], -1)          # conv 1x1 weights (Cout,Cin,1,1) are Linear weights
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != BF16:
            x2 = x2.to(BF16)
        x2 = x2.contiguous()
        if _own_gemm_ok(x2, w16.shape[0], w16.shape[1]):
            from .gemm_ops import gemm_nt
            y = gemm_nt(x2, w16, shadow(bias) if bias is not None else None)
        elif bias is not None:                       # widths that are not multiples of 8 (cls_score 1454, bbox_pred 4): library
            y = torch.addmm(shadow(bias), x2, w16.t())
        else:
            y = torch.mm(x2, w16.t())
        ctx.save_for_backward(x2, w16)
        ctx.weight, ctx.bias, ctx.xshape, ctx.xdtype = weight, bias, x.shape, x.dtype
        return y.view(*x.shape[:-1], w16.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w16 = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        dy2 = dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if _own_gemm_ok(dy2, w16.shape[1], w16.shape[0]):
                from .gemm_ops import gemm_nt
                dx = gemm_nt(dy2, shadow_t(weight)).view(ctx.xshape)
            else:
                dx = torch.mm(dy2, w16).view(ctx.xshape)
            if ctx.xdtype != BF16:
                dx = dx.to(ctx.xdtype)
        gw = gb = None
        if ctx.needs_input_grad[1]:
            def into(g):
                wgrad_into(g.view(g.shape[0], -1), dy2, x2)
            gw = accumulate_grad(weight, lambda: torch.mm(dy2.t(), x2, out_dtype=torch.float32).view(weight.shape),
                                 gemm_into=into)
        if bias is not None and ctx.needs_input_grad[2]:
            gb = accumulate_grad(bias, lambda: torch.sum(dy2, 0, dtype=torch.float32))
        return dx, gw, gb


def linear(x, weight, bias=None):
    with torch.autocast("cuda", enabled=False):
        return _LinearFn.apply(x, weight, bias)


class Linear(nn.Linear):
    """nn.Linear parameters (checkpoint-compatible); bf16 GEMMs over the arena shadow."""

    def forward(self, x):
        if x.dtype == torch.float32 and not torch.is_autocast_enabled():
            return torch.nn.functional.linear(x, self.weight, self.bias)   # fp32 parity mode (cfg.FP16 off)
        return linear(x, self.weight, self.bias)
