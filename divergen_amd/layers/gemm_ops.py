"""libdgx's own MFMA GEMM (dgx_gemm_bf16_nt, csrc/gemm_nt.hip) behind every Linear / 1x1 convolution of the path.

    acc[m][n] = sum_k a[m][k] * b[n][k]        a (M, K) bf16, b (N, K) bf16 (an nn.Linear weight as stored)

with the epilogues the Swin block needs fused in: bias; bias + exact GELU (both tensors written); bias + window_reverse /
roll / crop + DropPath + residual add; multiply by GELU'(f1).  The input gradient dx = dy W is the same call on the
transposed weight image (`FlatArena.p16t`, refreshed with the bf16 shadow after every optimizer step).
Reference call sites: swintransformer.py:133,155 (qkv / proj), :40-46 (Mlp), :296 (reduction), fpn.py:126-154, box_head.py:26-98.
"""
import ctypes

import torch

from .. import _lib as L

BF16 = torch.bfloat16
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL, EPI_GELU_GRAD, EPI_RELU_GRAD = 0, 1, 2, 3, 4, 5


def _check2(a, b):
    if a.dtype != BF16 or b.dtype != BF16:
        raise L.DgxError("dgx_gemm_bf16_nt takes bfloat16 operands")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise L.DgxError("gemm_nt: a (M, K) x b (N, K): got %s, %s" % (tuple(a.shape), tuple(b.shape)))
    if a.stride(1) != 1 or b.stride(1) != 1:
        raise L.DgxError("gemm_nt: operands must be K-contiguous")


_WS = {}
WS_BYTES = 64 << 20      # split-K scratch (fp32 slabs of skinny, long-K problems), allocated once per (device, stream)


def _workspace(dev):
    """One scratch buffer per (device, stream): two streams that run split-K GEMMs at the same time (the opt-in mask-branch
    side stream next to the box cascade) must not fold their fp32 slabs in the same buffer."""
    key = (dev, L.stream())
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(WS_BYTES, dtype=torch.uint8, device=dev)
    return ws


def _launch(a, b, ep):
    M, K = a.shape
    N = b.shape[0]
    if M * N * 8 <= WS_BYTES:          # at least two slabs fit: let the library decide whether to split K
        ws = _workspace(a.device)
        ep.workspace, ep.workspace_bytes = ws.data_ptr(), WS_BYTES
    L.check(L.lib().dgx_gemm_bf16_nt(a.data_ptr(), b.data_ptr(), M, N, K, a.stride(0), b.stride(0), ctypes.byref(ep), L.stream()),
            "dgx_gemm_bf16_nt")


def _dev_ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.DgxError("libdgx ops need GPU (ROCm) tensors; got a %s tensor -- no CPU fallback exists" % t.device)
    return t.data_ptr()


def gemm_nt(a, b, bias=None, out=None):
    """bf16 (M, N) = a b^T (+ bias)."""
    _check2(a, b)
    _dev_ptr(a), _dev_ptr(b)
    M, N = a.shape[0], b.shape[0]
    c = out if out is not None else torch.empty(M, N, dtype=BF16, device=a.device)
    ep = L.GemmEpilogue()
    ep.mode = EPI_BIAS if bias is not None else EPI_NONE
    ep.c, ep.ldc, ep.bias = c.data_ptr(), c.stride(0), _dev_ptr(bias)
    _launch(a, b, ep)
    return c


def gemm_nt_act(a, b, bias=None, relu=False):
    """bf16 (M, N) = relu?(a b^T + bias)."""
    _check2(a, b)
    M, N = a.shape[0], b.shape[0]
    c = torch.empty(M, N, dtype=BF16, device=a.device)
    ep = L.GemmEpilogue()
    ep.mode = EPI_BIAS if bias is not None else EPI_NONE
    ep.c, ep.ldc, ep.bias, ep.relu = c.data_ptr(), N, _dev_ptr(bias), int(bool(relu))
    _launch(a, b, ep)
    return c


def gemm_bias_gelu(a, b, bias):
    """(f1, act) = (a b^T + bias, GELU_erf(f1)), both bf16 (M, N): Mlp.fc1 + act (swintransformer.py:41-42)."""
    _check2(a, b)
    M, N = a.shape[0], b.shape[0]
    f1 = torch.empty(M, N, dtype=BF16, device=a.device)
    act = torch.empty(M, N, dtype=BF16, device=a.device)
    ep = L.GemmEpilogue()
    ep.mode = EPI_BIAS_GELU
    ep.c, ep.ldc, ep.bias, ep.c2 = f1.data_ptr(), N, _dev_ptr(bias), act.data_ptr()
    _launch(a, b, ep)
    return f1, act


def gemm_gelu_grad(a, b, f1):
    """bf16 (M, N) = (a b^T) * GELU'(f1): the input gradient of fc2 carried through the activation."""
    _check2(a, b)
    M, N = a.shape[0], b.shape[0]
    assert f1.shape == (M, N) and f1.dtype == BF16 and f1.is_contiguous()
    c = torch.empty(M, N, dtype=BF16, device=a.device)
    ep = L.GemmEpilogue()
    ep.mode = EPI_GELU_GRAD
    ep.c, ep.ldc, ep.aux, ep.ldaux = c.data_ptr(), N, f1.data_ptr(), N
    _launch(a, b, ep)
    return c


def gemm_relu_grad(a, b, act):
    """bf16 (M, N) = (a b^T) where act > 0 else 0: the input gradient of the layer that follows a ReLU, carried through it."""
    _check2(a, b)
    M, N = a.shape[0], b.shape[0]
    assert act.shape == (M, N) and act.dtype == BF16 and act.is_contiguous()
    c = torch.empty(M, N, dtype=BF16, device=a.device)
    ep = L.GemmEpilogue()
    ep.mode = EPI_RELU_GRAD
    ep.c, ep.ldc, ep.aux, ep.ldaux = c.data_ptr(), N, act.data_ptr(), N
    _launch(a, b, ep)
    return c


def gemm_bias_residual(a, b, bias, res, scale, B, H, W, ws, shift):
    """out (B, H*W, N) = res + scale[b] * bf16(a b^T + bias) with the rows of `a` in window order (ws > 0: window_reverse +
    roll + crop folded into the store) or token order (ws == 0); res fp32 | bf16; scale f32 (B) or None."""
    _check2(a, b)
    N = b.shape[0]
    res = res.contiguous()
    assert res.shape[-1] == N and res.numel() == B * H * W * N
    out = torch.empty_like(res)
    ep = L.GemmEpilogue()
    ep.mode = EPI_BIAS_RESIDUAL
    ep.bias, ep.residual, ep.out, ep.scale = _dev_ptr(bias), res.data_ptr(), out.data_ptr(), _dev_ptr(scale)
    ep.residual_dtype = L.dtype_code(res)
    ep.B, ep.H, ep.W, ep.ws, ep.shift = B, H, W, ws, shift
    _launch(a, b, ep)
    return out


def dev_time_us(a, b, bias=None, iters=30, mode=None):
    """Development: average KERNEL time (us) of one GEMM over back-to-back launches (HIP events inside libdgx).
    mode None = bias / plain; EPI_BIAS_GELU and EPI_GELU_GRAD time those epilogues on scratch tensors."""
    _check2(a, b)
    M, N = a.shape[0], b.shape[0]
    c = torch.empty(M, N, dtype=BF16, device=a.device)
    ep = L.GemmEpilogue()
    ep.mode = (EPI_BIAS if bias is not None else EPI_NONE) if mode is None else mode
    ep.c, ep.ldc, ep.bias = c.data_ptr(), N, _dev_ptr(bias)
    if mode in (EPI_BIAS_GELU, EPI_GELU_GRAD):
        extra = torch.randn(M, N, device=a.device).to(BF16)
        ep.c2, ep.aux, ep.ldaux = extra.data_ptr(), extra.data_ptr(), N
    fn = L.lib().dgx_dev_gemm_time_us
    fn.restype = ctypes.c_float
    fn.argtypes = [L.c_p, L.c_p, L.c_i, L.c_i, L.c_i, L.c_i64, L.c_i64, ctypes.POINTER(L.GemmEpilogue), L.c_i, L.c_p]
    return float(fn(a.data_ptr(), b.data_ptr(), M, N, a.shape[1], a.stride(0), b.stride(0), ctypes.byref(ep), iters, L.stream()))
