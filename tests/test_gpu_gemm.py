"""GPU parity of libdgx's MFMA GEMM (dgx_gemm_bf16_nt) and its fused epilogues through the C ABI, against fp32 math on the
same bf16-rounded inputs (the contract of the Linear sites it replaces: swintransformer.py:40-46,133,155,296).
Tolerance: the result is rounded to bf16 once (2^-8 relative) after fp32 accumulation; products of bf16 values are exact in
fp32, so only the summation order differs from the reference (<= 1e-5 of the row scale)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

from divergen_amd.layers import gemm_ops as G  # noqa: E402
from oracle import swin as OSW  # noqa: E402

TILES = ["256x192", "192x192", "128x192", "128x256", "256x128", "128x128", None]


def bf(x):
    return x.to(torch.bfloat16)


def close_bf16(got, ref, extra=0.0):
    got, ref = got.float().cpu(), ref.float().cpu()
    tol = (2.0 ** -8 + extra) * ref.abs() + 2e-5 * ref.abs().max() + 1e-6
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), (int(bad.sum()), float((got - ref).abs().max()), float(ref.abs().max()))


@pytest.fixture
def tile(request, dgx_dev):
    t = request.param
    if t is not None:
        bm, bn = t.split("x")
        dgx_dev("gemm_tile", int(bm) * 1000 + int(bn))
    return t


@pytest.mark.parametrize("tile", TILES, indirect=True)
@pytest.mark.parametrize("M,N,K", [(512, 384, 256), (1000, 200, 136), (77, 1160, 72), (2592, 1536, 1536), (8, 8, 8),
                                   (300, 192, 64), (10368, 768, 768)])
def test_gemm_nt_bias(tile, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a, b = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.3)
    bias = bf(torch.randn(N, generator=g))
    ref = a.float() @ b.float().t()
    close_bf16(G.gemm_nt(a.to(DEV), b.to(DEV)), ref)
    # bias is added in fp32 before the single rounding
    close_bf16(G.gemm_nt(a.to(DEV), b.to(DEV), bias.to(DEV)), ref + bias.float())


@pytest.mark.parametrize("M,N,K,S", [(1024, 1024, 12544, 0), (128, 256, 2304, 0), (2048, 1536, 6144, 4), (300, 200, 1096, 3), (512, 384, 640, 2)])
def test_gemm_split_k(dgx_dev, M, N, K, S):
    """Few output tiles and a long contraction: K is cut into slabs (fp32 partial sums in the workspace) and folded by a
    second launch with the same epilogue; S = 0 leaves the split count to the library's plan."""
    if S:
        dgx_dev("gemm_splitk", S)
    g = torch.Generator().manual_seed(K)
    a, b = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.1)
    bias = bf(torch.randn(N, generator=g))
    ref = a.float() @ b.float().t() + bias.float()
    close_bf16(G.gemm_nt(a.to(DEV), b.to(DEV), bias.to(DEV)), ref)
    f1, act = G.gemm_bias_gelu(a.to(DEV), b.to(DEV), bias.to(DEV))
    close_bf16(f1, ref)
    close_bf16(act, torch.nn.functional.gelu(f1.float().cpu()), extra=1e-3)
    if M % 4 == 0:
        B, H, W = 1, M // 4, 4
        res = torch.randn(B, H * W, N, generator=g)
        scale = torch.tensor([0.5])
        out = G.gemm_bias_residual(a.to(DEV), b.to(DEV), bias.to(DEV), res.to(DEV), scale.to(DEV), B, H, W, 0, 0)
        want = res + 0.5 * bf(ref).float().reshape(B, H * W, N)
        assert float((out.cpu() - want).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())


def test_gelu_epilogue_accuracy_vs_float64():
    """The epilogue's erfc-polynomial GELU against float64 erf on the same bf16 pre-activations: the difference must stay
    inside the single bf16 rounding of the result (plus the stated 1.5e-7 absolute error of the polynomial)."""
    x = torch.linspace(-9.0, 9.0, 4096 * 8).reshape(4096, 8)
    a = bf(torch.cat([x, torch.zeros(4096, 8)], 1))                       # (4096, 16): K = 16, picks x via identity columns
    w = bf(torch.cat([torch.eye(8), torch.zeros(8, 8)], 1))               # (8, 16)
    f1, act = G.gemm_bias_gelu(a.to(DEV), w.to(DEV), bf(torch.zeros(8)).to(DEV))
    assert torch.equal(f1.cpu(), a[:, :8])
    xd = f1.double().cpu()
    ref = 0.5 * xd * (1.0 + torch.erf(xd / 2 ** 0.5))
    err = (act.double().cpu() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 2e-6).all()), float(err.max())
    df = G.gemm_gelu_grad(bf(torch.cat([torch.ones(4096, 1), torch.zeros(4096, 15)], 1)).to(DEV),
                          bf(torch.cat([torch.ones(8, 1), torch.zeros(8, 15)], 1)).to(DEV), f1)
    gref = 0.5 * (1.0 + torch.erf(xd / 2 ** 0.5)) + xd * torch.exp(-0.5 * xd * xd) / (2 * torch.pi) ** 0.5
    gerr = (df.double().cpu() - gref).abs()
    assert bool((gerr <= 2.0 ** -8 * gref.abs() + 2e-6).all()), float(gerr.max())


def test_gemm_nt_identity_detects_transposes():
    """A = I against an asymmetric B: the output must be B^T exactly (no rounding: every output is one bf16 value)."""
    n = 448
    b = bf(torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125.0)
    out = G.gemm_nt(bf(torch.eye(n)).to(DEV), b.to(DEV))
    assert torch.equal(out.cpu(), b.t().contiguous())


def test_gemm_nt_strided_operands_and_output():
    g = torch.Generator().manual_seed(5)
    M, N, K = 700, 384, 320
    abig, bbig = bf(torch.randn(M, K + 64, generator=g)).to(DEV), bf(torch.randn(N, K + 8, generator=g)).to(DEV)
    a, b = abig[:, :K], bbig[:, :K]
    out = torch.zeros(M, N + 16, dtype=torch.bfloat16, device=DEV)
    G.gemm_nt(a, b, out=out[:, :N])
    close_bf16(out[:, :N], a.float().cpu() @ b.float().cpu().t())
    assert float(out[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("tile", ["256x192", "128x192", None], indirect=True)
def test_gemm_bias_gelu_and_grad(tile):
    g = torch.Generator().manual_seed(11)
    M, N, K = 1111, 768, 192
    a, w, bias = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.1), bf(torch.randn(N, generator=g))
    f1, act = G.gemm_bias_gelu(a.to(DEV), w.to(DEV), bias.to(DEV))
    ref_f1 = a.float() @ w.float().t() + bias.float()
    close_bf16(f1, ref_f1)
    # the activation is applied to the bf16-ROUNDED pre-activation (what a separate GELU pass over f1 reads)
    ref_act = torch.nn.functional.gelu(f1.float().cpu())
    close_bf16(act, ref_act, extra=1e-3)
    # backward through the activation fused into the input-gradient GEMM of fc2: d_f1 = (dy W2) * GELU'(f1)
    dy, w2t = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.1)    # (M, C) x W2^T image (4C, C)
    df1 = G.gemm_gelu_grad(dy.to(DEV), w2t.to(DEV), f1)
    da = bf(dy.float() @ w2t.float().t()).float()          # the separate-kernel path rounds dy W2 to bf16 first
    x = f1.float().cpu().requires_grad_(True)
    torch.nn.functional.gelu(x).backward(da)
    got, ref = df1.float().cpu(), x.grad
    assert float((got - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())


@pytest.mark.parametrize("tile", ["256x192", "192x192", None], indirect=True)
@pytest.mark.parametrize("B,H,W,ws,shift,dt", [(2, 30, 26, 12, 6, torch.float32), (1, 24, 24, 12, 0, torch.bfloat16),
                                               (2, 17, 20, 0, 0, torch.float32), (3, 10, 13, 7, 3, torch.bfloat16)])
def test_gemm_bias_residual_window_map(tile, B, H, W, ws, shift, dt):
    """proj / fc2 epilogue: window_reverse + roll + crop + DropPath + residual add folded into the store
    (swintransformer.py:239-255), against the oracle's partition / unpartition index arithmetic."""
    g = torch.Generator().manual_seed(B * 100 + H)
    C = 192
    if ws:
        Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
        M = B * (Hp // ws) * (Wp // ws) * ws * ws
    else:
        M = B * H * W
    a, w, bias = bf(torch.randn(M, C, generator=g)), bf(torch.randn(C, C, generator=g) * 0.1), bf(torch.randn(C, generator=g))
    res = torch.randn(B, H * W, C, generator=g).to(dt)
    scale = torch.tensor([1.0 / 0.7, 0.0, 1.0 / 0.7][:B])
    y = bf(a.float() @ w.float().t() + bias.float()).float()                     # Linear output, rounded once
    if ws:
        yy = OSW.unpartition(y.reshape(-1, ws * ws, C), ws, Hp, Wp)
        if shift:
            yy = torch.roll(yy, (shift, shift), (1, 2))
        yy = yy[:, :H, :W].reshape(B, H * W, C)
    else:
        yy = y.reshape(B, H * W, C)
    ref = (res.float() + scale[:, None, None] * yy).to(dt)
    out = G.gemm_bias_residual(a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV), scale.to(DEV), B, H, W, ws, shift)
    assert out.dtype == dt and out.shape == res.shape
    # y itself may differ from the host's rounding by one bf16 ulp on a few elements (summation order at a rounding
    # boundary): compare with a 2^-8 budget on the added term
    err = (out.float().cpu() - ref.float()).abs()
    tol = 2.0 ** -7 * (scale[:, None, None] * yy).abs() + (2.0 ** -8 * ref.float().abs() if dt == torch.bfloat16 else 0) + 1e-5
    assert not bool((err > tol).any()), float(err.max())


def test_gemm_rejects_bad_arguments():
    a = torch.zeros(16, 12, dtype=torch.bfloat16, device=DEV)     # K % 8 != 0
    b = torch.zeros(8, 12, dtype=torch.bfloat16, device=DEV)
    from divergen_amd._lib import DgxError
    with pytest.raises(DgxError):
        G.gemm_nt(a, b)
    with pytest.raises(DgxError):
        G.gemm_nt(torch.zeros(16, 16), torch.zeros(8, 16))        # not bf16 / not on the GPU


@pytest.mark.parametrize("M,N,K", [(1024, 1024, 1024), (1024, 12544, 1024), (300, 200, 136), (2048, 1536, 6144)])
def test_gemm_relu_grad(M, N, K):
    """DGX_EPI_RELU_GRAD: (a b^T) masked by the saved activation (box-head FCs: box_head.py:26-98) -- exact masking of the
    bf16-rounded product, split-K and plain paths."""
    g = torch.Generator().manual_seed(M + K)
    a, b = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) * 0.1)
    act = bf(torch.relu(torch.randn(M, N, generator=g)))
    ref = (a.float() @ b.float().t()) * (act.float() > 0)
    got = G.gemm_relu_grad(a.to(DEV), b.to(DEV), act.to(DEV))
    close_bf16(got, ref)
    assert bool((got.float().cpu()[act.float() <= 0] == 0).all())
