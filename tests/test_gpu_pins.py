"""Parity pins at the OPERATING POINT of bench.py (Swin-L, 1024^2, 2 images per GPU): the kernels whose time the headline rests on,
checked in the launch configuration the bench times -- not at the small shapes of the unit tests.

* the grouped weight gradients exactly as layers/swin_block.py queues them in stage 2: 7 blocks = 28 problems per launch of the
  persistent loader-wave kernel (csrc/wgrad_lw.hip), M = 10 368 (qkv / proj) and 8 192 (fc1 / fc2), beta 0 and 1, bias sums,
  against an fp64 contraction; a 32-problem group; a group whose M is not a multiple of the 64-row K-tile
  (reference: the autograd of F.linear behind swintransformer.py:40-46,133,155);
* the top rows of profiles/r04_gemm_insitu.txt (shape x tail mode) through DEFAULT dispatch against fp32 math, with
  dgx_gemm_last_form asserting that the kernel form the bench runs is the one that was checked;
* both again with dgx_set_reserved_cus(16) -- the grid the persistent kernels use beside RCCL's channels at N > 1 (engine/ddp.py).

The references are fp32 / fp64 contractions by torch on the GPU (rocBLAS as the checker, not the product) AND an fp64 CPU contraction of
sampled output rows, so that no result depends on one library.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

from divergen_amd import _lib as L  # noqa: E402
from divergen_amd.layers import gemm_ops as G  # noqa: E402
from divergen_amd.layers.linear_ops import wgrad_group_form, wgrad_grouped  # noqa: E402
from oracle import swin as OSW  # noqa: E402

BF = torch.bfloat16


def _rand(gen, *shape, scale=1.0):
    return (torch.randn(*shape, device=DEV, generator=gen) * scale).to(BF)


@pytest.fixture
def reserved(request):
    n = request.param
    L.lib().dgx_set_reserved_cus(n)
    yield n
    L.lib().dgx_set_reserved_cus(0)


# ------------------------------------------------------------------------------------------------------------------ weight gradients
def _stage2_block(gen, C=768, Mw=10368, Mt=8192):
    """(dy, x) of the four Linears of one Swin-L stage-2 block, in the order swin_block.backward queues them: fc2, fc1, proj, qkv."""
    return [(_rand(gen, Mt, C, scale=0.5), _rand(gen, Mt, 4 * C, scale=0.5)),          # fc2: dY (tokens, C), X (tokens, 4C)
            (_rand(gen, Mt, 4 * C, scale=0.5), _rand(gen, Mt, C, scale=0.5)),          # fc1
            (_rand(gen, Mw, C, scale=0.5), _rand(gen, Mw, C, scale=0.5)),              # proj (window rows)
            (_rand(gen, Mw, 3 * C, scale=0.5), _rand(gen, Mw, C, scale=0.5))]          # qkv


def _check_group(pairs, beta, with_bias, sample_rows=24, seed=0):
    gen = torch.Generator(device=DEV).manual_seed(1000 + seed)
    probs, g0s, b0s = [], [], []
    for k, (dy, x) in enumerate(pairs):
        g0 = torch.randn(dy.shape[1], x.shape[1], device=DEV, generator=gen)
        b0 = torch.randn(dy.shape[1], device=DEV, generator=gen) if with_bias and k % 5 != 4 else None      # some problems without a bias
        g0s.append(g0.clone())
        b0s.append(None if b0 is None else b0.clone())
        probs.append((g0, dy, x, b0))
    assert wgrad_group_form(probs) == 1, "this group must take the persistent loader-wave form (what bench.py times)"
    wgrad_grouped(probs, beta=beta)
    torch.cuda.synchronize()
    cpu_gen = torch.Generator().manual_seed(seed)
    for (gw, dy, x, gb), g0, b0 in zip(probs, g0s, b0s):
        ref = dy.double().t() @ x.double()                       # fp64 contraction (products of bf16 values are exact in it)
        if beta:
            ref += beta * g0.double()
        sc = float(ref.abs().max())
        err = float((gw.double() - ref).abs().max())
        assert err <= 2e-5 * sc + 1e-3, ("weight gradient", tuple(gw.shape), dy.shape[0], err, sc)
        # the same rows contracted on the CPU: independent of the GPU's fp64 GEMM
        rows = torch.randint(0, dy.shape[1], (sample_rows,), generator=cpu_gen)
        ref_c = dy[:, rows.to(DEV)].cpu().double().t() @ x.cpu().double()
        if beta:
            ref_c += beta * g0[rows.to(DEV)].cpu().double()
        assert float((gw[rows.to(DEV)].cpu().double() - ref_c).abs().max()) <= 2e-5 * sc + 1e-3
        if gb is not None:
            bref = dy.double().sum(0) + (beta * b0.double() if beta else 0)
            assert float((gb.double() - bref).abs().max()) <= 2e-5 * float(bref.abs().max()) + 1e-3, ("bias gradient", tuple(gb.shape))


@pytest.mark.parametrize("reserved", [0, 16], indirect=True)
@pytest.mark.parametrize("beta", [0.0, 1.0])
def test_wgrad_lw_at_the_bench_group_7_stage2_blocks(beta, reserved):
    """7 Swin-L stage-2 blocks = 28 problems, 1 008 items of 256 x 192 = 3.94 rounds of the chip, K-tile counts 162 (M = 10 368) and 128
    (M = 8 192): the rings' 12-half unroll wraps 27 / 21.3 times per item -- the configuration `_defer_wgrads` builds in the bench."""
    gen = torch.Generator(device=DEV).manual_seed(5)
    pairs = [p for _ in range(7) for p in _stage2_block(gen)]
    assert len(pairs) == 28
    _check_group(pairs, beta, with_bias=True, seed=int(beta) + reserved)


def test_wgrad_lw_32_problems_and_ragged_contraction():
    """The largest group the ABI takes (32 problems), and one whose contraction is not a multiple of the 64-row K-tile nor of the 32-row
    granule (M = 8 192 + 40, 10 368 - 24), with ragged tiles in both directions."""
    gen = torch.Generator(device=DEV).manual_seed(6)
    pairs = [p for _ in range(8) for p in _stage2_block(gen, C=576, Mw=4608 + 16, Mt=4096)]
    assert len(pairs) == 32
    _check_group(pairs, 1.0, with_bias=True, seed=2)
    gen = torch.Generator(device=DEV).manual_seed(7)
    pairs = [p for _ in range(3) for p in _stage2_block(gen, C=776, Mw=10368 - 24, Mt=8192 + 40)]
    _check_group(pairs, 0.0, with_bias=True, seed=3)


# --------------------------------------------------------------------------------------------------------------- forward / dgrad GEMMs
NT, LW, TWO, K192 = 0, 1, 2, 3      # dgx_gemm_last_form: gemm_nt, gemm_lw (persistent loader-wave), gemm_nt two workgroups per CU, gemm_k192 (resident panel)
# (M, N, K, mode, window map (B, H, W, ws, shift, residual dtype) or None, expected (form, bm, bn) under DEFAULT dispatch).
# Shapes x tail modes: the top rows of profiles/r04_gemm_insitu.txt (>= 0.1 ms/step each; 9.6 of the family's 11.1 ms/step).
# The expected form is this round's dispatch (csrc/gemm_nt.hip::use_lw / use_two_wg): when the dispatch changes, this table changes
# with it -- that is the point of the pin.
BENCH_GEMMS = [
    (8192, 3072, 768, 4, None, None),                                   # fc2 input gradient x GELU'(f1), stage 2
    (8192, 3072, 768, 2, None, None),                                   # fc1 + GELU, stage 2
    (8192, 768, 3072, 3, (2, 64, 64, 0, 0, BF), (LW, 128, 192)),        # fc2 + DropPath + residual
    (10368, 2304, 768, 1, None, (LW, 256, 192)),                        # qkv
    (8192, 768, 3072, 0, None, (LW, 128, 192)),                         # fc1 input gradient
    (10368, 768, 2304, 0, None, (LW, 192, 192)),                        # qkv input gradient
    (10368, 768, 768, 3, (2, 64, 64, 12, 6, BF), None),                 # proj + window reverse + roll + crop + DropPath + residual
    (10368, 768, 768, 0, None, (LW, 192, 192)),                         # proj input gradient
    (131072, 768, 192, 2, None, (K192, 32, 192)),                       # stage 0 fc1 + GELU
    (131072, 768, 192, 4, None, (K192, 32, 192)),                       # stage 0 fc2 input gradient
    (131072, 192, 768, 3, (2, 256, 256, 0, 0, torch.float32), None),    # stage 0 fc2 + residual (fp32 stream)
    (32768, 1536, 384, 2, None, None),                                  # stage 1
    (32768, 1536, 384, 4, None, None),
    (139392, 576, 192, 1, None, (K192, 32, 192)),                       # stage 0 qkv (264-padded grid)
    (1024, 12544, 1024, 0, None, (LW, 128, 256)),                       # box head fc1 input gradient
    (1024, 1024, 12544, 1, None, (LW, 128, 256)),                       # box head fc1 (split-K)
    (32768, 384, 1536, 3, (2, 128, 128, 0, 0, BF), (LW, 256, 192)),     # stage 1 fc2 + residual
    (131072, 192, 768, 0, None, (LW, 256, 192)),                        # stage 0 fc1 input gradient
    (139392, 192, 192, 3, (2, 256, 256, 12, 6, torch.float32), (K192, 32, 192)),   # stage 0 proj + residual
    (139392, 192, 192, 0, None, (K192, 32, 192)),                       # stage 0 proj input gradient
    (2048, 6144, 1536, 4, None, (NT, 256, 192)),                        # stage 3 fc2 input gradient x GELU'(f1)
    (34848, 1152, 384, 1, None, (LW, 192, 192)),                        # stage 1 qkv
    (2592, 4608, 1536, 1, None, (LW, 128, 192)),                        # stage 3 qkv
]


def _close(got, ref, what, extra=0.0):
    got, ref = got.float(), ref.float()
    tol = (2.0 ** -8 + extra) * ref.abs() + 2e-5 * float(ref.abs().max()) + 1e-6
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), (what, int(bad.sum()), float((got - ref).abs().max()), float(ref.abs().max()))


def _gelu_grad64(x):
    x = x.double()
    return 0.5 * (1.0 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5


def _run_bench_gemm(M, N, K, mode, wmap):
    gen = torch.Generator(device=DEV).manual_seed(M + 3 * N + 7 * K + mode)
    a, b = _rand(gen, M, K), _rand(gen, N, K, scale=0.06)
    y = a.float() @ b.float().t()                                   # fp32 accumulate of exact bf16 products
    rows = torch.randint(0, M, (48,), generator=torch.Generator().manual_seed(M + mode)).to(DEV)
    y_cpu = a[rows].cpu().double() @ b.cpu().double().t()            # sampled rows, contracted on the host in fp64
    assert float((y[rows].cpu().double() - y_cpu).abs().max()) <= 1e-4 * float(y_cpu.abs().max()), "the GPU reference itself"
    what = (M, N, K, mode)
    if mode == 0:
        _close(G.gemm_nt(a, b), y, what)
    elif mode == 1:
        bias = _rand(gen, N)
        _close(G.gemm_nt(a, b, bias), y + bias.float(), what)
    elif mode == 2:
        bias = _rand(gen, N)
        f1, act = G.gemm_bias_gelu(a, b, bias)
        _close(f1, y + bias.float(), what)
        _close(act, torch.nn.functional.gelu(f1.float()), what, extra=1e-3)     # GELU of the ROUNDED pre-activation (what a separate pass reads)
    elif mode == 4:
        f1 = _rand(gen, M, N, scale=1.5)
        got = G.gemm_gelu_grad(a, b, f1)
        ref = (y.to(BF).double() * _gelu_grad64(f1)).float()                    # dy W2 rounded to bf16 first, like the separate-kernel path
        assert float((got.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()), what
    elif mode == 3:
        B, H, W, ws, shift, rdt = wmap
        bias = _rand(gen, N)
        res = torch.randn(B, H * W, N, device=DEV, generator=gen).to(rdt)
        scale = torch.tensor([1.0 / 0.7, 0.5][:B], device=DEV)
        out = G.gemm_bias_residual(a, b, bias, res, scale, B, H, W, ws, shift)
        yy = (y + bias.float()).to(BF).float()
        if ws:
            Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
            assert M == B * (Hp // ws) * (Wp // ws) * ws * ws
            yy = OSW.unpartition(yy.reshape(-1, ws * ws, N), ws, Hp, Wp)
            if shift:
                yy = torch.roll(yy, (shift, shift), (1, 2))
            yy = yy[:, :H, :W].reshape(B, H * W, N)
        else:
            yy = yy.reshape(B, H * W, N)
        ref = (res.float() + scale[:, None, None] * yy).to(rdt)
        err = (out.float() - ref.float()).abs()
        # one bf16 ulp of y (summation order at a rounding boundary) and, for a bf16 stream, one ulp of the sum (the shifted addend
        # may flip the rounding of the result: 15-55 of 6.3 M elements do, on gemm_nt and gemm_lw alike)
        tol = 2.0 ** -7 * (scale[:, None, None] * yy).abs() + (2.0 ** -7 * ref.float().abs() if rdt == BF else 0) + 1e-5
        assert out.dtype == rdt and not bool((err > tol).any()), (what, float(err.max()))
    bm, bn, sp = L.c_i(), L.c_i(), L.c_i()
    form = L.lib().dgx_gemm_last_form(bm, bn, sp)
    return form, bm.value, bn.value, sp.value


@pytest.mark.parametrize("reserved", [0, 16], indirect=True)
@pytest.mark.parametrize("M,N,K,mode,wmap,expect", BENCH_GEMMS)
def test_bench_gemm_shapes_through_default_dispatch(M, N, K, mode, wmap, expect, reserved):
    form, bm, bn, splits = _run_bench_gemm(M, N, K, mode, wmap)
    if expect is None:        # the K <= 768 GEMMs with a cold-operand / two-tensor tail: round 5 moves them between forms; either is checked here
        assert form in (LW, TWO), (form, bm, bn)
    else:
        assert (form, bm, bn) == expect, ("default dispatch launched another kernel form than the pin expects", form, bm, bn, expect)
    if (M, N, K) == (1024, 1024, 12544):
        assert splits > 1


@pytest.mark.parametrize("force", [0, 1])
@pytest.mark.parametrize("M,N,K,mode,wmap,expect", [r for r in BENCH_GEMMS if r[3] in (2, 3, 4) and r[2] <= 768])
def test_bench_gemm_fused_tails_on_both_forms(M, N, K, mode, wmap, expect, force, dgx_dev):
    """The K <= 768 fused-tail shapes on gemm_nt AND on gemm_lw: whichever the dispatcher picks in a later build has been checked
    (forcing a form also takes the K = 192 shapes away from the resident-panel kernel, which the default-dispatch test covers)."""
    dgx_dev("gemm_lw", force)
    form, bm, bn, _ = _run_bench_gemm(M, N, K, mode, wmap)
    assert (form == LW) == bool(force)


@pytest.mark.parametrize("reserved", [0, 16], indirect=True)
@pytest.mark.parametrize("M,N,K,mode,wmap", [(131072 + 40, 768, 192, 2, None), (4356 * 32 - 24, 576, 192, 1, None), (65536, 192, 192, 0, None),
                                             (139392, 192, 192, 3, (2, 256, 256, 12, 6, BF)), (40000, 384, 192, 4, None)])
def test_resident_panel_gemm_edges(M, N, K, mode, wmap, reserved):
    """gemm_k192 (one workgroup = one 192-column panel of W resident in LDS, one wave = one 32-row tile at a time) away from the benchmark's
    sizes: row counts that are no multiple of 32 (the last tile of some wave is partial), one / two / three / four column panels (three do
    not divide the workgroups of an XCD), a bf16 residual stream through the window map, and the grid the reducer leaves (`reserved`)."""
    form, bm, bn, _ = _run_bench_gemm(M, N, K, mode, wmap)
    assert (form, bm, bn) == (K192, 32, 192)
