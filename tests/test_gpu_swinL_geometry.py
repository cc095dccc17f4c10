"""GPU parity at the geometry `bench.py` times (BASELINE configs[2]: Swin-L, window 12, 1024 px, 2 images/GPU).

The kernel tests elsewhere use small window batches; at the benchmarked sizes the attention kernels walk CHUNKS of windows
per workgroup (register prefetch of the next window, bias-table gradient accumulated across the chunk), which small batches
never reach.  Here the same C-ABI entry points run on the exact launch shapes of the four Swin-L stages at 1024 px
(968x6, 242x12, 72x24, 18x48 window-heads of 144 tokens) against the fp32 oracle math on the same bf16 inputs, a Swin-L
stage-0 BasicLayer pair (W-MSA + SW-MSA on the 264-padded grid) runs against oracle/swin.py, and the registry-built
Swin-L model runs a whole training forward against the assembled oracle.
Reference: DG/divergen/modeling/backbone/swintransformer.py:126-157 (WindowAttention), :201-257 (block), :361-400
(BasicLayer + shift mask)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from divergen_amd import layers as la  # noqa: E402
from oracle import swin as OSW  # noqa: E402
from tests.test_gpu_kernels import _attn_ref, bf  # noqa: E402


# (B_, nH, grid side H of the stage at 1024 px, masked): 2 images, window 12
@pytest.mark.parametrize("B_,nH,H,masked", [(968, 6, 256, True), (968, 6, 256, False), (242, 12, 128, True),
                                            (72, 24, 64, True), (18, 48, 32, True), (18, 48, 32, False)])
def test_window_attention_at_bench_geometry(B_, nH, H, masked):
    ws, N = 12, 144
    nW = (-(-H // ws)) ** 2
    assert B_ == 2 * nW
    g = torch.Generator().manual_seed(B_ + nH)
    qkv = bf(torch.randn(B_, N, 3 * nH * 32, generator=g) * 1.5)
    table = torch.randn((2 * ws - 1) ** 2, nH, generator=g)
    region = la.shift_regions(H, H, ws) if masked else None           # the ids the model feeds the kernel
    assert region is None or tuple(region.shape) == (nW, N)
    scale = 32 ** -0.5
    go = bf(torch.randn(B_, N, nH * 32, generator=g))
    qr = qkv.clone().float().requires_grad_(True)
    tr = table.clone().requires_grad_(True)
    ref = _attn_ref(qr, tr, region, nH, ws, scale)
    ref.backward(go.float())

    qd = qkv.to(DEV).requires_grad_(True)
    td = table.to(DEV).requires_grad_(True)
    out = la.window_attention_core(qd, td, region.to(DEV) if masked else None, nW if masked else 1, nH, ws, scale)
    out.backward(go.to(DEV))
    torch.cuda.synchronize()

    def close(a, b, frac, l2, what):
        """max error against the tensor's scale (catches a wrong element) AND relative L2 error (catches a small systematic
        error that a max-scaled bound hides under the tensor's largest entries)."""
        err, sc = float((a - b).abs().max()), float(b.abs().max())
        r2 = float((a - b).double().norm() / b.double().norm().clamp(min=1e-30))
        print("%s (B_ %d nH %d masked %s): max err / scale %.2e, relative L2 %.2e" % (what, B_, nH, masked, err / sc, r2))
        assert err <= frac * sc, (what, err, sc)
        assert r2 <= l2, (what, r2)
    # bf16 I/O, bf16 P / dS operands, fp32 accumulation: 1.5 % of the tensor's scale (the bar of the small-batch test); relative
    # L2: one bf16 rounding of the result is 2^-9 / sqrt(3) = 1.1e-3, the bf16 P / dS operands add about as much again
    close(out.float().cpu(), ref.detach(), 0.015, 4e-3, "out")
    close(qd.grad.float().cpu(), qr.grad, 0.015, 6e-3, "dqkv")
    # the bias-table gradient sums B_/nW * N*N terms per entry in fp32 atomics across workgroup chunks: same 1.5 %
    close(td.grad.cpu(), tr.grad, 0.015, 6e-3, "dtable")


@pytest.mark.parametrize("ws,B_,nH,reserved", [(12, 74, 24, 0), (12, 72, 24, 16), (12, 13, 48, 0), (7, 103, 24, 0), (7, 5, 3, 0)])
def test_window_attention_backward_runs_across_heads(ws, B_, nH, reserved):
    """The backward kernel's workgroups walk runs of (window, head) units; the runs over the windows a head has left after its full runs
    cross head boundaries: the bias-gradient row is flushed (pair matrix over the staging images, one lane per table entry, slot in the
    workspace, the last run of a head adds the slots up) and the next head's bias row loaded in the middle of a run.  Shapes whose runs
    do that (74 x 24: 240 runs of 7 + 16 of 6 over 4 left-over windows per head; 13 x 48 and window 7 at 103 x 24 likewise -- window 7
    also re-zeroes the images' padding rows behind the matrix), the stage-2 shape on the grid the data-parallel reducer leaves
    (`dgx_set_reserved_cus(16)`: 216 runs of 8), and a launch with fewer units than CUs: dqkv and the table gradient against the fp32
    oracle math, the table gradient BIT-identical from run to run (it was a sum of float atomics in arrival order before round 5),
    and -- the gradient is ADDED to what the table's gradient holds -- twice the value after a second backward."""
    from divergen_amd import _lib as L
    N = ws * ws
    g = torch.Generator().manual_seed(B_ * 7 + nH)
    qkv = bf(torch.randn(B_, N, 3 * nH * 32, generator=g) * 1.5)
    table = torch.randn((2 * ws - 1) ** 2, nH, generator=g)
    scale = 32 ** -0.5
    go = bf(torch.randn(B_, N, nH * 32, generator=g))
    qr = qkv.clone().float().requires_grad_(True)
    tr = table.clone().requires_grad_(True)
    _attn_ref(qr, tr, None, nH, ws, scale).backward(go.float())

    def run(twice=False):
        qd = qkv.to(DEV).requires_grad_(True)
        td = table.to(DEV).requires_grad_(True)
        out = la.window_attention_core(qd, td, None, 1, nH, ws, scale)
        out.backward(go.to(DEV), retain_graph=twice)
        if twice:
            out.backward(go.to(DEV))
        torch.cuda.synchronize()
        return qd.grad.float().cpu(), td.grad.cpu()
    L.set_reserved_cus(reserved)
    try:
        (dq, dt), (dq2, dt2), (_, dt3) = run(), run(), run(twice=True)
    finally:
        L.set_reserved_cus(0)
    assert torch.equal(dt, dt2) and torch.equal(dq, dq2)
    for a, b, what in ((dq, qr.grad, "dqkv"), (dt, tr.grad, "dtable")):
        err, sc = float((a - b).abs().max()), float(b.abs().max())
        r2 = float((a - b).double().norm() / b.double().norm())
        assert err <= 0.015 * sc and r2 <= 6e-3, (what, err, sc, r2)
    assert float((dt3 - 2 * dt).abs().max()) <= 1e-5 * float(dt.abs().max())


def test_window_attention_backward_back_to_back_launches():
    """The table gradient's workspace protocol (slot stores, per-head counters that the last run of a head resets, one workspace per
    stream) under launches of ten different shapes queued back to back, three times over, with no host synchronisation in between: every
    launch's table gradient against the fp32 evaluation of the same sums and bit-identical in a second pass over the same inputs."""
    torch.manual_seed(0)
    shapes = [(12, 72, 24), (12, 74, 24), (12, 13, 48), (12, 242, 12), (7, 103, 24), (7, 50, 3), (12, 5, 6), (12, 968, 6), (12, 18, 48), (7, 8, 1)]
    scale = 32 ** -0.5
    cases = []
    for _ in range(3):
        for ws, B_, nH in shapes:
            N = ws * ws
            cases.append((ws, nH, (torch.randn(B_, N, 3 * nH * 32, device=DEV) * 1.5).to(torch.bfloat16),
                          torch.randn((2 * ws - 1) ** 2, nH, device=DEV), torch.randn(B_, N, nH * 32, device=DEV).to(torch.bfloat16)))
    passes = []
    for _ in range(2):
        res = []
        for ws, nH, qkv, table, go in cases:
            qd, td = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
            la.window_attention_core(qd, td, None, 1, nH, ws, scale).backward(go)
            res.append(td.grad)
        torch.cuda.synchronize()
        passes.append(res)
    for (ws, nH, qkv, table, go), a, b in zip(cases, passes[0], passes[1]):
        B_, N = qkv.shape[0], ws * ws
        q, k, v = qkv.float().view(B_, N, 3, nH, 32).permute(2, 0, 3, 1, 4)
        idx = OSW.relative_position_index(ws).to(DEV).reshape(-1)
        p = torch.softmax((q * scale) @ k.transpose(-2, -1) + table[idx].reshape(N, N, nH).permute(2, 0, 1)[None], -1)
        dp = go.float().view(B_, N, nH, 32).permute(0, 2, 1, 3) @ v.transpose(-2, -1)
        ds = p * (dp - (dp * p).sum(-1, keepdim=True))
        ref = torch.zeros_like(table).index_add_(0, idx, ds.sum(0).permute(1, 2, 0).reshape(N * N, nH))
        assert torch.equal(a, b), (ws, B_, nH)
        assert float((a - ref).abs().max()) <= 0.015 * float(ref.abs().max()), (ws, B_, nH)


def test_swinL_stage0_layer_pair_vs_oracle():
    """BasicLayer of Swin-L stage 0 at 1024 px: 256x256 tokens (padded to 264 for window 12), C = 192, 6 heads, one W-MSA
    and one SW-MSA block + PatchMerging, one image; product under bf16 autocast (fused block path through the arenas) vs
    oracle/swin.py fp32 with the same weights: forward, input gradient and two parameter gradients."""
    from divergen_amd.modeling.backbone.swintransformer import BasicLayer, PatchMerging
    from divergen_amd.solver import FlatArena
    torch.manual_seed(3)
    C, nH, ws, H = 192, 6, 12, 256
    layer = BasicLayer(dim=C, depth=2, num_heads=nH, window_size=ws, drop_path=0.0, downsample=PatchMerging)
    for n_, p_ in layer.named_parameters():
        torch.nn.init.normal_(p_, std=0.04)
        if "norm" in n_ and n_.endswith("weight"):
            p_.data.add_(1.0)
    p = {k: v.detach().clone() for k, v in layer.state_dict().items() if "relative_position_index" not in k}
    layer = layer.to(DEV).train()
    arena = FlatArena(layer)
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(1, H * H, C, generator=g)
    g1 = torch.randn(1, H * H, C, generator=g)
    g2 = torch.randn(1, (H // 2) ** 2, 2 * C, generator=g)

    xr = x0.clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    x_out, x_down, _, _ = OSW.basic_layer(xr, H, H, pr, "", 2, nH, ws, True)
    ((x_out * g1).sum() + (x_down * g2).sum()).backward()

    arena.zero_grad()
    x = x0.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_out, _, _, y_down, wh, ww = layer(x, H, H)
    assert (wh, ww) == (H // 2, H // 2)
    ((y_out.float() * g1.to(DEV)).sum() + (y_down.float() * g2.to(DEV)).sum()).backward()
    torch.cuda.synchronize()

    def close(a, b, frac, l2, what):
        err, sc = float((a - b).abs().max()), float(b.abs().max())
        r2 = float((a - b).double().norm() / b.double().norm().clamp(min=1e-30))
        print("%s: max err / scale %.2e, relative L2 %.2e" % (what, err / sc, r2))
        assert err <= frac * sc, (what, err, sc)
        assert r2 <= l2, (what, r2)
    # same budgets as the reference-golden BasicLayer test (bf16 GEMMs + bf16 attention operands through 2 blocks); the relative
    # L2 bounds sit at a few bf16 roundings (2^-9 each) accumulated over the ~12 stored tensors of a block pair
    close(y_out.float().cpu(), x_out.detach(), 0.03, 1e-2, "x_out")
    close(y_down.float().cpu(), x_down.detach(), 0.03, 1e-2, "x_down")
    close(x.grad.float().cpu(), xr.grad, 0.05, 2e-2, "dx")
    named = dict(layer.named_parameters())
    for k in ("blocks.1.attn.qkv.weight", "blocks.0.mlp.fc1.weight", "blocks.1.attn.relative_position_bias_table",
              "blocks.0.attn.proj.bias", "downsample.reduction.weight"):
        close(named[k].grad.float().cpu(), pr[k].grad, 0.06, 2e-2, k)


def test_swinL_training_forward_vs_assembled_oracle(monkeypatch):
    """tests/test_gpu_model.py::test_end_to_end_losses_vs_assembled_oracle on the registry-built Swin-L (L-22k-384,
    window 12) model at 384 px."""
    from tests.test_gpu_model import run_e2e_vs_oracle
    run_e2e_vs_oracle(monkeypatch, "L-22k-384", 384)
