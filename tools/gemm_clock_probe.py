"""Shader clock and per-shape time of the own GEMM inside a sustained, cache-cold stream of launches (development).

The eight Linear GEMMs of a Swin-L stage-2 block are launched round-robin for `--iters` rounds on rotating operand copies
(every operand comes from HBM, as in the training step); HIP events give the time per shape, and the kernel's debug stamps
(shader-clock counter next to the constant 100 MHz counter) give the clock the CUs actually ran at."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from divergen_amd import _lib as L  # noqa: E402
from divergen_amd.layers import gemm_ops as G  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--copies", type=int, default=6)
ap.add_argument("--hot", action="store_true", help="one copy of everything (what the per-shape probe measures)")
a = ap.parse_args()
shapes = [("qkv.fwd", 10368, 2304, 768), ("proj.fwd", 10368, 768, 768), ("fc1.fwd", 8192, 3072, 768), ("fc2.fwd", 8192, 768, 3072),
          ("fc1.dgrad", 8192, 768, 3072), ("fc2.dgrad", 8192, 3072, 768), ("proj.dgrad", 10368, 768, 768), ("qkv.dgrad", 10368, 768, 2304)]
g = torch.Generator(device="cuda").manual_seed(0)
nc = 1 if a.hot else a.copies
ops = []
for name, M, N, K in shapes:
    A = [torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16) for _ in range(nc)]
    B = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(nc)]
    C = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(nc)]
    ops.append((A, B, C))
dbg = torch.zeros(8 * 4096, dtype=torch.int64, device="cuda")
setdbg = L.lib().dgx_dev_gemm_set_debug
setdbg.argtypes = [ctypes.c_void_p]
ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in shapes] for _ in range(a.iters)]
for it in range(a.iters):
    for si, (A, B, C) in enumerate(ops):
        k = it % nc
        last = it == a.iters - 1
        if last:
            dbg.zero_()
            setdbg(dbg.data_ptr())
        ev[it][si][0].record()
        G.gemm_nt(A[k], B[k], out=C[k])
        ev[it][si][1].record()
        if last:
            torch.cuda.synchronize()
            setdbg(None)
            d = dbg.view(-1, 8).cpu()
            d = d[d[:, 0] > 0]
            cyc = (d[:, 4] - d[:, 0]).double()
            rt = (d[:, 6] - d[:, 5]).double()
            ok = rt > 0
            mhz = float((cyc[ok] / rt[ok]).mean()) * 100.0
            print("%-10s blocks %4d  shader clock %.0f MHz  (block: %.0f cycles = %.1f us)" % (
                shapes[si][0], len(d), mhz, float(cyc.mean()), float(rt.mean()) / 100.0), flush=True)
torch.cuda.synchronize()
tot = 0.0
for si, (name, M, N, K) in enumerate(shapes):
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in (ev[it][si] for it in range(5, a.iters - 1)))
    med = ts[len(ts) // 2]
    tot += med
    print("%-10s %6d %5d %5d  median %6.1f us  %5.0f TF/s" % (name, M, N, K, med, 2.0 * M * N * K / med / 1e6))
print("block total %.1f us (%s operands)" % (tot, "hot" if a.hot else "cold"))
