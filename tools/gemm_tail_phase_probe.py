"""Per-phase cycle counts of the fused-tail GEMMs of the HBM-bound Swin stages (K = 192 / 384) on the two-workgroup form: prologue
(first loads landed), main loop, staging, read-out -- averaged over the workgroups of one launch."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from divergen_amd import _lib as L  # noqa: E402
from divergen_amd.layers import gemm_ops as G  # noqa: E402

# optional: "key=value" dev settings on the command line (e.g. gemm_lw=1 gemm_tile=128192)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    assert L.lib().dgx_dev_set(k.encode(), int(v)) == 0, kv
g = torch.Generator(device="cuda").manual_seed(0)
setdbg = L.lib().dgx_dev_gemm_set_debug
setdbg.argtypes = [ctypes.c_void_p]
for name, M, N, K, mode in [("s0.fc1+gelu", 131072, 768, 192, 2), ("s0.fc2dgrad*gelu'", 131072, 768, 192, 4), ("s0.bias", 131072, 768, 192, 1),
                            ("s1.fc1+gelu", 32768, 1536, 384, 2), ("s2.fc1+gelu", 8192, 3072, 768, 2), ("s2.fc2dgrad", 8192, 3072, 768, 4)]:
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    f1 = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    run = {1: lambda: G.gemm_nt(x, w, b), 2: lambda: G.gemm_bias_gelu(x, w, b), 4: lambda: G.gemm_gelu_grad(x, w, f1)}[mode]
    nblk = ((M + 127) // 128) * ((N + 191) // 192)
    dbg = torch.zeros(8 * (nblk + 64), dtype=torch.int64, device="cuda")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    setdbg(dbg.data_ptr())
    run()
    torch.cuda.synchronize()
    setdbg(None)
    bm, bn, sp = L.c_i(), L.c_i(), L.c_i()
    form = L.lib().dgx_gemm_last_form(bm, bn, sp)
    d = dbg.view(-1, 8).cpu()
    d = d[d[:, 0] > 0]
    ph = [(d[:, i + 1] - d[:, i]).float().mean().item() for i in range(4)]
    span = float(d[:, 4].max() - d[:, 0].min())
    clk = (d[:, 4] - d[:, 0]).float().sum().item() / max((d[:, 6] - d[:, 5]).float().sum().item(), 1.0) * 100.0     # MHz: shader ticks per 100 MHz tick
    byts = 2.0 * (M * K + N * K) + (4.0 if mode in (2,) else 2.0) * M * N + (2.0 * M * N if mode == 4 else 0.0)
    print("%-18s %7dx%5dx%4d form %d %dx%d | %6.1f us  %.2f TB/s | blocks %5d  prologue %6.0f  mainloop %6.0f  stage %6.0f  store %6.0f  total %6.0f cycles/tile, kernel span %.0f cycles, ~%.0f MHz"
          % (name, M, N, K, form, bm.value, bn.value, us, byts / us / 1e6, len(d), ph[0], ph[1], ph[2], ph[3], sum(ph), span, clk), flush=True)
