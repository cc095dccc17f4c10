"""Dev helper: what the fused tails cost on the stage-2 MLP shape: 8192 x 3072 x 768 on the two-workgroup form with a plain bf16 output
(mode 0), bias (1), bias + GELU with two outputs (2), GELU' with a read operand (4); and the same for the K = 192 shape of stage 0."""
import sys
import torch
sys.path.insert(0, ".")
from divergen_amd import _lib as L
from divergen_amd.layers import gemm_ops as G

g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K in ((8192, 3072, 768), (131072, 768, 192), (10368, 2304, 768)):
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    f1 = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    for form in ("default", "2wg", "lw"):
        L.lib().dgx_dev_set(b"reset", 0)
        if form == "2wg":
            L.lib().dgx_dev_set(b"gemm_lw", 0); L.lib().dgx_dev_set(b"gemm_2wg", 1)
        elif form == "lw":
            L.lib().dgx_dev_set(b"gemm_lw", 1)
        for name, run in (("plain", lambda: G.gemm_nt(x, w)), ("bias", lambda: G.gemm_nt(x, w, b)), ("bias+gelu (2 outputs)", lambda: G.gemm_bias_gelu(x, w, b)),
                          ("gelu' (reads f1)", lambda: G.gemm_gelu_grad(x, w, f1))):
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            bm, bn, sp = L.c_i(), L.c_i(), L.c_i()
            fm = L.lib().dgx_gemm_last_form(bm, bn, sp)
            us = e0.elapsed_time(e1) * 50.0
            print("%6d x %4d x %4d %-8s %-24s form %d %dx%d  %6.1f us  %5.0f TF/s" % (M, N, K, form, name, fm, bm.value, bn.value, us, 2.0 * M * N * K / us / 1e6), flush=True)
L.lib().dgx_dev_set(b"reset", 0)
