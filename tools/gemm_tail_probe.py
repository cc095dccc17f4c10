"""Dev probe: what the fused tails cost on ONE shape, back to back on hot operands: plain, bias, bias + GELU (two tensors out),
x GELU'(aux) and x ReLU'(aux) (same traffic as GELU', trivial arithmetic): the difference of the last two is the arithmetic of
the GELU' tail, the difference ReLU' - plain its extra traffic."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from divergen_amd.layers import gemm_ops as G  # noqa: E402


def t(fn, it=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


for M, N, K in [(8192, 3072, 768), (131072, 768, 192), (32768, 1536, 384)]:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    aux = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    fl = 2.0 * M * N * K
    rows = [("plain", lambda: G.gemm_nt(a, b)), ("bias", lambda: G.gemm_nt(a, b, bias)), ("plain", lambda: G.gemm_nt(a, b)), ("bias0", lambda: G.gemm_nt(a, b, torch.zeros_like(bias))), ("bias+gelu", lambda: G.gemm_bias_gelu(a, b, bias)),
            ("x gelu'", lambda: G.gemm_gelu_grad(a, b, aux)), ("x relu'", lambda: G.gemm_relu_grad(a, b, aux))]
    print("M %d N %d K %d" % (M, N, K))
    for name, fn in rows:
        us = t(fn)
        print("  %-10s %7.1f us  %5.0f TF/s" % (name, us, fl / us / 1e6))
