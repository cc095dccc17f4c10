"""Which operand's coldness costs the own GEMM more than the library: rotate only A, only B, only C."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from divergen_amd.layers import gemm_ops as G
from tools.gemm_cold_probe import run
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K in [("s2.fc1", 8192, 3072, 768), ("s1.fc1", 32768, 1536, 384), ("s2.qkv", 10368, 2304, 768)]:
    n = 9
    A = [torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16) for _ in range(n)]
    B = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(n)]
    C = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(n)]
    big = torch.empty(300 << 20, dtype=torch.uint8, device="cuda")
    for label, fa, fb, fc in [("hot", 0, 0, 0), ("coldA", 1, 0, 0), ("coldB", 0, 1, 0), ("coldC", 0, 0, 1), ("all", 1, 1, 1)]:
        row = []
        for pfd in (0, 1):
            os.environ["DGX_GEMM_KROT"] = str(pfd)
            row.append("krot%d %6.1f" % (pfd, run(lambda i: G.gemm_nt(A[i * fa], B[i * fb], out=C[i * fc]), n)))
        tl = run(lambda i: torch.mm(A[i * fa], B[i * fb].t(), out=C[i * fc]), n)
        print("%-7s %-6s own %s  lib %6.1f" % (name, label, "  ".join(row), tl), flush=True)
