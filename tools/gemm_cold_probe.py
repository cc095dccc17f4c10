"""Own GEMM vs library GEMM with COLD operands: every launch reads a different copy of A and B (rotating through > 256 MiB, the
size of the memory-side cache), as inside a training step where activations come from HBM.  usage: gemm_cold_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from divergen_amd.layers import gemm_ops as G  # noqa: E402

shapes = [("s2.fc1", 8192, 3072, 768), ("s2.fc1d", 8192, 768, 3072), ("s2.qkv", 10368, 2304, 768), ("s2.qkvd", 10368, 768, 2304),
          ("s2.proj", 10368, 768, 768), ("s1.fc1", 32768, 1536, 384), ("s1.fc1d", 32768, 384, 1536)]
g = torch.Generator(device="cuda").manual_seed(0)


def run(fn, n, iters=4):
    """n launches (one per buffer copy) captured into a hipGraph and replayed: no host gaps between kernels."""
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            for i in range(n):
                fn(i)
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        gr.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * n) * 1e3


if __name__ != "__main__":
    shapes = []
for name, M, N, K in shapes:
    per = (M * K + N * K + M * N) * 2
    ncopy = max(2, int(600e6 // per) + 1)
    A = [torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16) for _ in range(ncopy)]
    B = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16) for _ in range(ncopy)]
    C = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(ncopy)]
    fl = 2.0 * M * N * K
    t_own_cold = run(lambda i: G.gemm_nt(A[i], B[i], out=C[i]), ncopy)
    os.environ["DGX_GEMM_NT"] = "1"
    t_2wg_cold = run(lambda i: G.gemm_nt(A[i], B[i], out=C[i]), ncopy)
    os.environ.pop("DGX_GEMM_NT")
    t_2wgb_cold = 0.0
    t_lib_cold = run(lambda i: torch.mm(A[i], B[i].t(), out=C[i]), ncopy)
    t_own_hot = run(lambda i: G.gemm_nt(A[0], B[0], out=C[0]), ncopy)
    t_lib_hot = run(lambda i: torch.mm(A[0], B[0].t(), out=C[0]), ncopy)
    print("%-8s copies=%2d | cold: nt %6.1f (%3.1f) own %6.1f us (%4.0f TF) lib %6.1f us (%4.0f TF) | hot: own %6.1f us lib %6.1f us" % (
        name, ncopy, t_2wg_cold, t_2wgb_cold, t_own_cold, fl / t_own_cold / 1e6, t_lib_cold, fl / t_lib_cold / 1e6, t_own_hot, t_lib_hot), flush=True)
    del A, B, C
