import time, torch, sys
dev = "cuda"
def probe(label, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    t_cpu = (time.perf_counter()-t)/n
    torch.cuda.synchronize(); t_all=(time.perf_counter()-t)/n
    print("%-50s cpu/launch %.1f us   wall %.1f us" % (label, t_cpu*1e6, t_all*1e6))
for lib in ("default", "cublas", "cublaslt"):
    if lib != "default":
        torch.backends.cuda.preferred_blas_library(lib)
    print("== blas:", lib, torch.backends.cuda.preferred_blas_library())
    M, C, Co = 32768, 256, 256
    col = torch.randn(M, 9*C, device=dev, dtype=torch.bfloat16); g2 = torch.randn(M, Co, device=dev, dtype=torch.bfloat16)
    wm = torch.randn(9*C, Co, device=dev, dtype=torch.bfloat16); b = torch.randn(Co, device=dev, dtype=torch.bfloat16)
    probe("conv fwd addmm (M,2304)@(2304,256)", lambda: torch.addmm(b, col, wm))
    probe("conv dcol g2@wm.t() (M,256)@(256,2304)", lambda: g2 @ wm.t())
    probe("conv wgrad col.t()@g2 (2304,M)@(M,256)", lambda: col.t() @ g2)
    x = torch.randn(131072, 192, device=dev, dtype=torch.bfloat16); w = torch.randn(576, 192, device=dev, dtype=torch.bfloat16)
    gy = torch.randn(131072, 576, device=dev, dtype=torch.bfloat16)
    probe("qkv fwd linear (131072,192)x(576,192)^T", lambda: torch.nn.functional.linear(x, w))
    probe("qkv dgrad gy@w", lambda: gy @ w)
    probe("qkv wgrad gy.t()@x", lambda: gy.t() @ x)
    x2 = torch.randn(8192, 768, device=dev, dtype=torch.bfloat16); w2 = torch.randn(3072, 768, device=dev, dtype=torch.bfloat16)
    gy2 = torch.randn(8192, 3072, device=dev, dtype=torch.bfloat16)
    probe("fc1 s2 fwd (8192,768)x(3072,768)^T", lambda: torch.nn.functional.linear(x2, w2))
    probe("fc1 s2 dgrad", lambda: gy2 @ w2)
    probe("fc1 s2 wgrad", lambda: gy2.t() @ x2)
