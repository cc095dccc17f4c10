import time, torch
dev="cuda"
def probe(label, fn, flops, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/n
    print("%-64s %8.1f us  %7.1f TF/s" % (label, dt*1e6, flops/dt/1e12))
for (M,K,N) in [(131072,192,576),(131072,192,768),(131072,768,192),(32768,384,1152),(32768,384,1536),(10368,768,2304),(8192,768,3072),(8192,3072,768)]:
    x = torch.randn(M,K,device=dev,dtype=torch.bfloat16); gy = torch.randn(M,N,device=dev,dtype=torch.bfloat16)
    fl = 2.0*M*K*N
    probe("M%d K%d N%d  TN gy.t()@x -> fp32" % (M,K,N), lambda: torch.mm(gy.t(), x, out_dtype=torch.float32), fl)
    probe("   same -> bf16", lambda: torch.mm(gy.t(), x), fl)
    probe("   explicit transposes + NT (fp32 out)", lambda: torch.mm(gy.t().contiguous(), x.t().contiguous().t(), out_dtype=torch.float32), fl)
    xt = x.t().contiguous(); gyt = gy.t().contiguous()
    probe("   NT only (pre-transposed)", lambda: torch.mm(gyt, xt.t(), out_dtype=torch.float32), fl)
    probe("   x.t()@gy (K,N) -> fp32", lambda: torch.mm(x.t(), gy, out_dtype=torch.float32), fl)
