import sys, time, torch
sys.path.insert(0, ".")
from divergen_amd.layers.linear_ops import wgrad_into
dev="cuda"
def probe(label, fn, flops, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/n
    print("%-50s %8.1f us  %7.1f TF/s" % (label, dt*1e6, flops/dt/1e12))
for (M,K,N) in [(131072,192,576),(131072,192,768),(131072,768,192),(32768,384,1152),(32768,1536,384),(10368,768,2304),(8192,768,3072),(8192,3072,768),(2048,1536,6144),(32768,2304,256)]:
    x = torch.randn(M,K,device=dev,dtype=torch.bfloat16); gy = torch.randn(M,N,device=dev,dtype=torch.bfloat16)
    acc = torch.zeros(N,K,device=dev)
    fl = 2.0*M*K*N
    probe("M%d K%d N%d hipblaslt" % (M,K,N), lambda: torch.addmm(acc, gy.t(), x, out_dtype=torch.float32, out=acc), fl)
    probe("   dgx_linear_wgrad", lambda: wgrad_into(acc, gy, x), fl)
