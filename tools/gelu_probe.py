import sys, torch
sys.path.insert(0, ".")
from divergen_amd import _lib as L
M, N = 8192, 3072
x = torch.randn(M, N, device="cuda").bfloat16(); dy = torch.randn(M, N, device="cuda").bfloat16()
dx = torch.empty_like(x); bg = torch.zeros(N, device="cuda")
lib = L.lib()
ws = torch.empty(int(lib.dgx_gelu_bwd_workspace_bytes(M, N)), dtype=torch.uint8, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print("fused gelu_bwd+colsum %.1f us" % t(lambda: lib.dgx_gelu_bwd_colsum(L.ptr(dy), L.ptr(x), L.ptr(dx), L.ptr(bg), M, N, 1.0, L.ptr(ws), L.stream())))
print("fused gelu_bwd only   %.1f us" % t(lambda: lib.dgx_gelu_bwd_colsum(L.ptr(dy), L.ptr(x), L.ptr(dx), None, M, N, 1.0, L.ptr(ws), L.stream())))
print("torch gelu_backward   %.1f us" % t(lambda: torch.ops.aten.gelu_backward(dy, x)))
y = torch.empty_like(x)
print("dgx gelu_fwd          %.1f us" % t(lambda: lib.dgx_gelu_fwd(L.ptr(x), L.ptr(y), x.numel(), L.stream())))
print("torch gelu            %.1f us" % t(lambda: torch.nn.functional.gelu(x)))
