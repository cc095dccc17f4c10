import sys, torch
sys.path.insert(0, ".")
from divergen_amd import _lib as L
lib = L.lib()
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (T, C, dt) in [(8192, 768, torch.bfloat16), (32768, 384, torch.bfloat16), (131072, 192, torch.float32), (2048, 1536, torch.bfloat16)]:
    x = torch.randn(T, C, device="cuda").to(dt); dy = torch.randn(T, C, device="cuda").bfloat16()
    g = torch.randn(C, device="cuda"); mean = x.float().mean(1).contiguous(); rstd = (x.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dx = torch.empty_like(x); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    nb = lib.dgx_layernorm_bwd_blocks(T); part = torch.empty(nb * 2 * C, device="cuda")
    us = t(lambda: lib.dgx_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(g), L.ptr(dx), L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(part), T, C, 0, 0, 0, 0, 0, L.dtype_code(x), L.stream()))
    by = T * C * (2 + 3 * x.element_size())
    print("T=%6d C=%4d %s blocks=%d  ln_bwd(+param reduce) %.1f us  %.2f TB/s" % (T, C, dt, nb, us, by / us / 1e6))
