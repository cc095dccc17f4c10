import sys, time, torch
sys.path.insert(0, ".")
from divergen_amd import layers as la
from divergen_amd import _lib as L
dev = "cuda"
def bench(B_, nH, ws=12, nW=1, iters=20):
    N = ws * ws
    qkv = (torch.randn(B_, N, 3 * nH * 32, device=dev)).to(torch.bfloat16).requires_grad_(True)
    table = torch.randn((2 * ws - 1) ** 2, nH, device=dev, requires_grad=True)
    region = torch.zeros(nW, N, dtype=torch.int8, device=dev) if nW > 1 else None
    go = torch.randn(B_, N, nH * 32, device=dev).to(torch.bfloat16)
    out = la.window_attention_core(qkv, table, region, nW, nH, ws, 32 ** -0.5)
    out.backward(go)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(iters):
        e[0].record()
        out = la.window_attention_core(qkv, table, region, nW, nH, ws, 32 ** -0.5)
        e[1].record()
        out.backward(go)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    tf, tb = tf / iters * 1e3, tb / iters * 1e3
    fl = B_ * nH * 4.0 * N * N * 32
    print("B_=%5d nH=%2d nW=%3d  fwd %7.1f us (%6.1f TF/s)  bwd %7.1f us (%6.1f TF/s)  units %d  bwd ns/unit/CU %.0f" % (
        B_, nH, nW, tf, fl / tf / 1e6, tb, 2.5 * fl / tb / 1e6, B_ * nH, tb * 1e3 * 256 / (B_ * nH)))
bench(968, 6); bench(968, 6); bench(242, 12); bench(72, 24); bench(18, 48)
