"""Dev probe: the grouped tower convolution (five FPN levels of 2 x 1024^2, Cin = Cout = 256) and single-image convolutions at the
mask-head / tower row counts, back to back; run twice with DGX_GEMM_192x256=0 / 1 (tools: A/B of the 192 x 256 tile)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from divergen_amd import _lib as L  # noqa: E402

lib = L.lib()
DEV = "cuda"
C = 256
w = (torch.randn(C, 9 * C, device=DEV) * 0.05).to(torch.bfloat16)
b = torch.randn(C, device=DEV).to(torch.bfloat16)


def pad(x):
    n, h, w_, c = x.shape
    xp = torch.empty(int(lib.dgx_conv3x3_pad_rows(n, h, w_)), c, dtype=torch.bfloat16, device=DEV)
    L.check(lib.dgx_conv3x3_pad(L.ptr(x), L.ptr(xp), n, h, w_, c, L.stream()), "pad")
    return xp


def timeit(fn, it=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


shapes = [(2, 128, 128), (2, 64, 64), (2, 32, 32), (2, 16, 16), (2, 8, 8)]
xs = [torch.randn(n, h, w_, C, device=DEV).to(torch.bfloat16) for n, h, w_ in shapes]
xps = [pad(x) for x in xs]
ys = [torch.empty(n, h, w_, C, dtype=torch.bfloat16, device=DEV) for n, h, w_ in shapes]
items = (L.ConvItem * len(xs))()
for i, ((n, h, w_), xp, y) in enumerate(zip(shapes, xps, ys)):
    items[i].xpad, items[i].y, items[i].N, items[i].H, items[i].W = L.ptr(xp), L.ptr(y), n, h, w_
us = timeit(lambda: L.check(lib.dgx_conv3x3_gemm_multi(items, len(xs), L.ptr(w), L.ptr(b), C, C, 0, L.stream()), "multi"))
M = sum(n * h * w_ for n, h, w_ in shapes)
print("grouped tower  M %6d: %7.1f us  %5.0f TF/s" % (M, us, 2.0 * M * C * 9 * C / us / 1e6))
for n, h, w_ in [(192, 14, 14), (128, 14, 14), (2, 128, 128), (256, 14, 14), (64, 28, 28)]:
    x = torch.randn(n, h, w_, C, device=DEV).to(torch.bfloat16)
    xp = pad(x)
    y = torch.empty(n, h, w_, C, dtype=torch.bfloat16, device=DEV)
    us = timeit(lambda: L.check(lib.dgx_conv3x3_gemm(L.ptr(xp), L.ptr(w), L.ptr(b), L.ptr(y), n, h, w_, C, C, 1, None, 0, L.stream()), "conv"))
    M = n * h * w_
    print("single         M %6d: %7.1f us  %5.0f TF/s" % (M, us, 2.0 * M * C * 9 * C / us / 1e6))
