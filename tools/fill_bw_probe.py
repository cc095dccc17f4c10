import torch
x = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")   # 1 GB
y = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: x.zero_()); print("fill 1 GB: %.3f ms = %.2f TB/s written" % (ms, 1.0737 / ms))
ms = t(lambda: y.copy_(x)); print("copy 1 GB: %.3f ms = %.2f TB/s read + %.2f TB/s written" % (ms, 1.0737 / ms, 1.0737 / ms))
ms = t(lambda: torch.add(x, 1.0, out=y)); print("add  1 GB: %.3f ms = %.2f TB/s each way" % (ms, 1.0737 / ms))
s = x[:64 * 1024 * 1024]
ms = t(lambda: s.zero_()); print("fill 256 MB: %.3f ms = %.2f TB/s" % (ms, 0.2684 / ms))
