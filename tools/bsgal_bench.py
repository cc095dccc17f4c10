"""BSGAL gradient-bank kernels vs the reference's torch formulation (custom_rcnn.py:1046-1086) on a Swin-L-sized arena."""
import sys

import torch

sys.path.insert(0, ".")
from divergen_amd.engine import bsgal as BG  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


class _A:
    def __init__(self, n):
        self.g = torch.zeros(n, device="cuda")


def main(n=197_000_000):
    g1, g2 = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    bank = BG.GradBank(_A(n), "MOMENTUM0.9")
    emb = torch.zeros(n, 1, device="cuda")

    def ref_update():
        emb.mul_(0.9)
        emb.add_(g1.unsqueeze(-1) * (1 - 0.9))

    def ref_sim():
        return (g1 * g2).sum() / (g1.norm() * g2.norm() + 1e-8)
    t_u, t_s = timeit(lambda: bank.update(g1, 3)), timeit(lambda: BG.grad_sim(g1, g2))
    r_u, r_s = timeit(ref_update), timeit(ref_sim)
    gb = n * 4 / 1e9
    print("n = %d floats (%.2f GB per vector)" % (n, gb))
    print("bank update : %.3f ms  (%.2f TB/s over 3 streams)   torch formulation %.3f ms" % (t_u, 3 * gb / t_u, r_u))
    print("similarity  : %.3f ms  (%.2f TB/s over 2 streams)   torch formulation %.3f ms" % (t_s, 2 * gb / t_s, r_s))


if __name__ == "__main__":
    main()
