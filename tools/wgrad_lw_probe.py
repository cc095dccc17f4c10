"""Grouped weight-gradient launches of Swin-L stages 2 / 3 (1024^2, batch 2): the persistent loader-wave form (wgrad_lw.hip) against
the split-M 256x256 form (wgrad256.hip), timed with events over back-to-back launches; DGX_WGRAD_LW=0 forces the second.
usage: python tools/wgrad_lw_probe.py [blocks ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _dev  # noqa: E402
_dev.apply_env()       # DGX_GEMM_LW / DGX_GEMM_TILE / DGX_WGRAD_LW ... of the calling script -> dgx_dev_set

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from divergen_amd.layers.linear_ops import wgrad_grouped

dev = "cuda"
BETA = float(os.environ.get("PROBE_BETA", "1"))
BIAS = os.environ.get("PROBE_BIAS", "1") == "1"
def block(C, Mw, Mt):
    mk = lambda M, Nn, Kk: (torch.zeros(Nn, Kk, device=dev), (torch.randn(M, Nn, device=dev) * 0.3).bfloat16(), (torch.randn(M, Kk, device=dev) * 0.3).bfloat16(),
                            torch.zeros(Nn, device=dev) if BIAS else None)
    return [mk(Mw, 3 * C, C), mk(Mw, C, C), mk(Mt, 4 * C, C), mk(Mt, C, 4 * C)]

def run(label, probs, per=12):
    fl = sum(2.0 * p[1].shape[0] * p[1].shape[1] * p[2].shape[1] for p in probs)
    def go():
        if os.environ.get("DGX_WGRAD_LW", "1") == "0":
            for i in range(0, len(probs), per):
                wgrad_grouped(probs[i:i + per], beta=BETA)
        else:
            wgrad_grouped(probs, beta=BETA)
    for _ in range(3): go()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    a.record()
    for _ in range(n): go()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print("%-28s %3d problems  %8.1f us  %7.1f TF/s" % (label, len(probs), ms * 1e3, fl / ms / 1e9), flush=True)

nb = [int(v) for v in sys.argv[1:]] or [7, 4, 6]
for k in nb:
    run("stage 2 x %d blocks" % k, [p for _ in range(k) for p in block(768, 10368, 8192)])
run("stage 3 x 2 blocks", [p for _ in range(2) for p in block(1536, 2592, 2048)])
