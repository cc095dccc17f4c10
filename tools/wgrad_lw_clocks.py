"""Phase clocks of workgroup 0 of the loader-wave weight-gradient kernel (development build: tools/build_dev_wl.sh, DGX_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("DGX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "divergen_amd", "csrc", "_obj", "libdgx_dev.so"))
import torch
from divergen_amd import _lib as L
from divergen_amd.layers.linear_ops import wgrad_grouped
dev = "cuda"
def block(C, Mw, Mt):
    mk = lambda M, Nn, Kk: (torch.zeros(Nn, Kk, device=dev), (torch.randn(M, Nn, device=dev) * 0.3).bfloat16(), (torch.randn(M, Kk, device=dev) * 0.3).bfloat16(), torch.zeros(Nn, device=dev) if os.environ.get("PROBE_BIAS", "0") == "1" else None)
    return [mk(Mw, 3 * C, C), mk(Mw, C, C), mk(Mt, 4 * C, C), mk(Mt, C, 4 * C)]
probs = [p for _ in range(7) for p in block(768, 10368, 8192)]
raw = ctypes.CDLL(os.environ["DGX_LIB"])
import numpy as np
out = (ctypes.c_ulonglong * (3 * 1024 * 4))()
for _ in range(2): wgrad_grouped(probs, beta=float(os.environ.get("PROBE_BETA", "1")))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); wgrad_grouped(probs, beta=float(os.environ.get("PROBE_BETA", "1"))); e1.record(); torch.cuda.synchronize(); raw.dgx_dev_wl_clocks(out, 0)
us = e0.elapsed_time(e1) * 1e3
st = np.array(list(out), dtype=np.int64).reshape(3, 1024, 4)
for wi, tag in enumerate(("MFMA wave 0 (group 0)", "MFMA wave 4 (group 1)")):
    s_ = st[wi]; n = int((s_[:, 2] > 0).sum()); s_ = s_[:n]
    print("%-24s" % tag, "first 2 halves %.0f  12 halves %.0f = %.0f cycles per K-tile (%d full rounds of the rings)" % (
        np.median(s_[:, 1] - s_[:, 0]), np.median(s_[:, 2] - s_[:, 0]), np.median(s_[:, 2] - s_[:, 0]) / 6.0, n))
s_ = st[2]; n = int((s_[:, 3] > 0).sum()); s_ = s_[:n]
print("loader wave 8            wait for the granules %.0f  barrier %.0f  issue 8 loads %.0f  bias %.0f | median half %.0f cycles (%d halves)" % (
    np.median(s_[1:, 0] - s_[:-1, 3]), np.median(s_[:, 1] - s_[:, 0]), np.median(s_[:, 2] - s_[:, 1]), np.median(s_[:, 3] - s_[:, 2]), np.median(s_[1:, 0] - s_[:-1, 0]), n))
out2 = (ctypes.c_ulonglong * (3 * 1024 * 4))()
raw.dgx_dev_wl_clocks(out2, 1)
it = np.array(list(out2)[:96], dtype=np.int64).reshape(3, 8, 4)
for wi in (0, 1):      # row j >= 1: [after barrier E of item j-1, after its read-out, after barrier P of item j]
    r = it[wi]
    print("wave %d, item boundaries:" % (4 * wi), " | ".join("read-out %d, then %d to the next item's first barrier" % (r[j, 1] - r[j, 0], r[j, 2] - r[j, 1]) for j in range(1, 4)))
c = list(out2)[92:96]
print("workgroup 0 lifetime: %d shader cycles in %.1f us (100 MHz counter) = %.0f MHz" % (c[2] - c[0], (c[3] - c[1]) / 100.0, (c[2] - c[0]) / ((c[3] - c[1]) / 100.0)))
