"""Phase clocks of workgroup 0 of the loader-wave weight-gradient kernel (development build: tools/build_dev_wl.sh, DGX_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("DGX_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "divergen_amd", "csrc", "_obj", "libdgx_dev.so"))
import torch
from divergen_amd import _lib as L
from divergen_amd.layers.linear_ops import wgrad_grouped
dev = "cuda"
def block(C, Mw, Mt):
    mk = lambda M, Nn, Kk: (torch.zeros(Nn, Kk, device=dev), (torch.randn(M, Nn, device=dev) * 0.3).bfloat16(), (torch.randn(M, Kk, device=dev) * 0.3).bfloat16(), None)
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
names = [["read + wait", "barrier", "mfma + wait", "barrier"], ["read + wait", "barrier", "mfma + wait", "barrier"], ["issue B (+bias)", "barrier", "issue A + wait tile", "barrier"]]
for wi, tag in enumerate(("MFMA wave 0 (group 0)", "MFMA wave 4 (group 1)", "loader wave 8")):
    s_ = st[wi]
    n = int((s_[:, 3] > 0).sum())
    s_ = s_[:n]
    seg = np.stack([s_[1:, 0] - s_[:-1, 3], s_[1:, 1] - s_[1:, 0], s_[1:, 2] - s_[1:, 1], s_[1:, 3] - s_[1:, 2]], 1)
    med = np.median(seg, 0)
    print("%-24s" % tag, "  ".join("%s %.0f" % (names[wi][i], med[i]) for i in range(4)), " | median K-tile %.0f cycles (%d K-tiles)" % (np.median(s_[1:, 3] - s_[:-1, 3]), n))
s_ = st[0]; n = int((s_[:, 3] > 0).sum()); s_ = s_[:n]
tot = s_[-1, 3] - s_[0, 0]
gaps = np.sort(s_[1:, 0] - s_[:-1, 3])[::-1][:6]
print("workgroup 0: %d cycles first to last stamp; launch %.0f us by events -> >= %.0f MHz if it spans the launch; largest read gaps (item boundaries: read-out + prologue): %s" % (tot, us, tot / us, gaps.tolist()))
out2 = (ctypes.c_ulonglong * (3 * 1024 * 4))()
raw.dgx_dev_wl_clocks(out2, 1)
it = np.array(list(out2)[:96], dtype=np.int64).reshape(3, 8, 4)
for wi in (0, 1):
    r = it[wi]
    print("wave %d items:" % (4 * wi), " | ".join("read-out %d, then to barrier #0 of the next item %d" % (r[k, 1] - r[k, 0], r[k + 1, 2] - r[k, 1]) for k in range(3)))
c = list(out2)[92:96]
print("workgroup 0 lifetime: %d shader cycles in %.1f us (100 MHz counter) = %.0f MHz" % (c[2] - c[0], (c[3] - c[1]) / 100.0, (c[2] - c[0]) / ((c[3] - c[1]) / 100.0)))
