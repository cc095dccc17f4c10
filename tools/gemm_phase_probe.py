"""Per-phase cycle counts of one GEMM launch (development): s_memtime stamps written by thread 0 of every workgroup at
kernel entry / first tile landed / main loop done / tile staged / stores issued."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _dev  # noqa: E402
_dev.apply_env()       # DGX_GEMM_LW / DGX_GEMM_TILE / DGX_WGRAD_LW ... of the calling script -> dgx_dev_set

from divergen_amd import _lib as L  # noqa: E402
from divergen_amd.layers import gemm_ops as G  # noqa: E402

shapes = [("s2.fc1", 8192, 3072, 768), ("s2.fc2", 8192, 768, 3072), ("s2.proj", 10368, 768, 768), ("s2.qkv", 10368, 2304, 768)]
tiles = sys.argv[1].split(",") if len(sys.argv) > 1 else ["256x192", "128x192"]
diags = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
g = torch.Generator(device="cuda").manual_seed(0)
dbg = torch.zeros(8 * 4096, dtype=torch.int64, device="cuda")
setdbg = L.lib().dgx_dev_gemm_set_debug
setdbg.argtypes = [ctypes.c_void_p]
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    for tile, dg in [(t_, d_) for t_ in tiles for d_ in diags]:
        if tile.startswith("g"):                       # g256 / g192: the 4-wave 256 x BN kernel (gemm256.hip)
            os.environ["DGX_GEMM256"] = tile[1:]
        else:
            os.environ.pop("DGX_GEMM256", None)
            __import__("_dev").set_tile(tile)
        os.environ["DGX_GEMM_DIAG"] = str(dg)
        for _ in range(3):
            G.gemm_nt(x, w)
        torch.cuda.synchronize()
        dbg.zero_()
        setdbg(dbg.data_ptr())
        G.gemm_nt(x, w)
        torch.cuda.synchronize()
        setdbg(None)
        d = dbg.view(-1, 8).cpu()
        d = d[d[:, 0] > 0]
        t0 = d[:, 0].min()
        st = (d[:, 0] - t0).float()
        ph = [(d[:, i + 1] - d[:, i]).float() for i in range(4)]
        tot = (d[:, 4] - d[:, 0]).float()
        order = st.argsort()
        n = len(order)
        first, last = order[: min(256, n)], order[-min(256, n):]
        def m(v, idx):
            return float(v[idx].mean())
        NT = (K + 63) // 64
        span = float(d[:, 4].max() - t0)
        us = G.dev_time_us(x, w, None)
        print("   span %.0f ticks, %.1f us by HIP events -> %.0f ticks/us; last-starting 256 blocks start at %.0f" % (span, us, span / us, m(st, last)))
        allb = torch.arange(n)
        print("%-8s %-8s diag=%d blocks=%d | prologue %.0f mainloop %.0f (%.0f per K-tile) stage %.0f store %.0f total %.0f" % (
            name, tile, dg, n, m(ph[0], allb), m(ph[1], allb), m(ph[1], allb) / NT, m(ph[2], allb), m(ph[3], allb), m(tot, allb)), flush=True)
