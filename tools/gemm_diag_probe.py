"""Tile / stage-count matrix of the own GEMM on a few Swin-L shapes (development; kernel time from HIP events in libdgx).
usage: gemm_diag_probe.py [tiles] [modes]   e.g.  256x192,128x192  0,2,4"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _dev  # noqa: E402
_dev.apply_env()       # DGX_GEMM_LW / DGX_GEMM_TILE / DGX_WGRAD_LW ... of the calling script -> dgx_dev_set

from divergen_amd.layers import gemm_ops as G  # noqa: E402

shapes = [("s2.fc1", 8192, 3072, 768), ("s2.fc2", 8192, 768, 3072), ("s2.qkv", 10368, 2304, 768), ("s2.qkvd", 10368, 768, 2304),
          ("s2.proj", 10368, 768, 768), ("s3.fc2", 2048, 1536, 6144), ("s0.fc1", 131072, 768, 192), ("s1.fc1", 32768, 1536, 384),
          ("fc6", 1024, 1024, 12544), ("conv.p3", 32768, 256, 2304), ("conv.p6", 512, 256, 2304)]
tiles = sys.argv[1].split(",") if len(sys.argv) > 1 else ["256x192", "192x192", "128x192", "128x256", "128x128"]
modes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    fl = 2.0 * M * N * K
    for tile in tiles:
        if (N % 192 == 0) != tile.endswith("192"):
            continue
        for st in ("", "2"):
            if tile in ("256x192", "128x256") and st == "2":
                continue
            __import__("_dev").set_tile(tile)
            if st:
                os.environ["DGX_GEMM_STAGES"] = st
            else:
                os.environ.pop("DGX_GEMM_STAGES", None)
            row = []
            for m in modes:
                t = G.dev_time_us(x, w, b if m in (1, 2) else None, iters=30, mode=m if m else None) * 1e-6
                row.append("mode%d %6.1fus %5.0fTF" % (m, t * 1e6, fl / t / 1e12))
            print("%-8s %-8s st=%-4s | %s" % (name, tile, st or "dflt", " | ".join(row)), flush=True)
