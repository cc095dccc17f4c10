"""Which own GEMM form is fastest for the compact stage-2 proj shape (8192 x 768 x 768: exactly 256 tiles of 128 x 192, one per CU,
nothing to overlap a tile's prologue / read-out with)?  Back-to-back kernel time per forced tile / form (dgx_dev_set)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from divergen_amd import _lib as L  # noqa: E402
from divergen_amd.layers import gemm_ops as G  # noqa: E402

lib = L.lib()
torch.manual_seed(0)
for (M, N, K) in ((8192, 768, 768), (8192, 2304, 768), (8192, 768, 2304), (2048, 1536, 1536), (6272, 768, 768)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    print("shape", M, N, K, "GFLOP %.1f" % (2.0 * M * N * K / 1e9))
    for lw in (-1, 0, 1):
        for tile in (0, 128192, 128128, 128256, 192192, 256192, 192256):
            for twg in ((-1, 0) if lw == 0 else (-1,)):
                lib.dgx_dev_set(b"reset", 0)
                lib.dgx_dev_set(b"gemm_lw", lw)
                lib.dgx_dev_set(b"gemm_tile", tile)
                lib.dgx_dev_set(b"gemm_2wg", twg)
                try:
                    us = G.dev_time_us(a, b, bias, iters=40)
                except Exception as e:
                    us = float("nan")
                bm = ctypes_bm = None
                print("   lw %2d tile %6d 2wg %2d : %7.1f us  %6.0f TF/s" % (lw, tile, twg, us, 2.0 * M * N * K / us / 1e6))
lib.dgx_dev_set(b"reset", 0)
