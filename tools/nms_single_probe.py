"""Dev helper: the single-image NMS (dgx_nms: mask + sweep, the inference path's call) timed with HIP events at a few sizes and densities."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from divergen_amd import layers as la  # noqa: E402

g = torch.Generator().manual_seed(3)
for n, spread in ((1000, 300.0), (1000, 2000.0), (4000, 600.0), (4000, 4000.0)):
    xy = torch.rand(n, 2, generator=g) * spread
    wh = torch.rand(n, 2, generator=g) * 60 + 4
    boxes = torch.cat([xy, xy + wh], 1).cuda()
    scores = torch.rand(n, generator=g).cuda()
    for _ in range(3):
        keep = la.nms(boxes, scores, 0.5)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        keep = la.nms(boxes, scores, 0.5)
    b.record()
    torch.cuda.synchronize()
    print("n %5d spread %6.0f: kept %5d  %.1f us per call (sort + mask + sweep + compaction)" % (n, spread, keep.numel(), a.elapsed_time(b) * 1e3 / 20))
