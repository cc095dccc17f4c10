"""Dev helper: the copy-paste compositor alone (no training stream beside it): per-kernel time of one image's composition at the
benchmark's geometry (1024^2, 8-16 objects, 19 pastes) under the torch profiler, and the bytes it has to move."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from torch.profiler import ProfilerActivity, profile
from bench import make_pastes
from divergen_amd import layers as la
from divergen_amd.data import synthetic_batch

dev = "cuda"
rng = np.random.default_rng(7)
for n_gt in (8, 12, 16):
    d = synthetic_batch(1, 1024, 1203, seed=1234 + n_gt, n_gt=n_gt, device=dev)[0]
    ps = la.pack_pastes(make_pastes(rng, 1024), dev)
    inst = d["instances"]
    img, gm, gb, gc = d["image"], inst.gt_masks.tensor.view(torch.uint8), inst.gt_boxes.tensor, inst.gt_classes
    for _ in range(3):
        la.copy_paste(img.clone(), gm, gb, gc, ps, lazy_masks=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            la.copy_paste(img.clone(), gm, gb, gc, ps, lazy_masks=True)
        torch.cuda.synchronize()
    H, W = img.shape[-2:]
    n, K = gm.shape[0], ps.K
    algo = (3 + 3 + 4 + 4 + 2 * n + K) * H * W      # image in/out, cover out/in, masks in/out, pasted masks out
    tot = 0.0
    print("n %d K %d: algorithmic bytes %.1f MB" % (n, K, algo / 1e6))
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
        if e.device_time_total > 0 and ("cp_" in e.key or "copy" in e.key.lower() or "fill" in e.key.lower() or e.device_time_total / 10 > 3):
            print("   %-70s %7.1f us x %.1f" % (e.key[:70], e.device_time_total / e.count, e.count / 10))
        tot += e.device_time_total / 10
    print("   all device time per composition %.1f us = %.2f TB/s of the algorithmic bytes" % (tot, algo / tot / 1e6))
