"""Dev helper: per-section HOST issue time (no syncs added) next to the GPU time between the same points."""
import sys
import time

import torch

sys.path.insert(0, ".")
from divergen_amd.config import get_cfg
from divergen_amd.data import synthetic_batch
from divergen_amd.modeling import build_model
from divergen_amd.solver import build_optimizer
from divergen_amd.utils.events import EventStorage

cfg = get_cfg()
cfg.merge_from_file("configs/DiverGen_swinL.yaml")
cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH",
                     "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json"])
torch.manual_seed(42)
model = build_model(cfg).train()
opt = build_optimizer(cfg, model)
batch = synthetic_batch(2, 1024, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
names = ["preproc", "backbone+fpn", "proposal_gen", "roi_heads", "loss_sum", "backward", "optimizer"]
acc_h = [0.0] * len(names)
acc_g = [0.0] * len(names)
N = 8
with EventStorage(0):
    for it in range(N + 3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        th = []

        def mark(i):
            th.append(time.perf_counter())
            ev[i].record()
        torch.cuda.synchronize()
        mark(0)
        opt.zero_grad()
        images = model.preprocess_image(batch)
        gt = [x["instances"] for x in batch]
        mark(1)
        feats = model._features(images)
        mark(2)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            props, pl = model.proposal_generator(images, feats, gt)
            mark(3)
            props, dl = model.roi_heads(images, feats, props, gt)
            mark(4)
        total = sum(pl.values()) + sum(dl.values())
        mark(5)
        total.backward()
        mark(6)
        opt.step()
        mark(7)
        torch.cuda.synchronize()
        if it >= 3:
            for i in range(len(names)):
                acc_h[i] += (th[i + 1] - th[i]) * 1e3 / N
                acc_g[i] += ev[i].elapsed_time(ev[i + 1]) / N
print("%-14s %9s %9s" % ("section", "host ms", "gpu ms"))
for n, h, g in zip(names, acc_h, acc_g):
    print("%-14s %9.2f %9.2f" % (n, h, g))
print("%-14s %9.2f %9.2f" % ("sum", sum(acc_h), sum(acc_g)))
