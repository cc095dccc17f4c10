"""Dev helper: host issue time next to GPU time for the sections of a step around the RoI heads' device->host read, with the
proposal generator's backward run early (meta_arch/custom_rcnn.py early_proposal_backward) or late.  No profiler: perf_counter
stamps + HIP events at the same points, and a gradient hook on the FPN outputs for "heads backward issued".
   python tools/heads_host_probe.py [early|late] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from divergen_amd.config import get_cfg  # noqa: E402
from divergen_amd.data import synthetic_batch  # noqa: E402
from divergen_amd.modeling import build_model  # noqa: E402
from divergen_amd.engine import total_loss  # noqa: E402
from divergen_amd.solver import build_optimizer  # noqa: E402
from divergen_amd.utils.events import EventStorage  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "early"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs/DiverGen_swinL.yaml"))
cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", os.path.join(ROOT, "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json")])
torch.manual_seed(42)
model = build_model(cfg).train()
opt = build_optimizer(cfg, model)
batch = synthetic_batch(2, 1024, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
names = ["backbone+fpn fwd", "proposal gen fwd", "early bwd (issue)", "wait for counts", "roi heads fwd (rest)",
         "loss sum", "bwd: start -> FPN grads", "bwd: rest", "optimizer"]
acc_h = dict.fromkeys(names, 0.0)
acc_g = dict.fromkeys(names, 0.0)
stamps = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    stamps.append((name, time.perf_counter(), e))


fired = []


def grad_hook(g):
    if not fired:
        fired.append(1)
        mark("bwd: start -> FPN grads")
    return g


def after_backbone(_m, _i, out):
    mark("backbone+fpn fwd")
    for f in out.values():
        if f.requires_grad:
            f.register_hook(grad_hook)


model.backbone.register_forward_hook(after_backbone)
model.proposal_generator.register_forward_hook(lambda *_a: mark("proposal gen fwd"))
model.roi_heads.register_forward_hook(lambda *_a: mark("roi heads fwd (rest)"))
orig_sync = torch.cuda.Event.synchronize


def sync(self):          # the sampler's wait for the counts: the early backward has just been issued
    mark("early bwd (issue)")
    orig_sync(self)
    mark("wait for counts")


model.early_proposal_backward = mode == "early"
with EventStorage(0):
    for it in range(N + 4):
        stamps.clear()
        del fired[:]
        torch.cuda.synchronize()
        mark(None)
        opt.zero_grad()
        torch.cuda.Event.synchronize = sync
        try:
            losses = model(batch)
        finally:
            torch.cuda.Event.synchronize = orig_sync
        total = total_loss(losses)
        mark("loss sum")
        total.backward()
        mark("bwd: rest")
        opt.step()
        mark("optimizer")
        torch.cuda.synchronize()
        if it >= 4:
            for (n0, t0, e0), (n1, t1, e1) in zip(stamps[:-1], stamps[1:]):
                acc_h[n1] += (t1 - t0) * 1e3 / N
                acc_g[n1] += e0.elapsed_time(e1) / N
print("mode", mode)
print("%-26s %9s %9s" % ("section", "host ms", "gpu ms"))
for n in names:
    print("%-26s %9.2f %9.2f" % (n, acc_h[n], acc_g[n]))
print("%-26s %9.2f %9.2f" % ("sum", sum(acc_h.values()), sum(acc_g.values())))
