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
from divergen_amd.modeling.meta_arch.custom_rcnn import _JoinGradients  # noqa: E402
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
names = ["zero_grad+preproc", "backbone+fpn fwd", "proposal gen fwd", "early bwd (issue)", "wait for counts", "roi heads fwd (rest)",
         "loss sum", "bwd: start -> FPN grads", "bwd: rest", "optimizer"]
acc_h = dict.fromkeys(names, 0.0)
acc_g = dict.fromkeys(names, 0.0)
stamps = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    stamps.append((name, time.perf_counter(), e))


with EventStorage(0):
    for it in range(N + 4):
        stamps.clear()
        torch.cuda.synchronize()
        mark(None)
        opt.zero_grad()
        images = model.preprocess_image(batch)
        gt = [x["instances"] for x in batch]
        mark("zero_grad+preproc")
        feats = model._features(images)
        mark("backbone+fpn fwd")
        fired = []

        def hook(g):
            if not fired:
                fired.append(1)
                mark("bwd: start -> FPN grads")
            return g
        with torch.autocast("cuda", dtype=torch.bfloat16):
            keys = list(feats.keys())
            for k in keys:
                feats[k].register_hook(hook)
            if mode == "early":
                stubs = [feats[k].detach().requires_grad_(True) for k in keys]
                props, pl = model.proposal_generator(images, dict(zip(keys, stubs)), gt)
                mark("proposal gen fwd")
                pl_total = torch.stack([v.float().reshape(()) for v in pl.values()]).sum()

                def before():
                    pl_total.backward()
                    mark("early bwd (issue)")
                model.roi_heads.__dict__["_before_host_read"] = before
                orig_sync = torch.cuda.Event.synchronize

                def sync(self):
                    orig_sync(self)
                    mark("wait for counts")
                torch.cuda.Event.synchronize = sync
                f2 = dict(zip(keys, _JoinGradients.apply(stubs, *[feats[k] for k in keys])))
                props, dl = model.roi_heads(images, f2, props, gt)
                torch.cuda.Event.synchronize = orig_sync
                pl = {k: v.detach() for k, v in pl.items()}
            else:
                props, pl = model.proposal_generator(images, feats, gt)
                mark("proposal gen fwd")
                props, dl = model.roi_heads(images, feats, props, gt)
            mark("roi heads fwd (rest)")
        total = sum(pl.values()) + sum(dl.values())
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
