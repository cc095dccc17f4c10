"""Dev helper: kernel launches and device time per forward region of the heads (torch profiler ranges; graphs off so the
launches are visible).  Backward kernels run on the autograd thread and are reported as one remainder line."""
import sys
import collections
import torch
sys.path.insert(0, ".")
from torch.profiler import ProfilerActivity, profile, record_function
from divergen_amd.utils import graphs
graphs.ENABLED = False
from divergen_amd.config import get_cfg
from divergen_amd.data import synthetic_batch
from divergen_amd.modeling import build_model
from divergen_amd.solver import build_optimizer
from divergen_amd.utils.events import EventStorage
from divergen_amd.modeling.dense_heads import centernet as CN
from divergen_amd.modeling.roi_heads import detic_roi_heads as RH, detic_fast_rcnn as FR, mask_head as MH, poolers as PL
from divergen_amd.modeling.meta_arch import custom_rcnn as CR
from divergen_amd.modeling.backbone import fpn as FP


def wrap(owner, name, label):
    orig = getattr(owner, name)

    def f(*a, **k):
        with record_function("R:" + label):
            return orig(*a, **k)
    setattr(owner, name, f)


wrap(CR.CustomRCNN, "preprocess_image", "preprocess_image")
wrap(CR.CustomRCNN, "_features", "backbone+fpn")
wrap(CN.CenterNet, "_run_head", "centernet head")
wrap(CN.CenterNet, "compute_grids", "compute_grids")
wrap(CN.CenterNet, "_get_ground_truth", "centernet targets+label inds")
wrap(CN.CenterNet, "losses", "centernet losses")
wrap(CN.CenterNet, "predict_instances", "predict_instances")
wrap(RH.DeticCascadeROIHeads, "label_and_sample_proposals", "label_and_sample")
wrap(RH.DeticCascadeROIHeads, "_forward_box_train", "box cascade (3 stages)")
wrap(RH.DeticCascadeROIHeads, "_forward_mask", "mask branch")
wrap(RH, "select_foreground_proposals", "select_foreground")
wrap(MH, "mask_rcnn_loss", "mask loss")
wrap(PL.ROIPooler, "forward", "roi pooler")
wrap(FR.DeticFastRCNNOutputLayers, "losses_from_tensors", "box losses")

cfg = get_cfg()
cfg.merge_from_file("configs/DiverGen_swinL.yaml")
cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json"])
torch.manual_seed(42)
model = build_model(cfg).train()
opt = build_optimizer(cfg, model)
batch = synthetic_batch(2, 1024, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")


def step():
    opt.zero_grad()
    with record_function("R:forward total"):
        l = model(batch)
    with record_function("R:backward total"):
        sum(l.values()).backward()
    with record_function("R:optimizer"):
        opt.step()


with EventStorage(0):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
evs = prof.events()
ranges = [e for e in evs if e.name.startswith("R:")]
ops = [e for e in evs if e.kernels and not e.name.startswith("R:")]
rows = collections.OrderedDict()
for r in sorted(ranges, key=lambda e: e.time_range.start):
    n = t = 0
    for o in ops:
        if o.thread == r.thread and r.time_range.start <= o.time_range.start <= r.time_range.end:
            n += len(o.kernels)
            t += sum(k.duration for k in o.kernels)
    a = rows.setdefault(r.name[2:], [0, 0.0, 0.0, 0])
    a[0] += n
    a[1] += t / 1e3
    a[2] += (r.time_range.end - r.time_range.start) / 1e3
    a[3] += 1
if len(sys.argv) > 1 and sys.argv[1] == "backward":      # everything that is not inside the forward / optimizer ranges
    fw = [r for r in ranges if r.name[2:] in ("forward total", "optimizer")]
    hist, tim = collections.Counter(), collections.Counter()
    for o in ops:
        if any(o.thread == r.thread and r.time_range.start <= o.time_range.start <= r.time_range.end for r in fw):
            continue
        nm = o.name if len(sys.argv) < 3 or sys.argv[2] != o.name else o.name + " " + str(o.input_shapes)[:90]
        hist[nm] += len(o.kernels)
        tim[nm] += sum(k.duration for k in o.kernels)
    for k, v in hist.most_common(45):
        print("   %-46s %5d launches %9.3f ms" % (k, v, tim[k] / 1e3))
elif len(sys.argv) > 1:          # op histogram inside one region, e.g.  python tools/launch_regions.py "backbone+fpn"
    want = [r for r in ranges if r.name[2:] == sys.argv[1]][:1]
    hist = collections.Counter()
    tim = collections.Counter()
    for r in want:
        for o in ops:
            if o.thread == r.thread and r.time_range.start <= o.time_range.start <= r.time_range.end:
                nm = o.name if len(sys.argv) < 3 or sys.argv[2] != o.name else o.name + " " + str(o.input_shapes)[:90]
                hist[nm] += len(o.kernels)
                tim[nm] += sum(k.duration for k in o.kernels)
    for k, v in hist.most_common(40):
        print("   %-40s %5d launches %9.3f ms" % (k, v, tim[k] / 1e3))
allk = sum(len(o.kernels) for o in ops)
allt = sum(sum(k.duration for k in o.kernels) for o in ops) / 1e3
print("%-34s %8s %10s %9s %6s" % ("region (nested regions overlap)", "launches", "device ms", "host ms", "calls"))
for k, (n, t, h, c) in rows.items():
    print("%-34s %8d %10.3f %9.3f %6d" % (k, n, t, h, c))
fw = rows.get("forward total", [0, 0, 0, 0])
print("%-34s %8d %10.3f" % ("all ops with kernels", allk, allt))
print("%-34s %8d %10.3f" % ("not in forward/optimizer (backward)", allk - fw[0] - rows.get("optimizer", [0])[0], allt - fw[1] - rows.get("optimizer", [0, 0])[1]))
