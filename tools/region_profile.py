"""Dev helper: per-region host time, GPU time and launch count inside the heads' forward (regions = wrapped methods;
a device sync brackets each region, so the numbers are the region's own cost, not the overlapped schedule)."""
import sys
import time
import collections
import torch
sys.path.insert(0, ".")
from divergen_amd.config import get_cfg
from divergen_amd.data import synthetic_batch
from divergen_amd.modeling import build_model
from divergen_amd.solver import build_optimizer
from divergen_amd.utils.events import EventStorage
from divergen_amd.modeling.dense_heads import centernet as CN, centernet_head as CH
from divergen_amd.modeling.roi_heads import detic_roi_heads as RH, detic_fast_rcnn as FR, mask_head as MH, poolers as PL, box_head as BH

stats = collections.OrderedDict()
depth = [0]


def wrap(cls, name, label=None):
    orig = getattr(cls, name)
    label = label or "%s.%s" % (cls.__name__ if hasattr(cls, "__name__") else cls, name)

    def f(*a, **k):
        top = depth[0] == 0
        depth[0] += 1
        if top:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        r = orig(*a, **k)
        depth[0] -= 1
        if top:
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            s = stats.setdefault(label, [0.0, 0.0, 0])
            s[0] += (t1 - t0) * 1e3
            s[1] += (t2 - t0) * 1e3
            s[2] += 1
        return r
    setattr(cls, name, f)


wrap(CH.CenterNetHead, "forward")
wrap(CN.CenterNet, "compute_grids")
wrap(CN.CenterNet, "_get_ground_truth")
wrap(CN.CenterNet, "losses")
wrap(CN.CenterNet, "predict_instances")
wrap(RH.DeticCascadeROIHeads, "label_and_sample_proposals")
wrap(RH.DeticCascadeROIHeads, "_match_and_label_boxes")
wrap(RH.DeticCascadeROIHeads, "_run_stage")
wrap(FR.DeticFastRCNNOutputLayers, "losses")
wrap(FR.DeticFastRCNNOutputLayers, "predict_boxes")
wrap(RH.DeticCascadeROIHeads, "_create_proposals_from_boxes")
wrap(RH.DeticCascadeROIHeads, "_forward_mask")

cfg = get_cfg()
cfg.merge_from_file("configs/DiverGen_swinL.yaml")
cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json"])
torch.manual_seed(42)
model = build_model(cfg).train()
opt = build_optimizer(cfg, model)
batch = synthetic_batch(2, 1024, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
N = 5
with EventStorage(0):
    for it in range(N + 2):
        if it == 2:
            stats.clear()
        opt.zero_grad()
        sum(model(batch).values()).backward()
        opt.step()
print("%-52s %8s %10s %6s" % ("region (forward only)", "host ms", "host+gpu ms", "calls"))
for k, (h, g, n) in stats.items():
    print("%-52s %8.2f %10.2f %6d" % (k, h / N, g / N, n // N))
print("%-52s %8.2f %10.2f" % ("sum", sum(v[0] for v in stats.values()) / N, sum(v[1] for v in stats.values()) / N))
