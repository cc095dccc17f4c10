"""Dev helper: per-shape GEMM time in one training step (torch profiler, record_shapes)."""
import os, sys, time
sys.path.insert(0, ".")
import torch
sys.path.insert(0, ".")
from torch.profiler import ProfilerActivity, profile
from divergen_amd.config import get_cfg
from divergen_amd.data import synthetic_batch
from divergen_amd.modeling import build_model
from divergen_amd.solver import build_optimizer
from divergen_amd.utils.events import EventStorage
cfg = get_cfg(); cfg.merge_from_file("configs/DiverGen_swinL.yaml")
cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json"])
torch.manual_seed(42)
model = build_model(cfg).train(); opt = build_optimizer(cfg, model)
batch = synthetic_batch(2, 1024, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
def step():
    opt.zero_grad(); l = model(batch); sum(l.values()).backward(); opt.step()
with EventStorage(0):
    for _ in range(3): step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::linear")]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print("total GEMM device time %.2f ms" % (tot / 1e3))
for e in rows[:40]:
    sh = [s for s in e.input_shapes if len(s) == 2]
    fl = 0
    if e.key == "aten::mm" and len(sh) >= 2: fl = 2.0 * sh[0][0] * sh[0][1] * sh[1][1]
    if e.key == "aten::addmm" and len(sh) >= 2: fl = 2.0 * sh[-2][0] * sh[-2][1] * sh[-1][1]
    t = e.self_device_time_total / e.count
    print("%-11s n=%3d  %8.1f us each  %7.2f ms total  %6.0f TF/s  %s" % (e.key, e.count, t, e.self_device_time_total / 1e3, fl / t / 1e6 if t else 0, e.input_shapes))
