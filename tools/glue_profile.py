"""Dev helper: which small torch ops (copies, casts, adds, fills, sums) cost GPU time, by op and input shape."""
import sys
import torch
sys.path.insert(0, ".")
from torch.profiler import ProfilerActivity, profile
from divergen_amd.config import get_cfg
from divergen_amd.data import synthetic_batch
from divergen_amd.modeling import build_model
from divergen_amd.solver import build_optimizer
from divergen_amd.utils.events import EventStorage
from divergen_amd.utils import graphs
graphs.ENABLED = False      # graphs hide the ops
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
skip = ("aten::mm", "aten::addmm", "aten::bmm")
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0 and e.key.startswith("aten::") and e.key not in skip]
rows.sort(key=lambda e: -e.self_device_time_total)
print("glue ops device time %.2f ms" % (sum(e.self_device_time_total for e in rows) / 1e3))
for e in rows[:60]:
    print("%7.3f ms n=%3d avg %6.1f us  %-28s %s" % (e.self_device_time_total / 1e3, e.count, e.self_device_time_total / e.count, e.key, str(e.input_shapes)[:110]))
