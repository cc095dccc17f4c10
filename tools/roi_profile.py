"""Dev helper: op-level profile (host + device) of the RoI heads' forward only."""
import sys
sys.path.insert(0, ".")
import torch
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
which = sys.argv[1] if len(sys.argv) > 1 else "roi"
def run(prof=False):
    opt.zero_grad()
    images = model.preprocess_image(batch); gt = [x["instances"] for x in batch]
    feats = model._features(images)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        if prof and which == "prop":
            torch.cuda.synchronize(); p = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]); p.__enter__()
        props, pl = model.proposal_generator(images, feats, gt)
        if prof and which == "prop":
            torch.cuda.synchronize(); p.__exit__(None, None, None)
        if prof and which == "roi":
            torch.cuda.synchronize(); p = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]); p.__enter__()
        props, dl = model.roi_heads(images, feats, props, gt)
        if prof and which == "roi":
            torch.cuda.synchronize(); p.__exit__(None, None, None)
    total = sum(pl.values()) + sum(dl.values())
    total.backward(); opt.step()
    return p if prof else None
with EventStorage(0):
    for _ in range(3): run()
    prof = run(True)
ka = prof.key_averages()
dev = sorted([e for e in ka if e.self_device_time_total > 0], key=lambda e: -e.self_device_time_total)
print("%s forward: device %.2f ms in %d launches; host self cpu %.2f ms" % (which, sum(e.self_device_time_total for e in dev if not e.key.startswith("aten::") and not e.key.startswith("_")) / 1e3, sum(e.count for e in dev if not e.key.startswith("aten::") and not e.key.startswith("_")), sum(e.self_cpu_time_total for e in ka) / 1e3))
ops = sorted([e for e in ka if e.key.startswith("aten::") or e.key.startswith("_")], key=lambda e: -(e.self_device_time_total + e.self_cpu_time_total))
for e in ops[:45]:
    print("%-34s n=%4d  dev %7.3f ms  cpu %7.3f ms" % (e.key[:34], e.count, e.self_device_time_total / 1e3, e.self_cpu_time_total / 1e3))
