"""Dev helper: kernel-level profile of the FPN top-down + CenterNet tower (eager, fwd+bwd) on bench shapes."""
import sys
sys.path.insert(0, ".")
import torch
from torch.profiler import ProfilerActivity, profile
from divergen_amd.config import get_cfg
from divergen_amd.modeling import build_model
from divergen_amd.solver import build_optimizer
from divergen_amd.utils import graphs
graphs.ENABLED = False
cfg = get_cfg(); cfg.merge_from_file("configs/DiverGen_swinL.yaml")
cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json"])
torch.manual_seed(42)
model = build_model(cfg).train(); opt = build_optimizer(cfg, model)
fpn = model.backbone
head = model.proposal_generator.centernet_head
feats = {"swin1": torch.randn(2, 384, 128, 128, device="cuda").bfloat16().to(memory_format=torch.channels_last).requires_grad_(True),
         "swin2": torch.randn(2, 768, 64, 64, device="cuda").bfloat16().to(memory_format=torch.channels_last).requires_grad_(True),
         "swin3": torch.randn(2, 1536, 32, 32, device="cuda").bfloat16().to(memory_format=torch.channels_last).requires_grad_(True)}
def run():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = fpn._top_down(feats)
        _, reg, hm = head([out[k] for k in ("p3", "p4", "p5", "p6", "p7")])
    loss = sum(r.float().square().mean() for r in reg) + sum(h.float().square().mean() for h in hm)
    loss.backward()
for _ in range(3): run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.self_device_time_total > 0 and not e.key.startswith("aten::") and not e.key[0] == "_" and "Backward" not in e.key]
rows.sort(key=lambda e: -e.self_device_time_total)
print("FPN top-down + CenterNet tower fwd+bwd: %.2f ms in %d kernels" % (sum(e.self_device_time_total for e in rows) / 1e3, sum(e.count for e in rows)))
for e in rows[:32]:
    print("%7.3f ms n=%4d avg %6.1f us  %s" % (e.self_device_time_total / 1e3, e.count, e.self_device_time_total / e.count, e.key[:120]))
