"""Dev helper: where the HOST spends its time issuing a training step (cProfile over N steps, no device syncs inside):
top functions by own time and by cumulative time.   python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from divergen_amd.config import get_cfg  # noqa: E402
from divergen_amd.data import synthetic_batch  # noqa: E402
from divergen_amd.engine import total_loss  # noqa: E402
from divergen_amd.modeling import build_model  # noqa: E402
from divergen_amd.solver import build_optimizer  # noqa: E402
from divergen_amd.utils.events import EventStorage  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""       # "roi": profile only inside roi_heads.forward (the host-bound part of the step)
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs/DiverGen_swinL.yaml"))
cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", os.path.join(ROOT, "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json")])
torch.manual_seed(42)
model = build_model(cfg).train()
model.early_proposal_backward = True
opt = build_optimizer(cfg, model)
SIZE = int(os.environ.get("SIZE", "1024"))
batch = synthetic_batch(2, SIZE, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")


def step():
    opt.zero_grad()
    losses = model(batch)
    total_loss(losses).backward()
    opt.step()


with EventStorage(0):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    if ONLY == "roi":
        inner = model.roi_heads.forward

        def wrapped(*a, **k):
            pr.enable()
            try:
                return inner(*a, **k)
            finally:
                pr.disable()
        model.roi_heads.forward = wrapped
    else:
        pr.enable()
    for _ in range(N):
        step()
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.strip_dirs()
print("==== by own time (ms per step)")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:45]
for (f, line, name), (cc, nc, tt, ct, _) in rows:
    print("%8.3f ms own %8.3f ms cum %7.1f calls/step  %s:%d %s" % (tt * 1e3 / N, ct * 1e3 / N, nc / N, f, line, name))
print("==== by cumulative time, divergen_amd functions only")
rows = sorted(((k, v) for k, v in st.stats.items() if k[0].endswith(".py") and "torch" not in k[0]), key=lambda kv: -kv[1][3])[:60]
for (f, line, name), (cc, nc, tt, ct, _) in rows:
    print("%8.3f ms cum %8.3f ms own %7.1f calls/step  %s:%d %s" % (ct * 1e3 / N, tt * 1e3 / N, nc / N, f, line, name))
