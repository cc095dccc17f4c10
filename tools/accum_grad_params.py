"""Dev helper: which parameters still receive their gradient THROUGH autograd (a defined tensor handed to an AccumulateGrad node)
instead of being written in place into the gradient arena by the producing kernel.  One training step with a tensor hook on every
parameter, after the hipGraph segments have been captured.

    python tools/accum_grad_params.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from divergen_amd.config import get_cfg  # noqa: E402
from divergen_amd.data import synthetic_batch  # noqa: E402
from divergen_amd.modeling import build_model  # noqa: E402
from divergen_amd.solver import build_optimizer  # noqa: E402
from divergen_amd.utils.events import EventStorage  # noqa: E402

cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs/DiverGen_swinL.yaml"))
cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", os.path.join(ROOT, "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json")])
torch.manual_seed(42)
model = build_model(cfg).train()
opt = build_optimizer(cfg, model)
batch = synthetic_batch(2, 1024, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")


def step():
    opt.zero_grad()
    losses = model(batch)
    sum(losses.values()).backward()
    opt.step()


fired = []
it = [0]
for n, p in model.named_parameters():
    if p.requires_grad:
        p.register_hook(lambda g, n=n: fired.append((it[0], n, tuple(g.shape), "stream %d" % torch.cuda.current_stream().stream_id)) if g is not None else None)
with EventStorage(0):
    for i in range(4):
        it[0] = i
        step()
    torch.cuda.synchronize()
print("autograd-delivered parameter gradients per step (of %d parameters):" % sum(1 for p in model.parameters() if p.requires_grad),
      [sum(1 for f in fired if f[0] == i) for i in range(4)])
for f in fired[:40]:
    print("  ", f)
