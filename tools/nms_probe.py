"""Dev helper: what the batched NMS of the bench workload sees -- candidates above threshold, kept boxes, and the duration of the mask /
sweep launches (HIP events around dgx_nms_batched) on the bench's model and synthetic batches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from divergen_amd.config import get_cfg  # noqa: E402
from divergen_amd.data import synthetic_batch  # noqa: E402
from divergen_amd.layers import box_ops  # noqa: E402
from divergen_amd.modeling import build_model  # noqa: E402
from divergen_amd.modeling.dense_heads import centernet as CN  # noqa: E402
from divergen_amd.solver import build_optimizer  # noqa: E402
from divergen_amd.utils.events import EventStorage  # noqa: E402

cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs/DiverGen_swinL.yaml"))
cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", os.path.join(ROOT, "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json")])
torch.manual_seed(42)
model = build_model(cfg).train()
opt = build_optimizer(cfg, model)
log = []
orig = box_ops.nms_batched_sorted


def probed(boxes, scores, n_valid, thr, max_keep=0, cap=None):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = orig(boxes, scores, n_valid, thr, max_keep=max_keep, cap=cap)
    b.record()
    log.append((a, b, n_valid, out[1], boxes.shape, thr, max_keep))
    return out


CN.nms_batched_sorted = probed
with EventStorage(0):
    for it in range(12):
        batch = synthetic_batch(2, 1024, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda", seed=it)
        opt.zero_grad()
        sum(model(batch).values()).backward()
        opt.step()
torch.cuda.synchronize()
for a, b, nv, nk, shp, thr, mk in log:
    print("K %s thr %.2f max_keep %d: n_valid %s kept %s  mask+sweep %.1f us" % (tuple(shp), thr, mk, nv.tolist(), nk.tolist(), a.elapsed_time(b) * 1e3))
