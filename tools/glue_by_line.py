"""Dev helper: every torch (aten) operator that touches GPU tensors during one training step, attributed to the innermost
divergen_amd source line that issued it (forward; TorchDispatchMode + the Python stack) or, for the backward pass, to the
autograd node that ran it.  These are the launches that are NOT libdgx kernels: the work-list for fusing glue into kernels.

    python tools/glue_by_line.py [--graphs] [--top 80] [--size 1024]
"""
import argparse
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", action="store_true", help="keep hipGraph segments on (their ops then run only at capture time)")
ap.add_argument("--top", type=int, default=80)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--swin", default="L-22k-384")
a = ap.parse_args()

from divergen_amd.utils import graphs  # noqa: E402
if not a.graphs:
    graphs.ENABLED = False
from divergen_amd.config import get_cfg  # noqa: E402
from divergen_amd.data import synthetic_batch  # noqa: E402
from divergen_amd.modeling import build_model  # noqa: E402
from divergen_amd.solver import build_optimizer  # noqa: E402
from divergen_amd.utils.events import EventStorage  # noqa: E402

cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "DiverGen_swinL.yaml"))
cfg.merge_from_list(["MODEL.SWIN.SIZE", a.swin, "MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH",
                     os.path.join(ROOT, "configs", "metadata", "ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json")])
torch.manual_seed(42)
model = build_model(cfg).train()
model.early_proposal_backward = True
from divergen_amd.engine import total_loss  # noqa: E402
opt = build_optimizer(cfg, model)
batch = synthetic_batch(2, a.size, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")

NO_KERNEL = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::permute", "aten::transpose", "aten::t", "aten::slice", "aten::select",
             "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::as_strided", "aten::split",
             "aten::unbind", "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::new_empty", "aten::_local_scalar_dense",
             "aten::split_with_sizes", "aten::is_", "aten::size", "aten::stride", "aten::record_stream", "aten::lift_fresh", "aten::narrow",
             "aten::unfold", "aten::view_as", "aten::_reshape_alias", "aten::chunk", "aten::movedim", "aten::flatten")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.defaultdict(collections.Counter)
        self.phase = "fwd"

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name().split(".")[0]
        out = func(*args, **(kwargs or {}))
        if name.startswith(NO_KERNEL):
            return out
        flat = [x for x in torch.utils._pytree.tree_leaves((args, kwargs, out)) if isinstance(x, torch.Tensor)]
        if not any(x.is_cuda for x in flat):
            return out
        site = None
        for fr in reversed(traceback.extract_stack()):
            if "/divergen_amd/" in fr.filename and "tools/" not in fr.filename:
                site = "%s:%d %s" % (fr.filename.split("/divergen_amd/")[-1], fr.lineno, fr.name)
                break
        if site is None:      # autograd's own arithmetic: name it by operand shapes
            shp = ["x".join(map(str, x.shape)) + ":" + str(x.dtype).replace("torch.", "") for x in flat[:3]]
            site = "[%s, no divergen_amd frame] %s" % (self.phase, " ".join(shp))
        self.rows[(self.phase, site)][name.replace("aten::", "")] += 1
        return out


with EventStorage(0):
    for _ in range(2):
        opt.zero_grad(); l = model(batch); total_loss(l).backward(); opt.step()
    torch.cuda.synchronize()
    cnt = Count()
    with cnt:
        opt.zero_grad()
        l = model(batch)
        tot = total_loss(l)
        cnt.phase = "bwd"
        tot.backward()
        cnt.phase = "opt"
        opt.step()
    torch.cuda.synchronize()

tot = sum(sum(c.values()) for c in cnt.rows.values())
print("aten ops on GPU tensors in one step (views excluded): %d   [custom autograd Functions' internals count at their own lines]" % tot)
for ph in ("fwd", "bwd", "opt"):
    sub = {k: v for k, v in cnt.rows.items() if k[0] == ph}
    print("---- %s: %d ops" % (ph, sum(sum(c.values()) for c in sub.values())))
    for (p_, s), c in sorted(sub.items(), key=lambda kv: -sum(kv[1].values()))[:a.top]:
        print("%4d  %-78s %s" % (sum(c.values()), s[:78], ", ".join("%s x%d" % (o, n) for o, n in c.most_common(6))))
