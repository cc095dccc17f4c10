"""Dev helper: where the step's device copies / fills come from (hipMemcpyAsync blits `__amd_rocclr_copyBuffer`, torch copy /
fill / add kernels): one profiled training step with Python stacks, device kernels grouped by the innermost divergen_amd
frame (forward) or the enclosing autograd node (backward).  hipGraph segments stay ON (their inner ops run at capture only; what
remains per step are the graph's input copies).

    python tools/copy_sources.py [pattern,pattern,...]      default: copyBuffer,fillBuffer,FillFunctor,copy_kernel,CUDAFunctor_add
"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from divergen_amd.config import get_cfg  # noqa: E402
from divergen_amd.data import synthetic_batch  # noqa: E402
from divergen_amd.modeling import build_model  # noqa: E402
from divergen_amd.solver import build_optimizer  # noqa: E402
from divergen_amd.utils.events import EventStorage  # noqa: E402

pats = (sys.argv[1] if len(sys.argv) > 1 else "copyBuffer,fillBuffer,FillFunctor,copy_kernel,CUDAFunctor_add,Memcpy,Memset").split(",")
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


with EventStorage(0):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()

evs = prof.events()
by_id = {e.id: e for e in evs}
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for e in evs:
    ks = [k for k in (e.kernels or []) if any(p in k.name for p in pats)]
    if not ks:
        continue
    where = None
    for fr in (e.stack or []):
        if "divergen_amd" in fr and "site-packages" not in fr:
            where = fr.split("divergen_amd/")[-1]
            break
    if where is None:                       # backward / engine thread: climb to the autograd node
        p = e
        while p is not None and "evaluate_function" not in p.name and "Backward" not in p.name:
            p = p.cpu_parent
        where = "[bwd] " + (p.name.split(": ")[-1] if p is not None else "?") if p is not None else "[no frame] " + e.name
    key = (where, e.name, str(e.input_shapes)[:70])
    a = agg[key]
    a[0] += len(ks)
    a[1] += sum(k.duration for k in ks)
    for k in ks:
        a[2][k.name.split("(")[0][-40:]] += 1
tot_n = sum(a[0] for a in agg.values())
tot_t = sum(a[1] for a in agg.values())
print("matching device kernels in one step: %d, %.3f ms" % (tot_n, tot_t / 1e3))
for key, (n, t, names) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print("%3d  %7.1f us  %-58s %-22s %s  {%s}" % (n, t, key[0][:58], key[1][:22], key[2], ", ".join("%s x%d" % kv for kv in names.most_common(8 if "Graphed" in key[0] or "Graphed" in key[1] else 2))))
