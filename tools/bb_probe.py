import os, sys, time, torch
sys.path.insert(0, ".")
from divergen_amd.config import get_cfg
from divergen_amd.modeling import build_model
from divergen_amd.modeling.meta_arch.custom_rcnn import _BackboneForGraph
cfg = get_cfg(); cfg.merge_from_file("configs/DiverGen_swinL.yaml")
cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json"])
model = build_model(cfg).train()
x = torch.randn(2, 3, 1024, 1024, device="cuda").to(memory_format=torch.channels_last)
names = list(model.backbone.output_shape().keys())
mod = _BackboneForGraph(model.backbone, names, True)
def run(fn, n=5):
    for _ in range(2):
        outs = fn(x); sum(o.float().mean() for o in outs).backward()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        outs = fn(x); torch.cuda.synchronize(); t1 = time.perf_counter()
        sum(o.float().mean() for o in outs).backward(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
print("eager backbone fwd+bwd ms:", run(mod))
g = torch.cuda.make_graphed_callables(mod, (x.clone(),))
print("graphed backbone fwd+bwd ms:", run(g))
