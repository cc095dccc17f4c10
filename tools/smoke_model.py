"""Dev helper: build the model from a reference YAML and run fwd+bwd on synthetic data (GPU)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from divergen_amd.config import get_cfg
from divergen_amd.data import synthetic_batch
from divergen_amd.modeling import build_model
from divergen_amd.utils.events import EventStorage

size = sys.argv[1] if len(sys.argv) > 1 else "T"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = get_cfg()
cfg.merge_from_file("configs/DiverGen_swinL.yaml")
cfg.merge_from_list(["MODEL.SWIN.SIZE", size, "MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH",
                     "configs/metadata/ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json"])
torch.manual_seed(42)
model = build_model(cfg).train()
batch = synthetic_batch(2, res, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
with EventStorage(0) as st:
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        losses = model(batch)
        total = sum(losses.values())
        total.backward()
        torch.cuda.synchronize()
        print(it, "%.3fs" % (time.time() - t), {k: round(float(v), 4) for k, v in losses.items()})
        model.zero_grad(set_to_none=True)
    print({k: round(float(v[0]), 3) for k, v in st.latest().items()})
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
