"""Where does the product's gradient leave the oracle's?  The end-to-end gradient comparison of tests/test_gpu_model.py, run once per
loss group (CenterNet losses / box losses / mask loss): the group whose backward path is off shows it on the backbone + FPN segments.
    python tools/grad_parity_probe.py [T|L-22k-384] [size]"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_gpu_model import run_e2e_vs_oracle  # noqa: E402

swin = sys.argv[1] if len(sys.argv) > 1 else "T"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
GROUPS = {"centernet": lambda k: "centernet" in k, "box": lambda k: "stage" in k, "mask": lambda k: k == "loss_mask", "all": None}
for name, f in GROUPS.items():
    print("==== losses:", name, flush=True)
    mp = pytest.MonkeyPatch()
    try:
        rep = run_e2e_vs_oracle(mp, swin, size, grads=True, loss_filter=f, grad_bounds=None)
        for row in rep["_grad_rows"]:
            if row[2] > 0 or row[3] > 0:
                print("    %-70s %-18s |g| %.3e rel %.3e (fp32 oracle %.3e)" % row)
    finally:
        mp.undo()
