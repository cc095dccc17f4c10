"""Library-GEMM algorithm selection for the shapes of the Swin-L CenterNet2 step.

The forward / input-gradient GEMMs go to hipBLASLt / rocBLAS through PyTorch; its default heuristic is not the
fastest solution for ~20 of the ~90 shapes of this model.  `gemm_gfx950.csv` is the result of PyTorch's TunableOp
search on an MI355X (ROCm 7.2 image; produced with PYTORCH_TUNABLEOP_ENABLED=1 python bench.py, 80 s) and is
applied read-only (tuning off).  TunableOp validates the file's header (PyTorch / hipBLASLt / rocBLAS versions,
gfx arch) and ignores it on a mismatch, so a different stack silently falls back to the default heuristic.
Must be called before the first GEMM (bench.py / train_net.py call it before importing torch).
DGX_TUNED_GEMM=0 disables it."""
import os
import shutil
import tempfile


def enable():
    if os.environ.get("DGX_TUNED_GEMM", "1") != "1" or "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        return False
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_gfx950.csv")
    if not os.path.isfile(src):
        return False
    # TunableOp reads <FILENAME stem><device ordinal>.csv: a private copy under EVERY ordinal this process could end up
    # using (LOCAL_RANK with torchrun, 0 when the launcher narrows the visible devices instead)
    d = tempfile.mkdtemp(prefix="dgx_tunableop_")
    for ordinal in range(16):
        shutil.copy(src, os.path.join(d, "results%d.csv" % ordinal))
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
    os.environ["PYTORCH_TUNABLEOP_RECORD_UNTUNED"] = "0"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(d, "results.csv")
    return True
