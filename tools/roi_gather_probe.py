"""Dev helper: the RoIAlign gather backward (dgx_roi_pooler_bwd_gather) alone at the benchmark's level geometry (128^2 / 64^2 / 32^2 x 2
images, C = 256) for different RoI populations: what the launch costs with no RoI, with small boxes only, with the large boxes an untrained
model proposes.  HIP events around 20 launches each."""
import sys
import torch
sys.path.insert(0, ".")
from divergen_amd import layers as la

dev = "cuda"
g = torch.Generator().manual_seed(1)
C, scales = 256, (1 / 8, 1 / 16, 1 / 32)
feats = [torch.zeros(2, C, s, s, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True) for s in (128, 64, 32)]


def boxes(n, lo, hi):
    ctr = torch.rand(n, 2, generator=g) * 1024
    wh = torch.rand(n, 2, generator=g) * (hi - lo) + lo
    r = torch.cat([(torch.arange(n) % 2).float()[:, None], (ctr - wh / 2).clamp(0, 1023), (ctr + wh / 2).clamp(1, 1024)], 1)
    return r[torch.argsort(r[:, 0], stable=True)].to(dev)


def time(rois, S, tag):
    import ctypes
    from divergen_amd import _lib as L
    R = rois.shape[0]
    go = torch.randn(R, S, S, C, device=dev).to(torch.bfloat16)
    grads = [torch.empty(2, s, s, C, device=dev, dtype=torch.bfloat16) for s in (128, 64, 32)]
    ptrs = (ctypes.c_void_p * 3)(*[x.data_ptr() for x in grads])
    Hs = (ctypes.c_int * 3)(128, 64, 32)
    rois = rois.float().contiguous()

    def launch():
        L.check(L.lib().dgx_roi_pooler_bwd_gather_accum(L.ptr(go), ptrs, Hs, Hs, 3, 3, 0.125, 1, L.ptr(rois), 2, C, R, S, S, 0, 0, 1.0,
                                                        L.dtype_code(go), L.stream()), "gather")
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record()
    torch.cuda.synchronize()
    print("%-52s R %5d S %2d: %7.1f us per launch" % (tag, R, S, e0.elapsed_time(e1) * 1e3 / 20))


time(boxes(1024, 8, 16), 7, "tiny boxes (one or two P3 pixels each)")
time(boxes(1024, 40, 120), 7, "small boxes (P3 / P4)")
time(boxes(1024, 300, 1000), 7, "large boxes (untrained model: mostly P5)")
time(boxes(256, 300, 1000), 7, "large boxes, a quarter of them")
time(boxes(64, 300, 1000), 7, "large boxes, 64")
time(boxes(1, 8, 16), 7, "one tiny box")
nb = boxes(1024, 40, 120)
nb[:, 0] = 7.0
time(nb, 7, "1024 boxes of an image that is not there (scan only)")
for n in (128, 256, 512):
    time(boxes(n, 8, 16), 7, "tiny boxes")
time(boxes(256, 40, 120), 14, "mask stage: small boxes, 14 x 14")
time(boxes(256, 300, 1000), 14, "mask stage: large boxes, 14 x 14")
