"""Evaluation post-processing timings: dgx_paste_masks / dgx_paste_rle vs the reference's GPU formulation
(F.grid_sample over the whole image, D2/layers/mask_ops.py) and the CPU oracle, 300 detections at 800x1333."""
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from divergen_amd.layers import mask_ops as MO  # noqa: E402


def detections(N, H, W, S, seed):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(N, 1, 6, 6, generator=g)
    masks = F.interpolate(low, size=(S, S), mode="bicubic", align_corners=False)[:, 0].clamp(0, 1)
    cx, cy = torch.rand(N, generator=g) * W, torch.rand(N, generator=g) * H
    bw, bh = torch.rand(N, generator=g) ** 2 * W * 0.8 + 1, torch.rand(N, generator=g) ** 2 * H * 0.8 + 1
    boxes = torch.stack([(cx - bw / 2).clamp(0, W), (cy - bh / 2).clamp(0, H), (cx + bw / 2).clamp(0, W),
                         (cy + bh / 2).clamp(0, H)], 1)
    return masks, boxes


def torch_paste(masks, boxes, H, W, thr=0.5):
    N = masks.shape[0]
    x0, y0, x1, y1 = boxes.split(1, dim=1)
    ys = torch.arange(0, H, device=masks.device, dtype=torch.float32) + 0.5
    xs = torch.arange(0, W, device=masks.device, dtype=torch.float32) + 0.5
    ys = (ys[None] - y0) / (y1 - y0) * 2 - 1
    xs = (xs[None] - x0) / (x1 - x0) * 2 - 1
    grid = torch.stack([xs[:, None, :].expand(N, H, W), ys[:, :, None].expand(N, H, W)], dim=3)
    return F.grid_sample(masks[:, None], grid, align_corners=False)[:, 0] >= thr


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    N, H, W, S = 300, 800, 1333, 28
    masks, boxes = detections(N, H, W, S, 5)
    m, b = masks.cuda(), boxes.cuda()
    t_paste = timeit(lambda: MO.paste_masks_in_image(m, b, (H, W)))
    t_rle = timeit(lambda: MO.paste_masks_rle_counts(m, b, (H, W)))
    t_torch = timeit(lambda: torch_paste(m, b, H, W), n=5)
    for _ in range(3):
        rles = MO.paste_masks_rle(m, b, (H, W))
    t0 = time.perf_counter()
    for _ in range(5):
        rles = MO.paste_masks_rle(m, b, (H, W))
    t_full = (time.perf_counter() - t0) * 1e3 / 5
    bits = MO.paste_masks_in_image(m, b, (H, W))
    assert torch.equal(bits, torch_paste(m, b, H, W)), "kernel vs torch grid_sample"
    t0 = time.perf_counter()
    host = bits.cpu().numpy()
    t_d2h = (time.perf_counter() - t0) * 1e3
    def rle_counts(mask):          # column-major run lengths on the host (numpy), what mask_util.encode computes first
        flat = np.asarray(mask, np.uint8).T.reshape(-1)
        change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
        return np.diff(np.concatenate([[0], change, [flat.size]]))
    t0 = time.perf_counter()
    for n in range(20):
        rle_counts(host[n])
    t_cpu_rle = (time.perf_counter() - t0) * 1e3 / 20 * N
    out_bytes = N * H * W
    print("paste_masks  (u8 N,H,W)        : %8.3f ms  (%.0f GB/s written)" % (t_paste, out_bytes / t_paste / 1e6))
    print("paste_rle    (counts only)     : %8.3f ms  incl. the nruns readback" % t_rle)
    print("torch grid_sample formulation  : %8.3f ms" % t_torch)
    print("paste_masks_rle end to end     : %8.3f ms  (kernel + D2H of counts + %d host strings)" % (t_full, len(rles)))
    print("bitmask D2H (what mask_util.encode needs first): %8.3f ms;  numpy RLE of %d masks on the host: %8.1f ms"
          % (t_d2h, N, t_cpu_rle))


if __name__ == "__main__":
    main()
