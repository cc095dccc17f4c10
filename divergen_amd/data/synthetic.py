"""Synthetic batches with the reference's batched-input contract (D2/modeling/meta_arch/rcnn.py:122-144):
list[dict] with `image` uint8 CHW, `instances` (gt_boxes, gt_classes, gt_masks, instance_source),
height/width/file_name.  Shapes follow SURVEY.md 8(d): n_gt = 12 boxes with w,h ~ U(32,400) px,
classes ~ U{0..num_classes-1}, masks = ellipses inscribed in the boxes, instance_source = [0]*8+[1]*4."""
import numpy as np
import torch

from ..structures import BitMasks, Boxes, Instances


def synthetic_batch(batch_size, size, num_classes, seed=1234, n_gt=12, device=None):
    rng = np.random.default_rng(seed)
    H = W = size
    out = []
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    for b in range(batch_size):
        img = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
        lim = min(400, size * 0.6)
        w = rng.uniform(min(32, lim / 2), lim, n_gt)
        h = rng.uniform(min(32, lim / 2), lim, n_gt)
        x0 = rng.uniform(0, W - w)
        y0 = rng.uniform(0, H - h)
        boxes = np.stack([x0, y0, x0 + w, y0 + h], 1).astype(np.float32)
        cx, cy = (boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2
        masks = (((xx[None] - cx[:, None, None]) / (w[:, None, None] / 2)) ** 2 +
                 ((yy[None] - cy[:, None, None]) / (h[:, None, None] / 2)) ** 2) <= 1.0
        inst = Instances((H, W))
        inst.gt_boxes = Boxes(torch.from_numpy(boxes))
        inst.gt_classes = torch.from_numpy(rng.integers(0, num_classes, n_gt).astype(np.int64))
        inst.gt_masks = BitMasks(torch.from_numpy(masks))
        src = np.zeros(n_gt, np.int64)
        src[n_gt - n_gt // 3:] = 1
        inst.instance_source = torch.from_numpy(src)
        d = {"image": torch.from_numpy(img), "instances": inst, "height": H, "width": W,
             "file_name": "synthetic_%d_%d.jpg" % (seed, b)}
        if device is not None:
            d["image"] = d["image"].to(device)
            d["instances"] = inst.to(device)
        out.append(d)
    return out
