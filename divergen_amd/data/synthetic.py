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


def write_mini_lvis(root, n_images=32, image_hw=(480, 640), n_obj=12, n_pool=64, pool_px=(256, 512), n_cats=1203, seed=0,
                    split="lvis_v1_train", poly_vertices=40):
    """A small LVIS-FORMAT dataset on disk, for running the REAL data path (json -> mapper -> instance pool -> compositor) where
    LVIS itself is absent (tests, bench.py --through-loader): `root`/coco/train2017/*.jpg, `root`/lvis/<split>.json with
    polygon annotations (ellipses as `poly_vertices`-gons, n_obj per image, categories drawn over all n_cats), and an instance
    pool `root`/pool/*.png (RGBA, soft-edged alpha, pool_px range) indexed by `root`/pool.json {0-based category: [keys]} in the
    reference's INST_POOL_PATH format.  Returns dict(root, pool_json, n_images, n_pool).  Deterministic in `seed`."""
    import json
    import os
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "coco", "train2017"), exist_ok=True)
    os.makedirs(os.path.join(root, "lvis"), exist_ok=True)
    os.makedirs(os.path.join(root, "pool"), exist_ok=True)

    def smooth(h, w, c):
        low = rng.integers(0, 256, (max(h // 32, 2), max(w // 32, 2), c), dtype=np.uint8)
        img = np.asarray(Image.fromarray(low if c > 1 else low[..., 0]).resize((w, h), Image.BICUBIC)).reshape(h, w, c).astype(np.int16)
        return np.clip(img + rng.integers(-12, 13, (h, w, c)), 0, 255).astype(np.uint8)

    H, W = image_hw
    images, anns = [], []
    for i in range(n_images):
        Image.fromarray(smooth(H, W, 3)).save(os.path.join(root, "coco", "train2017", "%012d.jpg" % (i + 1)), quality=90)
        images.append({"id": i + 1, "height": H, "width": W, "coco_url": "http://images.cocodataset.org/train2017/%012d.jpg" % (i + 1),
                       "neg_category_ids": [int(c) for c in rng.integers(1, n_cats + 1, 3)], "not_exhaustive_category_ids": []})
        for _ in range(n_obj):
            bw, bh = rng.uniform(0.05, 0.45) * W, rng.uniform(0.05, 0.45) * H
            x0, y0 = rng.uniform(0, W - bw), rng.uniform(0, H - bh)
            t = np.linspace(0, 2 * np.pi, poly_vertices, endpoint=False)
            px, py = x0 + bw / 2 + bw / 2 * np.cos(t), y0 + bh / 2 + bh / 2 * np.sin(t)
            poly = np.stack([px, py], 1).reshape(-1).round(2).tolist()
            anns.append({"id": len(anns) + 1, "image_id": i + 1, "category_id": int(rng.integers(1, n_cats + 1)),
                         "bbox": [float(x0), float(y0), float(bw), float(bh)], "segmentation": [poly], "area": float(np.pi * bw * bh / 4)})
    cats = [{"id": c, "name": "c%d" % c, "frequency": "fcr"[c % 3], "image_count": 10, "instance_count": 20} for c in range(1, n_cats + 1)]
    with open(os.path.join(root, "lvis", split + ".json"), "w") as f:
        json.dump({"images": images, "annotations": anns, "categories": cats}, f)
    pool = {}
    for k in range(n_pool):
        h, w = int(rng.integers(pool_px[0], pool_px[1] + 1)), int(rng.integers(pool_px[0], pool_px[1] + 1))
        yy, xx = np.mgrid[0:h, 0:w]
        soft = 255 * np.clip(2.0 * (1.0 - (((xx - w / 2) / (w / 2.3)) ** 2 + ((yy - h / 2) / (h / 2.3)) ** 2)), 0, 1)
        path = os.path.join(root, "pool", "inst%05d.png" % k)
        Image.fromarray(np.dstack([smooth(h, w, 3), soft.astype(np.uint8)[..., None]]), "RGBA").save(path)
        pool.setdefault(str(int(rng.integers(0, n_cats))), []).append(path)
    with open(os.path.join(root, "pool.json"), "w") as f:
        json.dump(pool, f)
    return {"root": root, "pool_json": os.path.join(root, "pool.json"), "n_images": n_images, "n_pool": n_pool}
