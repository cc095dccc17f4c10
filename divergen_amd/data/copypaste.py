"""DiverGen copy-paste data path with the GPU compositor.

Host-side mirror of InstPool (DG/divergen/data/custom_build_copypaste_mapper.py:94-566) for the shipped
configuration (INST_POOL_FORMAT 'RGBA', INST_POOL_SAMPLE_TYPE 'cas_random', CP_METHOD ['basic'],
USE_COPY_METHOD 'syn_copy'): class-balanced sampling of generated instances, size prior (Gaussian
relative-area prior for LVIS classes with statistics, U(RANDOM_SCALE_MIN, MAX) of the source size for
the rest), random placement.  Two halves:
  InstPool.prepare    numpy / PIL only, the reference's np.random call order (pinned on tests/golden/pool_draws.npz, draws of
                      the reference's own InstPool); runs in the DATALOADER.NUM_WORKERS loader processes and returns the K
                      patches packed into one flat buffer + descriptors (CPU tensors)
  InstPool.composite  training process: upload + ONE call into libdgx (dgx_copy_paste) for the pixels -- image blend, mask
                      occlusion updates, box recomputation and the occlusion filter of `_copy_paste`, all pastes at once."""
import json
import os
from collections import defaultdict

import numpy as np
import torch

from ..layers.copy_paste import PackedPastes, copy_paste
from ..structures import BitMasks, Boxes, Instances


def largest_connected_component(mask):
    """Role of get_largest_connect_component (mapper.py:25-36, cv2 contours): keep the largest blob."""
    from scipy import ndimage
    lab, n = ndimage.label(mask)
    if n <= 1:
        return mask
    sizes = ndimage.sum(mask, lab, index=np.arange(1, n + 1))
    return (lab == (1 + int(np.argmax(sizes)))).astype(mask.dtype)


class InstPool:
    def __init__(self, pool_json, train_size, area_stats_json=None, max_samples=20, random_scale=False,
                 random_scale_min=0.1, random_scale_max=2.0, random_scale_min_size=5, use_largest_part=True,
                 scale_min=10, scale_max=0.5, shape_jitter=0.2, mask_threshold=128,
                 instance_filter_min=0.01, instance_filter_max=1.0, loader=None):
        pool = json.load(open(pool_json)) if isinstance(pool_json, str) else pool_json
        self.per_cat_pool = defaultdict(list)
        self.dataset, self.data_to_cat = [], {}
        for cat, paths in pool.items():
            for p in paths:
                self.per_cat_pool[int(cat)].append(len(self.dataset))
                self.data_to_cat[p] = int(cat)          # pool json keys ARE the 0-based labels (mapper.py:143-152;
                                                        # written as `id - 1` by DG/filteration/clean_pool_if.py:173)
                self.dataset.append(p)
        self.cats = list(self.per_cat_pool.keys())
        self.HWms = json.load(open(area_stats_json)) if area_stats_json and os.path.exists(area_stats_json) else {}
        self.train_size, self.max_samples = train_size, max_samples
        self.random_scale, self.rs_min, self.rs_max, self.rs_min_size = random_scale, random_scale_min, random_scale_max, random_scale_min_size
        self.use_largest_part, self.scale_min, self.scale_max = use_largest_part, scale_min, scale_max
        self.shape_jitter, self.mask_threshold = shape_jitter, mask_threshold
        self.filter_min, self.filter_max = instance_filter_min, instance_filter_max
        self.loader = loader or self._pil_loader

    @classmethod
    def from_config(cls, cfg):
        """The constructor call of CopyPasteMapper.from_config (mapper.py:726-745) for INST_POOL_FORMAT 'RGBA', including
        the category filter of InstPool.__init__ (:115-133: keep pool categories whose LVIS frequency is in INST_POOL_FREQ).
        INPUT.INST_POOL_SHARDS (this build's key) switches the per-sample decode to the shard store."""
        with open(cfg.INPUT.INST_POOL_PATH) as f:
            pool = json.load(f)
        freq_path = cfg.MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH
        if freq_path and os.path.exists(freq_path):
            with open(freq_path) as f:
                infos = json.load(f)
            select = {info["id"] - 1 for info in infos if info["frequency"] in cfg.INPUT.INST_POOL_FREQ}
            pool = {k: v for k, v in pool.items() if int(k) in select}
        loader = None
        if cfg.INPUT.get("INST_POOL_SHARDS", ""):
            from .pool_store import PoolStore
            loader = PoolStore(cfg.INPUT.INST_POOL_SHARDS).loader
        return cls(pool, cfg.INPUT.TRAIN_SIZE, area_stats_json=cfg.INPUT.MEAN_STD2_PATH,
                   max_samples=cfg.INPUT.INST_POOL_MAX_SAMPLES, random_scale=cfg.INPUT.RANDOM_SCALE,
                   random_scale_min=cfg.INPUT.RANDOM_SCALE_MIN, random_scale_max=cfg.INPUT.RANDOM_SCALE_MAX,
                   random_scale_min_size=cfg.INPUT.RANDOM_SCALE_MIN_SIZE, use_largest_part=cfg.USE_LARGEST_PART, loader=loader)

    @staticmethod
    def _pil_loader(path):
        from PIL import Image
        mask_path = None
        if path.startswith("*"):
            path = path[1:]
        elif "|" in path:
            path, mask_path = path.split("|")[:2]
        rgba = np.array(Image.open(path).convert("RGBA"))
        if mask_path is not None:
            rgba[:, :, -1] = np.array(Image.open(mask_path))
        return rgba

    def sample_ids(self, nums):
        """_get_cls_balanced_random_samples (mapper.py:263-296): class first, then instance in class."""
        cats_pool = [self.per_cat_pool[c] for c in self.per_cat_pool if len(self.per_cat_pool[c]) > 0]
        ids = []
        if not cats_pool:
            return ids
        for _ in range(nums):
            cid = np.random.randint(0, len(cats_pool))
            ids.append(cats_pool[cid][np.random.randint(0, len(cats_pool[cid]))])
        return ids

    def load_rgba(self, key, image_hw):
        """_load_RGBA (mapper.py:359-456): returns (rgba (h,w,4) uint8, label) or None when rejected."""
        H, W = image_hw
        label = self.data_to_cat[key]
        try:
            rgba = self.loader(key)
        except Exception:
            return None
        if self.random_scale or str(label + 1) not in self.HWms:
            s = np.random.uniform(self.rs_min, self.rs_max)
            tw, th = int(rgba.shape[1] * s), int(rgba.shape[0] * s)
            if tw < self.rs_min_size or th < self.rs_min_size or tw >= W or th >= H:
                return None
        else:
            mean, std = self.HWms[str(label + 1)][:2]
            smin = self.scale_min / H if isinstance(self.scale_min, int) else self.scale_min
            smax = self.scale_max / H if isinstance(self.scale_max, int) else self.scale_max
            area = np.clip(mean + np.random.randn() * std, smin, smax)
            if not key.startswith("*"):
                seg = (rgba[..., 3:] > self.mask_threshold).astype("uint8")
                if self.use_largest_part:
                    seg = largest_connected_component(seg[..., 0])[..., None]
                ys, xs = np.where(seg[..., 0])
                frac = len(ys) / float(seg.shape[0] * seg.shape[1])
                if frac <= self.filter_min or frac >= self.filter_max:
                    return None
                y0, y1, x0, x1 = ys.min(), ys.max(), xs.min(), xs.max()
                if y1 <= y0 or x1 <= x0:
                    return None
                rgba[:, :, 3:] *= seg
                rgba = rgba[y0:y1 + 1, x0:x1 + 1]
            scale = area ** 2 * H * W
            ratio = rgba.shape[1] / rgba.shape[0] * np.random.uniform(1 - self.shape_jitter, 1 + self.shape_jitter)
            tw = np.sqrt(ratio * scale)
            th = tw / ratio
            tw, th = int(tw), int(th)
            if tw < 5 or tw >= W or th < 5 or th >= H:
                return None
        rgba = self._resize(rgba, tw, th)
        if np.random.rand() < 0.5:                                                          # RandomFlip(horizontal)
            rgba = np.ascontiguousarray(rgba[:, ::-1])
        return rgba, label

    @staticmethod
    def _resize(rgba, tw, th):
        """cv2.resize(img_RGBA, (target_W, target_H)) role (mapper.py:446); cv2 is not in this image, PIL bilinear is."""
        from PIL import Image
        return np.array(Image.fromarray(np.ascontiguousarray(rgba), "RGBA").resize((tw, th), Image.BILINEAR))

    @staticmethod
    def random_start_xy(rgba, train_hw):
        """random_start_xy (mapper.py:45-57): offsets so that the patch centre stays inside the canvas."""
        h, w = rgba.shape[:2]
        x_mid, y_mid = (0 + w) / 2.0, (0 + h) / 2.0
        H, W = train_hw
        return np.random.randint(-x_mid, W - x_mid), np.random.randint(-y_mid, H - y_mid)

    def draw(self, image_hw):
        """The random part of get_mix_result('cas_random') + _cat_a_new_image (mapper.py:213-261, :488-496) for one image of size
        image_hw, in the reference's np.random order: the number of samples, the (class, instance) pairs, then EVERY _load_RGBA
        (size prior / scale, jitter, flip), and only then the placement of the instances that survived
        (`datas = [self._load_RGBA(...) ...]` completes before `datas = [random_start_xy(...) ...]` starts).
        Returns (pastes [(rgba (h,w,4) uint8, x0, y0, label)], names).  Pure numpy / PIL: this is what a loader worker runs."""
        H, W = image_hw
        num = np.random.randint(0, self.max_samples)
        keys = [self.dataset[i] for i in self.sample_ids(num)]
        loaded = []
        for key in keys:
            r = self.load_rgba(key, (H, W))
            if r is not None:
                loaded.append((r[0], int(r[1]), str(key)))
        pastes, names = [], []
        for rgba, label, key in loaded:
            x0, y0 = self.random_start_xy(rgba, (H, W))
            pastes.append((rgba, int(x0), int(y0), label))
            names.append(key)
        return pastes, names

    def prepare(self, data):
        """Loader-worker half of get_mix_result: draw + decode + clean + resize + flip + place + pack.  Adds to the mapped sample
        `paste_pack` = dict(flat uint8, desc int32 (K,5), labels int64 (K), K) -- CPU tensors, the compositor's input form --
        and `paste_labels` / `paste_filename_list` (BSGAL's selection reads these).  No device, no libdgx."""
        from ..layers.copy_paste import pack_pastes_host
        H, W = data["image"].shape[-2:]
        pastes, names = self.draw((H, W))
        flat, desc, labels = pack_pastes_host(pastes)
        data = dict(data)
        data["paste_pack"] = {"flat": flat, "desc": desc, "labels": labels, "K": len(pastes)}
        data["paste_labels"], data["paste_filename_list"] = [p[3] for p in pastes], names
        return data

    @staticmethod
    def composite(data, device):
        """Training-process half: the prepared sample's tensors go to `device` (asynchronously when they are pinned) and ONE
        dgx_copy_paste call on the CURRENT stream blends all K patches, updates masks / boxes and drops covered objects.  The
        caller chooses the stream (data/build.py: a side stream, one batch ahead of the training stream)."""
        pk = data["paste_pack"]
        inst = data["instances"]
        H, W = data["image"].shape[-2:]
        up = lambda t: t.to(device, non_blocking=True)     # noqa: E731
        image, gm = up(data["image"]), up(inst.gt_masks.tensor.view(torch.uint8))
        gb, gc = up(inst.gt_boxes.tensor), up(inst.gt_classes)
        packed = PackedPastes(up(pk["flat"]), up(pk["desc"]), up(pk["labels"]), int(pk["K"]))
        out = copy_paste(image, gm, gb, gc, packed, lazy_masks=True)
        ni = Instances((H, W))
        ni.gt_boxes, ni.gt_classes = Boxes(out["boxes"]), out["labels"]
        ni.gt_masks, ni.instance_source = BitMasks(out["masks"].view(torch.bool), index=out["keep"]), out["source"]      # 0/1 bytes: a view; rows through the index
        data = {k: v for k, v in data.items() if k != "paste_pack"}
        data["image"], data["instances"], data["height"], data["width"] = out["image"], ni, H, W
        data["_uploaded"] = (image, gm, gb, gc)      # the un-pasted sample on the device (BSGAL keeps it; copy_paste wrote a clone)
        return data

    def __call__(self, data):
        """get_mix_result('cas_random') (mapper.py:213-261) on one mapped sample dict, both halves in this process."""
        dev = data["image"].device if data["image"].is_cuda else torch.device("cuda")
        data = self.composite(self.prepare(data), dev)
        data.pop("_uploaded", None)
        return data
