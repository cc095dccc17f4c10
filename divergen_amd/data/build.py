"""The real training data path: dataset json -> dataset dicts -> DatasetMapper (+ EfficientDetResizeCrop, RandomFlip) ->
instance copy-paste from the pool (GPU compositor) -> batches with the model's input contract.

Host-side mirror (CPU workers, numpy / PIL as in the reference) of
  DG/divergen/data/datasets/lvis_v1.py:27-119            custom_load_lvis_json  (here without lvis-api: plain json)
  D2/data/datasets/builtin.py + lvis.py                  the lvis_v1_* split table under $DETECTRON2_DATASETS
  D2/data/build.py:get_detection_dataset_dicts           empty-annotation filter
  DG/divergen/data/custom_build_augmentation.py:13-46    build_custom_augmentation
  DG/divergen/data/transforms/custom_augmentation_impl.py:24-72, custom_transform.py:27-91   EfficientDetResizeCrop
  D2/data/transforms/augmentation_impl.py RandomFlip, fvcore HFlipTransform
  DG/divergen/data/dataset_mapper.py:127-256             DatasetMapper.__call__ (training branch, bitmask / polygon masks)
  D2/data/detection_utils.py                             read_image, transform_instance_annotations, annotations_to_instances,
                                                         filter_empty_instances
  DG/divergen/data/custom_build_copypaste_mapper.py:856-958   CopyPasteMapper.__call__ for USE_COPY_METHOD 'syn_copy'
  DG/train_net.py:164-239                                mapper / sampler / loader assembly
Random draws use np.random in the reference's order (scale factor, offset_y, offset_x, flip), so a seeded worker walks the
same augmentation stream.  Polygon rasterisation restates pycocotools' rleFrPoly (maskApi.c; pycocotools is not vendored
by the reference and absent here: PARITY UNPINNED by reference vectors, pinned by known answers in tests/test_host_data.py).
"""
import copy
import json
import logging
import os

import numpy as np
import torch

from ..structures import BitMasks, Boxes, Instances
from ..utils import comm
from .samplers import RepeatFactorTrainingSampler, TrainingSampler

logger = logging.getLogger("divergen_amd")

# (image root, json) relative to $DETECTRON2_DATASETS: D2/data/datasets/builtin.py:_PREDEFINED_SPLITS_LVIS["lvis_v1"] and
# DG/divergen/data/datasets/lvis_v1.py:_CUSTOM_SPLITS_LVIS
SPLITS = {
    "lvis_v1_train": ("coco/", "lvis/lvis_v1_train.json"),
    "lvis_v1_val": ("coco/", "lvis/lvis_v1_val.json"),
    "lvis_v1_test_dev": ("coco/", "lvis/lvis_v1_image_info_test_dev.json"),
    "lvis_v1_test_challenge": ("coco/", "lvis/lvis_v1_image_info_test_challenge.json"),
    "lvis_v1_dev_val": ("coco/", "lvis/lvis_v1_dev_val.json"),
    "lvis_v1_mini_train": ("coco/", "lvis/lvis_v1_mini_train.json"),
    "lvis_v1_train_norare": ("coco/", "lvis/lvis_v1_train_norare.json"),
}
_REGISTERED = {}


def register_lvis_instances(name, json_file, image_root):
    _REGISTERED[name] = (json_file, image_root)


def dataset_files(name):
    if name in _REGISTERED:
        return _REGISTERED[name]
    if name not in SPLITS:
        raise KeyError("Dataset '{}' is not registered! Available: {}".format(name, sorted(list(SPLITS) + list(_REGISTERED))))
    root = os.getenv("DETECTRON2_DATASETS", "datasets")
    image_root, json_file = SPLITS[name]
    return os.path.join(root, json_file), os.path.join(root, image_root)


def load_lvis_json(json_file, image_root):
    """custom_load_lvis_json (lvis_v1.py:27-119): records with file_name / height / width / image_id /
    neg_category_ids (0-based) / not_exhaustive_category_ids / annotations (bbox XYWH, 0-based category_id, polygons)."""
    if not os.path.isfile(json_file):
        raise FileNotFoundError("LVIS annotation file {} not found (DETECTRON2_DATASETS={})".format(
            json_file, os.getenv("DETECTRON2_DATASETS", "datasets")))
    with open(json_file) as f:
        ds = json.load(f)
    cats = sorted(ds["categories"], key=lambda x: x["id"])
    catid2contid = {x["id"]: i for i, x in enumerate(cats)}
    if len(cats) == 1203:
        assert all(catid2contid[x["id"]] == x["id"] - 1 for x in cats)
    img_ann = {}
    for ann in ds.get("annotations", []):
        img_ann.setdefault(ann["image_id"], []).append(ann)
    ann_ids = [a["id"] for v in img_ann.values() for a in v]
    assert len(set(ann_ids)) == len(ann_ids), "Annotation ids in '{}' are not unique".format(json_file)
    out = []
    for img in sorted(ds["images"], key=lambda x: x["id"]):
        rec = {}
        if "file_name" in img:
            fn = img["file_name"]
            if fn.startswith("COCO"):
                fn = fn[-16:]
            rec["file_name"] = os.path.join(image_root, fn)
        elif "coco_url" in img:
            rec["file_name"] = os.path.join(image_root, img["coco_url"][30:])     # .../train2017/000000391895.jpg
        for k in ("height", "width"):
            if k in img:
                rec[k] = img[k]
        rec["not_exhaustive_category_ids"] = img.get("not_exhaustive_category_ids", [])
        rec["neg_category_ids"] = [catid2contid[x] for x in img.get("neg_category_ids", [])]
        if "pos_category_ids" in img:
            rec["pos_category_ids"] = [catid2contid[x] for x in img["pos_category_ids"]]
        image_id = rec["image_id"] = img["id"]
        objs = []
        for anno in img_ann.get(image_id, []):
            assert anno["image_id"] == image_id
            if anno.get("iscrowd", 0) > 0:
                continue
            obj = {"bbox": anno["bbox"], "bbox_mode": "XYWH_ABS", "category_id": catid2contid[anno["category_id"]]}
            if "segmentation" in anno:
                segm = anno["segmentation"]
                assert len(segm) > 0
                obj["segmentation"] = segm
            objs.append(obj)
        rec["annotations"] = objs
        out.append(rec)
    logger.info("Loaded {} images in the LVIS v1 format from {}".format(len(out), json_file))
    return out


def get_detection_dataset_dicts(names, filter_empty=True):
    """D2/data/build.py:get_detection_dataset_dicts for instance datasets: concatenate, drop images without annotations."""
    if isinstance(names, str):
        names = [names]
    dicts = []
    for n in names:
        d = load_lvis_json(*dataset_files(n))
        assert len(d), "Dataset '{}' is empty!".format(n)
        dicts.extend(d)
    if filter_empty:
        before = len(dicts)
        dicts = [d for d in dicts if any(a.get("iscrowd", 0) == 0 for a in d["annotations"])]
        logger.info("Removed {} images with no usable annotations. {} images left.".format(before - len(dicts), len(dicts)))
    return dicts


# ----------------------------------------------------------------------------------------------- transforms
class EfficientDetResizeCropTransform:
    """custom_transform.py:27-91 (uint8 images through PIL, coordinates scaled then shifted)."""

    def __init__(self, scaled_h, scaled_w, offset_y, offset_x, img_scale, target_size):
        self.scaled_h, self.scaled_w, self.offset_y, self.offset_x = scaled_h, scaled_w, offset_y, offset_x
        self.img_scale, self.target_size = img_scale, target_size

    def apply_image(self, img, nearest=False):
        from PIL import Image
        assert img.dtype == np.uint8
        pil = Image.fromarray(img).resize((self.scaled_w, self.scaled_h), Image.NEAREST if nearest else Image.BILINEAR)
        ret = np.asarray(pil)
        right = min(self.scaled_w, self.offset_x + self.target_size[1])
        lower = min(self.scaled_h, self.offset_y + self.target_size[0])
        return ret[self.offset_y:lower, self.offset_x:right]

    def apply_coords(self, coords):
        coords[:, 0] = coords[:, 0] * self.img_scale
        coords[:, 1] = coords[:, 1] * self.img_scale
        coords[:, 0] -= self.offset_x
        coords[:, 1] -= self.offset_y
        return coords


class HFlipTransform:
    def __init__(self, width):
        self.width = width

    def apply_image(self, img, nearest=False):
        return np.flip(img, axis=1)

    def apply_coords(self, coords):
        coords[:, 0] = self.width - coords[:, 0]
        return coords


class NoOpTransform:
    def apply_image(self, img, nearest=False):
        return img

    def apply_coords(self, coords):
        return coords


class EfficientDetResizeCrop:
    """custom_augmentation_impl.py:24-72: random scale of the target square, then a random crop offset."""

    def __init__(self, size, scale):
        self.target_size = (size, size) if size > 0 else None
        self.scale = scale

    def get_transform(self, img):
        scale_factor = np.random.uniform(*self.scale)
        width, height = img.shape[1], img.shape[0]
        if self.target_size is not None:
            img_scale = min(scale_factor * self.target_size[0] / height, scale_factor * self.target_size[1] / width)
        else:
            img_scale = scale_factor
        scaled_h = max(1, int(height * img_scale))
        scaled_w = max(1, int(width * img_scale))
        if self.target_size is not None:
            offset_y, offset_x, target_size = scaled_h - self.target_size[0], scaled_w - self.target_size[1], self.target_size
        else:
            offset_y, offset_x, target_size = 0, 0, (scaled_h, scaled_w)
        offset_y = int(max(0.0, float(offset_y)) * np.random.uniform(0, 1))
        offset_x = int(max(0.0, float(offset_x)) * np.random.uniform(0, 1))
        return EfficientDetResizeCropTransform(scaled_h, scaled_w, offset_y, offset_x, img_scale, target_size)


class ResizeTransform:
    """fvcore / D2 ResizeTransform: PIL bilinear for uint8 images, coordinates scaled per axis."""

    def __init__(self, h, w, new_h, new_w):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w

    def apply_image(self, img, nearest=False):
        from PIL import Image
        assert img.shape[:2] == (self.h, self.w)
        return np.asarray(Image.fromarray(img).resize((self.new_w, self.new_h), Image.NEAREST if nearest else Image.BILINEAR))

    def apply_coords(self, coords):
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords


class ResizeShortestEdge:
    """D2/data/transforms/augmentation_impl.py ResizeShortestEdge: shortest edge to `short` (one of a list, np.random.choice),
    longest edge capped at max_size; the evaluation input of D2's default DatasetMapper (INPUT.MIN_SIZE_TEST / MAX_SIZE_TEST)."""

    def __init__(self, short, max_size, sample_style="choice"):
        self.short = (short, short) if isinstance(short, int) else tuple(short)
        self.max_size, self.is_range = max_size, sample_style == "range"

    def get_transform(self, img):
        h, w = img.shape[:2]
        size = np.random.randint(self.short[0], self.short[1] + 1) if self.is_range else np.random.choice(self.short)
        if size == 0:
            return NoOpTransform()
        scale = size * 1.0 / min(h, w)
        newh, neww = (size, scale * w) if h < w else (scale * h, size)
        if max(newh, neww) > self.max_size:
            scale = self.max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return ResizeTransform(h, w, int(newh + 0.5), int(neww + 0.5))


class RandomFlip:
    """D2 RandomFlip(prob=0.5, horizontal=True): one np.random.uniform draw."""

    def __init__(self, prob=0.5):
        self.prob = prob

    def get_transform(self, img):
        return HFlipTransform(img.shape[1]) if np.random.uniform() < self.prob else NoOpTransform()


def build_custom_augmentation(cfg, is_train):
    """custom_build_augmentation.py:13-46 for INPUT.CUSTOM_AUG == 'EfficientDetResizeCrop' (the shipped configs)."""
    if cfg.INPUT.CUSTOM_AUG != "EfficientDetResizeCrop":
        raise NotImplementedError("INPUT.CUSTOM_AUG '{}': only EfficientDetResizeCrop (the shipped configs) is built".format(cfg.INPUT.CUSTOM_AUG))
    if is_train:
        aug = [EfficientDetResizeCrop(cfg.INPUT.TRAIN_SIZE, tuple(cfg.INPUT.SCALE_RANGE)), RandomFlip()]
    else:
        aug = [EfficientDetResizeCrop(cfg.INPUT.TEST_SIZE, (1, 1))]
    return aug


# ----------------------------------------------------------------------------------------------- polygons -> bitmask
def polygon_crossings(xy, h, w):
    """pycocotools maskApi.c rleFrPoly up to its sort: the column-major flat positions at which the fill toggles, for one polygon
    given as a flat [x0, y0, x1, y1, ...] list.  Restated from the published C source -- 5x upsampled integer polygon, all
    boundary points by the longer-axis walk, the points where the boundary crosses a pixel-column centre -- as numpy array
    arithmetic (a loader worker rasterises ~12 polygons per image; the per-point Python loop of round 1-5 cost 5 ms each)."""
    k = len(xy) // 2
    scale = 5.0
    pts = np.asarray(xy, dtype=np.float64).reshape(k, 2)
    x = (scale * pts[:, 0] + 0.5).astype(np.int64)              # C (int) cast: truncation toward zero, as astype does
    y = (scale * pts[:, 1] + 0.5).astype(np.int64)
    x, y = np.append(x, x[0]), np.append(y, y[0])
    us, vs = [], []
    for j in range(k):
        xs, xe, ys, ye = int(x[j]), int(x[j + 1]), int(y[j]), int(y[j + 1])
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            s = 0.0 if dx == 0 else (ye - ys) / dx
            t = np.arange(dx, -1, -1) if flip else np.arange(dx + 1)
            us.append(t + xs)
            vs.append((ys + s * t + 0.5).astype(np.int64))
        else:
            s = 0.0 if dy == 0 else (xe - xs) / dy
            t = np.arange(dy, -1, -1) if flip else np.arange(dy + 1)
            vs.append(t + ys)
            us.append((xs + s * t + 0.5).astype(np.int64))
    u, v = np.concatenate(us), np.concatenate(vs)
    # points along the y-boundary, downsampled
    u1, u0, v1, v0 = u[1:], u[:-1], v[1:], v[:-1]
    sel = u1 != u0
    u1, u0, v1, v0 = u1[sel], u0[sel], v1[sel], v0[sel]
    xd = (np.where(u1 < u0, u1, u1 - 1).astype(np.float64) + 0.5) / scale - 0.5
    ok = (np.floor(xd) == xd) & (xd >= 0) & (xd <= w - 1)
    yd = (np.where(v1 < v0, v1, v0).astype(np.float64) + 0.5) / scale - 0.5
    yd = np.ceil(np.clip(yd, 0.0, float(h)))
    return xd[ok].astype(np.int64) * h + yd[ok].astype(np.int64)


def polygon_to_rle_counts(xy, h, w):
    """The run lengths rleFrPoly returns (column-major, first run = zeros): sorted crossings -> differences -> zero-length
    runs merged into their neighbours (the first run may be zero).  Kept for the known-answer tests; the mapper fills masks
    from the crossings directly (polygons_to_bitmask)."""
    a = sorted(int(p) for p in polygon_crossings(xy, h, w))
    a.append(h * w)
    p = 0
    for j in range(len(a)):
        t = a[j]
        a[j] -= p
        p = t
    b = []
    j = 0
    if a:
        b.append(a[0])
        j = 1
    while j < len(a):
        if a[j] > 0:
            b.append(a[j])
            j += 1
        else:
            j += 1
            if j < len(a):
                b[-1] += a[j]
                j += 1
    return b


def polygons_to_bitmask(polygons, h, w, out=None):
    """D2/structures/masks.py:polygons_to_bitmask: union (rleMerge) of the polygons of one instance, decoded (h, w) bool
    (row-major; written into `out` when given).  Decoding the run lengths of rleFrPoly = the parity of the number of crossings
    at or before each column-major position (a zero-length run is two toggles at one position); only the columns between the
    first and the last crossing are computed, and transposed into the row-major mask."""
    m = np.zeros((h, w), dtype=bool) if out is None else out
    for poly in polygons:
        pos = polygon_crossings([float(c) for c in poly], h, w)
        pos = pos[pos < h * w]
        if pos.size == 0:
            continue
        c0, c1 = int(pos.min()) // h, int(pos.max()) // h
        tog = np.zeros((c1 - c0 + 1) * h, dtype=np.uint8)
        np.add.at(tog, pos - c0 * h, 1)                      # a few hundred crossings
        tog &= 1
        m[:, c0:c1 + 1] |= np.bitwise_xor.accumulate(tog).view(bool).reshape(c1 - c0 + 1, h).T
        if pos.size & 1:                      # an odd number of crossings: the last run of ones reaches the end of the canvas
            m[:, c1 + 1:] = True
    return m


# ----------------------------------------------------------------------------------------------- mapper
def read_image(file_name, fmt="BGR"):
    """D2/data/detection_utils.py:read_image: PIL decode, EXIF orientation applied, RGB -> requested channel order."""
    from PIL import Image, ImageOps
    with open(file_name, "rb") as f:
        image = Image.open(f)
        image = ImageOps.exif_transpose(image)
        image = np.asarray(image.convert("RGB"))
    return image[:, :, ::-1] if fmt == "BGR" else image


def transform_instance_annotations(anno, transforms, image_size):
    """D2 transform_instance_annotations: XYWH -> XYXY box through the 4 corners, clipped; polygons point-wise."""
    x, y, bw, bh = anno["bbox"]
    corners = np.array([[x, y], [x + bw, y], [x, y + bh], [x + bw, y + bh]], dtype=np.float64)
    for t in transforms:
        corners = t.apply_coords(corners)
    box = np.concatenate([corners.min(0), corners.max(0)])
    anno["bbox"] = np.minimum(np.maximum(box, 0), np.array(list(image_size) + list(image_size))[::-1])
    anno["bbox_mode"] = "XYXY_ABS"
    if "segmentation" in anno:
        polys = []
        for p in anno["segmentation"]:
            c = np.asarray(p, dtype=np.float64).reshape(-1, 2)
            for t in transforms:
                c = t.apply_coords(c)
            polys.append(c.reshape(-1))
        anno["segmentation"] = polys
    return anno


def annotations_to_instances(annos, image_size):
    """D2 annotations_to_instances with mask_format 'bitmask'."""
    h, w = image_size
    target = Instances(image_size)
    boxes = np.stack([a["bbox"] for a in annos]) if annos else np.zeros((0, 4))
    target.gt_boxes = Boxes(torch.as_tensor(boxes, dtype=torch.float32).reshape(-1, 4))
    target.gt_classes = torch.tensor([int(a["category_id"]) for a in annos], dtype=torch.int64)
    if annos and "segmentation" in annos[0]:
        masks = np.zeros((len(annos), h, w), dtype=bool)          # one contiguous block: it is pinned and uploaded as it is
        for i, a in enumerate(annos):
            polygons_to_bitmask(a["segmentation"], h, w, out=masks[i])
        target.gt_masks = BitMasks(torch.from_numpy(masks))
    return target


def filter_empty_instances(instances, box_threshold=1e-5):
    """D2 filter_empty_instances(by_box=True, by_mask=True)."""
    b = instances.gt_boxes.tensor
    keep = ((b[:, 2] - b[:, 0]) > box_threshold) & ((b[:, 3] - b[:, 1]) > box_threshold)
    if instances.has("gt_masks"):
        gm = instances.gt_masks.tensor
        keep &= torch.from_numpy(gm.numpy().reshape(gm.shape[0], -1).any(1)) if not gm.is_cuda else gm.flatten(1).any(1)
    return instances[keep]


class DatasetMapper:
    """DG/divergen/data/dataset_mapper.py:127-256, training branch without proposals / keypoints / sem-seg."""

    def __init__(self, cfg, is_train=True, augmentations=None):
        self.is_train = is_train
        if augmentations is None:
            # training: DG/train_net.py:181-182 (custom augmentation); evaluation with TEST_INPUT_TYPE 'default': D2's own
            # mapper, i.e. ResizeShortestEdge(MIN_SIZE_TEST, MAX_SIZE_TEST) (DG/train_net.py:93-97)
            if is_train or cfg.INPUT.TEST_INPUT_TYPE != "default":
                augmentations = build_custom_augmentation(cfg, is_train)
            else:
                augmentations = [ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, "choice")]
        self.augmentations = augmentations
        self.image_format = cfg.INPUT.FORMAT
        self.use_instance_mask = cfg.MODEL.MASK_ON
        if cfg.INPUT.MASK_FORMAT != "bitmask" and cfg.MODEL.MASK_ON and is_train:
            logger.warning("INPUT.MASK_FORMAT '%s': masks are rasterised to bitmasks on the loader side either way", cfg.INPUT.MASK_FORMAT)

    def __call__(self, dataset_dict):
        d = copy.deepcopy(dataset_dict)
        image = read_image(d["file_name"], self.image_format)
        if "width" in d and (d["width"], d["height"]) != (image.shape[1], image.shape[0]):
            raise ValueError("Mismatched image shape for {}: got {}, expect {}".format(
                d["file_name"], (image.shape[1], image.shape[0]), (d["width"], d["height"])))
        d.setdefault("width", image.shape[1])
        d.setdefault("height", image.shape[0])
        transforms = []
        for aug in self.augmentations:
            t = aug.get_transform(image)
            image = t.apply_image(image)
            transforms.append(t)
        image_shape = image.shape[:2]
        d["image"] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        if not self.is_train:
            return d
        if "annotations" in d:
            annos = []
            for obj in d.pop("annotations"):
                if obj.get("iscrowd", 0) != 0:
                    continue
                if not self.use_instance_mask:
                    obj.pop("segmentation", None)
                annos.append(transform_instance_annotations(obj, transforms, image_shape))
            inst = filter_empty_instances(annotations_to_instances(annos, image_shape))
            if not inst.has("gt_masks"):
                inst.gt_masks = BitMasks(torch.zeros(0, image_shape[0], image_shape[1], dtype=torch.bool))
            d["instances"] = inst
        return d


class CopyPasteMapper:
    """CopyPasteMapper.__call__ (mapper.py:856-958) for the shipped configuration: USE_COPY_METHOD 'syn_copy' with an instance
    pool -> InstPool.get_mix_result (divergen_amd/data/copypaste.py, pixels on the GPU compositor); the self-copy branch has
    no mix results in that configuration, so SimpleCopyPaste returns its input (custom_copypaste.py:254-259)."""

    def __init__(self, mapper, cfg):
        self.mapper = mapper
        self.use_scp = cfg.INPUT.USE_SCP
        self.num_src = cfg.INPUT.SCP_NUM_SRC
        self.method = cfg.INPUT.USE_COPY_METHOD
        self.inst_pool = None
        self.dataset = None
        self.pack = True               # one blob per sample across the process boundary (pack_sample); the loader's finish() unpacks
        self.ring = None               # SlotRing the workers wrote the blobs into (build_detection_train_loader)
        if self.method not in ("none", "syn_copy"):
            raise NotImplementedError("INPUT.USE_COPY_METHOD '{}': only 'syn_copy' / 'none' (the shipped configs) are built".format(self.method))
        if cfg.INPUT.INST_POOL and self.method == "syn_copy":
            from .copypaste import InstPool
            if cfg.INPUT.INST_POOL_SAMPLE_TYPE != "cas_random" or cfg.INPUT.INST_POOL_FORMAT != "RGBA":
                raise NotImplementedError("only INST_POOL_FORMAT 'RGBA' with INST_POOL_SAMPLE_TYPE 'cas_random' is built")
            self.inst_pool = InstPool.from_config(cfg)

        # BSGAL (BS/bsgal/data/custom_build_copypaste_mapper.py:237-297, :916-930, :958-963, :1038-1057): keep the un-pasted
        # sample and add a held-out image that shows one of the pasted classes
        self.active_select = bool(cfg.INPUT.get("ACTIVE_SELECT", False))
        self.active_test = cfg.MODEL.get("ACTIVE_TEST", "select")
        self.active_test_one = cfg.MODEL.get("ACTIVE_TEST_INS", "one") == "one" and "one_class" in cfg.INPUT.INST_POOL_SAMPLE_TYPE
        self.per_cat_pool_real = None

    def set_dataset(self, dataset):
        self.dataset = dataset
        if self.active_select:
            pool = {}
            for i, d in enumerate(dataset):
                for a in d.get("annotations", []):
                    pool.setdefault(a["category_id"], set()).add(i)
            self.per_cat_pool_real = {k: sorted(v) for k, v in pool.items()}

    def _held_out(self, paste_labels, own_labels):
        """The test image of one sample (:260-284): a class among the pasted ones ('select'), else among the image's own,
        else any; re-drawn until some training image shows it; then one of those images through the plain mapper."""
        ncls = 1203
        if not any(self.per_cat_pool_real.values()):
            raise ValueError("INPUT.ACTIVE_SELECT needs a training set with annotations: no category has an image to hold out")
        if self.active_test == "select":
            cand = sorted(set(paste_labels)) or sorted(set(own_labels))
            cls = int(np.random.choice(cand)) if cand else int(np.random.choice(range(ncls)))
        else:
            cls = int(np.random.choice(range(ncls)))
        while not self.per_cat_pool_real.get(cls):
            cls = int(np.random.choice(range(ncls)))
        pool = self.per_cat_pool_real[cls]
        idx = pool[np.random.randint(0, len(pool))]
        if self.active_test == "random_img":
            idx = np.random.randint(0, len(self.dataset))
        test = self.mapper(self.dataset[idx])
        if self.active_test_one:
            test["instances"] = test["instances"][test["instances"].gt_classes == cls]
        return cls, test

    def __call__(self, dataset_dict):
        """What a LOADER WORKER runs -- all of CopyPasteMapper.__call__ (mapper.py:856-958) except the pixel blend: decode +
        augment + rasterise, the self-copy index draw (:877), the instance pool's draws / decode / largest component / resize /
        flip / placement packed for the compositor (InstPool.prepare), BSGAL's held-out image.  CPU tensors only."""
        result = self.mapper(dataset_dict)
        if "instances" not in result or not result["instances"].has("gt_masks"):        # mapper.py:862-864
            return result
        if self.use_scp and self.dataset is not None:
            for _ in range(self.num_src):
                np.random.randint(0, len(self.dataset))
        if self.inst_pool is None:
            return result
        result = self.inst_pool.prepare(result)
        if self.active_select:
            cls, test = self._held_out(result.get("paste_labels", []), result["instances"].gt_classes.tolist())
            result["test_image"], result["test_instances"] = test["image"], test["instances"]
            result["test_image_class"], result["test_file_name"] = cls, test.get("file_name")
        return pack_sample(result) if self.pack else result

    def finish(self, result, device):
        """What the TRAINING PROCESS runs on one worker result, on the current stream: upload (asynchronous from pinned memory) and
        the compositor kernel; with INPUT.ACTIVE_SELECT the un-pasted sample stays available as origin_* (it is the uploaded
        input: the compositor writes a copy)."""
        from .copypaste import InstPool
        dev = torch.device(device)
        result = unpack_sample(result, dev, self.ring)
        if "instances" not in result:
            result["image"] = result["image"].to(dev, non_blocking=True)
            return result
        if "paste_pack" in result and dev.type == "cuda":
            out = InstPool.composite(result, dev)
            img, gm, gb, gc = out.pop("_uploaded")
            origin = Instances(tuple(img.shape[-2:]), gt_boxes=Boxes(gb), gt_classes=gc, gt_masks=BitMasks(gm.view(torch.bool)))
        else:
            out = {k: v for k, v in result.items() if k != "paste_pack"}
            out["image"], out["instances"] = out["image"].to(dev, non_blocking=True), out["instances"].to(dev)
            img, origin = out["image"], out["instances"]
        if self.active_select and "test_image" in out:
            out["origin_image"], out["origin_instances"] = img, origin
            if not origin.has("instance_source"):
                origin.instance_source = torch.zeros(len(origin), dtype=torch.int64, device=dev)
            out["test_image"], out["test_instances"] = out["test_image"].to(dev, non_blocking=True), out["test_instances"].to(dev)
        return out


# ---- one blob per sample.  A worker result used to cross the process boundary as 7 tensors per image (image, masks, boxes, classes,
# paste patches, paste descriptors, paste labels): 7 shared-memory segments whose file descriptors are handed over one authenticated
# socket round trip each (Python, under the GIL, in the training process's pin thread), 7 pinned copies, 7 host -> device copies issued
# by the training thread.  The worker now lays them out in ONE uint8 tensor (sections 64-byte aligned) + a small layout list; the
# training process pins and uploads that one tensor and takes the fields as VIEWS of the device copy.
_BLOB_FIELDS = (("image", lambda d: d["image"]), ("gt_masks", lambda d: d["instances"].gt_masks.tensor.view(torch.uint8)),
                ("gt_boxes", lambda d: d["instances"].gt_boxes.tensor), ("gt_classes", lambda d: d["instances"].gt_classes),
                ("flat", lambda d: d["paste_pack"]["flat"]), ("desc", lambda d: d["paste_pack"]["desc"]), ("labels", lambda d: d["paste_pack"]["labels"]))


def pack_sample(d):
    """Worker side: the sample's tensors -> d['blob'] (uint8) + d['blob_layout'] [(name, dtype, shape, byte offset)]; the tensor
    entries themselves are dropped.  Samples without paste_pack / instances pass through unchanged."""
    if "paste_pack" not in d or "instances" not in d or not d["instances"].has("gt_masks"):
        return d
    parts, layout, off = [], [], 0
    for name, get in _BLOB_FIELDS:
        t = get(d).contiguous()
        raw = t.view(-1).view(torch.uint8) if t.numel() else torch.zeros(0, dtype=torch.uint8)
        layout.append((name, str(t.dtype).replace("torch.", ""), tuple(t.shape), off))
        pad = (-raw.numel()) % 64
        parts.append(raw)
        if pad:
            parts.append(torch.zeros(pad, dtype=torch.uint8))
        off += raw.numel() + pad
    out = {k: v for k, v in d.items() if k not in ("image", "instances", "paste_pack")}
    out["blob"], out["blob_layout"] = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.uint8), layout
    out["blob_hw"], out["blob_K"] = tuple(d["image"].shape[-2:]), int(d["paste_pack"]["K"])
    extra = {k: v for k, v in d["instances"].get_fields().items() if k not in ("gt_masks", "gt_boxes", "gt_classes")}
    if extra:
        out["blob_extra_fields"] = extra
    return out


def unpack_sample(d, device, ring=None):
    """Training-process side: d['blob'] (or the slot of the page-locked ring it was written into) goes to `device` in ONE asynchronous
    copy; image / instances / paste_pack come back as views of it."""
    if "blob" not in d:
        return d
    if d.get("blob_slot") is not None:
        from .. import _lib as L
        off, n = d["blob_slot"]
        blob = torch.empty(n, dtype=torch.uint8, device=device)
        L.check(L.lib().dgx_memcpy_h2d_async(blob.data_ptr(), ring.buf.data_ptr() + off, n, L.stream()), "dgx_memcpy_h2d_async")
    else:
        blob = d["blob"].to(device, non_blocking=True)
    f = {}
    for name, dt, shape, off in d["blob_layout"]:
        dtype = getattr(torch, dt)
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        f[name] = blob[off:off + n].view(dtype).view(shape)
    out = {k: v for k, v in d.items() if not k.startswith("blob")}
    out["image"] = f["image"]
    inst = Instances(tuple(d["blob_hw"]), gt_boxes=Boxes(f["gt_boxes"]), gt_classes=f["gt_classes"], gt_masks=BitMasks(f["gt_masks"].view(torch.bool)))
    for k, v in d.get("blob_extra_fields", {}).items():
        inst.set(k, v.to(device) if hasattr(v, "to") else v)
    out["instances"] = inst
    out["paste_pack"] = {"flat": f["flat"], "desc": f["desc"], "labels": f["labels"], "K": d["blob_K"]}
    return out


class SlotRing:
    """Page-locked shared memory the loader workers write their sample blobs into (round 6).

    torch's DataLoader pins in a THREAD of the training process: per batch it unpickles the tensors (one authenticated socket round
    trip per shared-memory segment, Python under the GIL), allocates pinned memory and copies.  With the Swin blocks replayed as
    graphs the step's only host-bound stretch is the RoI heads behind the proposal sampler's device->host read, and every time that
    thread takes the GIL there the GPU waits: +1.9 ms per step measured (27.6 against 25.7 ms).  Here ONE region is allocated in shared
    memory before the workers fork and page-locked once (dgx_host_register); worker w owns slots [w * per_worker, (w + 1) * per_worker)
    and writes its k-th sample into slot k % per_worker; the training process receives (offset, bytes) and uploads straight from the
    slot (dgx_memcpy_h2d_async) -- no pin thread, no per-batch allocation, nothing but a few integers crosses the queue.
    Reuse is safe by the DataLoader's own bookkeeping: a worker holds at most `prefetch_factor` batches that the training process has
    not consumed, so with (prefetch_factor + 2) batches of slots a slot is rewritten no earlier than two further batches of that worker
    have been consumed -- 2 x num_workers steps after its upload was issued."""

    def __init__(self, num_workers, per_worker, slot_bytes):
        from .. import _lib as L
        self.num_workers, self.per_worker, self.slot_bytes = int(num_workers), int(per_worker), (int(slot_bytes) + 4095) // 4096 * 4096
        n = self.num_workers * self.per_worker * self.slot_bytes
        st = os.statvfs("/dev/shm") if os.path.isdir("/dev/shm") else None
        if st is not None and st.f_bavail * st.f_frsize < n + (1 << 30):
            raise MemoryError("SlotRing: /dev/shm has %.1f GB free, %.1f GB needed" % (st.f_bavail * st.f_frsize / 1e9, n / 1e9))
        self.buf = torch.empty(n, dtype=torch.uint8).share_memory_()
        L.check(L.lib().dgx_host_register(self.buf.data_ptr(), n), "dgx_host_register")
        self._registered, self._pid = True, os.getpid()

    def offset(self, worker, k):
        return (worker * self.per_worker + k % self.per_worker) * self.slot_bytes

    def close(self):
        if self.__dict__.get("_registered") and self.__dict__.get("_pid") == os.getpid():      # (never from a forked worker)
            from .. import _lib as L
            L.lib().dgx_host_unregister(self.buf.data_ptr())
            self._registered = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _RingCollate:
    """collate_fn (runs in the WORKER): the samples' blobs into this worker's next slots; what is returned carries (offset, bytes)."""

    def __init__(self, ring):
        self.ring, self.k = ring, 0

    def __call__(self, batch):
        info = torch.utils.data.get_worker_info()
        if info is None or self.ring is None:
            return batch
        for d in batch:
            blob = d.get("blob")
            if blob is None or blob.numel() > self.ring.slot_bytes:      # an image with more ground truth than a slot holds: the ordinary way
                continue
            off, n = self.ring.offset(info.id, self.k), int(blob.numel())
            self.k += 1
            self.ring.buf[off:off + n].copy_(blob)
            d["blob"], d["blob_slot"] = None, (off, n)
        return batch


class _MapDataset(torch.utils.data.Dataset):
    def __init__(self, dicts, fn):
        self.dicts, self.fn = dicts, fn

    def __len__(self):
        return len(self.dicts)

    def __getitem__(self, i):
        return self.fn(self.dicts[i])


def _worker_init(worker_id, base_seed, in_worker=True):
    seed = (base_seed + worker_id) % (2 ** 31)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if in_worker:
        torch.set_num_threads(1)      # 16 workers per GPU: one core each


def _tensors_of(obj):
    """Every device tensor reachable from one batched-input dict (image, Instances fields, lazily indexed masks)."""
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)
    elif isinstance(obj, Instances):
        yield from _tensors_of(obj.get_fields())
    elif isinstance(obj, Boxes):
        yield obj.tensor
    elif isinstance(obj, BitMasks):
        yield obj._base
        if obj._index is not None:
            yield obj._index


class BatchAhead:
    """Host batches (lists of worker results, CPU tensors) -> device batches, prepared ONE BATCH AHEAD on a side HIP stream.

    The copy-paste compositor is the data-loading side of the step (the reference runs it in loader workers): it depends on
    nothing the optimizer produces.  So while step t trains, the batch of step t+1 is uploaded (pinned memory, asynchronous)
    and composited on `side`, and its one data-dependent shape (objects that end up fully covered are dropped: one
    device->host count per image) is read back from THAT stream instead of draining the training stream.  __next__ hands over
    batch t (the training stream waits for its event and takes ownership of the tensors) and then issues batch t+1 -- before the
    caller's forward: the host is ahead of the GPU at that point, whereas after the forward's device->host read every host
    microsecond is GPU idle time.  bench.py's timed step is this class fed with fixed host batches; train_net.py's is this class
    fed by the DataLoader."""

    def __init__(self, host_batches, finish, device):
        self.it, self.finish, self.device = iter(host_batches), finish, torch.device(device)
        self.side = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self.nxt, self.started = None, False
        self.wait_s = 0.0                   # seconds __next__ last spent blocked on the workers (the loop's real data_time)

    def _compose(self):
        import time
        t0 = time.perf_counter()
        try:
            host = next(self.it)
        except StopIteration:
            return None
        self.wait_s = time.perf_counter() - t0
        if self.side is None:
            return [self.finish(d, self.device) for d in host], None
        with torch.cuda.stream(self.side):
            batch = [self.finish(d, self.device) for d in host]
            ev = torch.cuda.Event()
            ev.record(self.side)
        return batch, ev

    def __iter__(self):
        return self

    def __next__(self):
        if not self.started:
            self.started, self.nxt = True, self._compose()
        cur = self.nxt
        if cur is None:
            raise StopIteration
        batch, ev = cur
        if ev is not None:
            cs = torch.cuda.current_stream(self.device)
            cs.wait_event(ev)                       # the training stream consumes the composited tensors ...
            for t in _tensors_of(batch):            # ... and owns them from here on (allocator stream bookkeeping)
                if t.is_cuda:
                    t.record_stream(cs)
        self.nxt = self._compose()
        return batch


# Intra-op CPU threads of the TRAINING process while a loader runs.  torch sizes its thread pool by the visible cores (128 on the
# 256-core MI355X host), but the process shares a cgroup CPU quota with its workers (16 CPUs on the box measured: /sys/fs/cgroup/cpu.max
# 1600000 100000): the pin thread's 12-30 MB copies fanned out over 128 OpenMP threads whose spin-waits burnt the quota, the kernel
# throttled the whole cgroup, and the training thread's host time per step DOUBLED (profiles/r06_loader_ab.txt: 57.0 ms/step with the
# default pool, 28.5 with 4 threads, same workers, same batches; 25.2 with the same batches and no loader running).  The training
# process has no CPU tensor work of its own besides those copies.  bench.py --loader-dev main_threads=N overrides (0 = leave alone).
DEV = {"pin": "ring", "strategy": None, "main_threads": 4}      # pin: "ring" (SlotRing) | "loader" (the DataLoader's pin thread) | "none"


def build_detection_train_loader(cfg, per_gpu, device, seed):
    """DG/train_net.py:164-239 + D2/data/build.py:build_detection_train_loader: dataset dicts, sampler by
    DATALOADER.SAMPLER_TRAIN, the whole mapper (copy-paste preparation included) in DATALOADER.NUM_WORKERS worker processes, each
    sample written by its worker into a slot of a page-locked shared-memory ring (SlotRing; the DataLoader's pin thread when the ring
    cannot be had); the training process only uploads and runs the compositor kernel, one batch ahead on a side stream (BatchAhead)."""
    import functools
    dicts = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
    name = cfg.DATALOADER.SAMPLER_TRAIN
    if name == "TrainingSampler":
        sampler = TrainingSampler(len(dicts), seed=seed)
    elif name == "RepeatFactorTrainingSampler":
        rf = RepeatFactorTrainingSampler.repeat_factors_from_category_frequency(dicts, cfg.DATALOADER.REPEAT_THRESHOLD)
        sampler = RepeatFactorTrainingSampler(rf, seed=seed)
    else:
        raise ValueError("Unknown training sampler: {}".format(name))
    mapper = CopyPasteMapper(DatasetMapper(cfg, True), cfg)
    mapper.set_dataset(dicts)
    rank_seed = seed * 1009 + comm.get_rank() * 131
    nw = cfg.DATALOADER.NUM_WORKERS
    on_gpu = torch.device(device).type == "cuda"
    if DEV["strategy"]:
        torch.multiprocessing.set_sharing_strategy(DEV["strategy"])
    if DEV["main_threads"] and on_gpu and nw > 0:
        torch.set_num_threads(min(torch.get_num_threads(), int(DEV["main_threads"])))
    pin_thread, ring = on_gpu and DEV["pin"] == "loader", None
    if on_gpu and nw > 0 and DEV["pin"] == "ring":
        # one slot holds an image with up to ~45 ground-truth masks at TRAIN_SIZE^2 (more: that sample takes the ordinary way)
        size = int(cfg.INPUT.TRAIN_SIZE)
        try:
            ring = SlotRing(nw, (int(cfg.DATALOADER.PREFETCH_FACTOR) + 2) * per_gpu, size * size * 48 + (8 << 20))
        except Exception as e:
            logger.warning("loader: no page-locked slot ring (%s); falling back to the DataLoader's pin thread", e)
            pin_thread = True
    mapper.ring = ring
    loader = torch.utils.data.DataLoader(
        _MapDataset(dicts, mapper), sampler=sampler, batch_size=per_gpu, drop_last=True, num_workers=nw,
        collate_fn=_RingCollate(ring) if ring is not None else _identity,
        worker_init_fn=functools.partial(_worker_init, base_seed=rank_seed), pin_memory=pin_thread,
        prefetch_factor=cfg.DATALOADER.PREFETCH_FACTOR if nw > 0 else None, persistent_workers=nw > 0)
    if nw == 0:
        _worker_init(0, rank_seed, in_worker=False)
    return BatchAhead(loader, mapper.finish, device)


def _identity(batch):
    return batch


def build_detection_test_loader(cfg, dataset_name, device):
    """D2/data/build.py:build_detection_test_loader: every image once, sharded over ranks by InferenceSampler, batch size 1,
    test-time mapper (no annotations)."""
    from .samplers import InferenceSampler
    dicts = get_detection_dataset_dicts([dataset_name], filter_empty=False)
    mapper = DatasetMapper(cfg, False)
    loader = torch.utils.data.DataLoader(_MapDataset(dicts, mapper), sampler=InferenceSampler(len(dicts)), batch_size=1,
                                         num_workers=cfg.DATALOADER.NUM_WORKERS, collate_fn=lambda b: b)
    for batch in loader:
        for d in batch:
            d["image"] = d["image"].to(device, non_blocking=True)
        yield batch
