"""CPU tests: samplers vs the reference's known-answer tests, checkpointer round trip + key matching."""
import math
import os

import pytest

import torch


def test_repeat_factor_sampler_kat_d2t():
    # D2T/data/test_sampler.py:76-90
    from divergen_amd.data.samplers import RepeatFactorTrainingSampler
    dataset_dicts = [
        {"annotations": [{"category_id": 0}, {"category_id": 1}]},
        {"annotations": [{"category_id": 0}]},
        {"annotations": []},
    ]
    rep = RepeatFactorTrainingSampler.repeat_factors_from_category_frequency(dataset_dicts, 0.5)
    assert torch.allclose(rep, torch.tensor([math.sqrt(3 / 2), 1.0, 1.0]))


def test_training_sampler_seeded_and_sharded():
    # D2T/data/test_sampler.py:63-74: seeded stream; ranks take a strided slice of ONE stream
    from divergen_amd.data.samplers import TrainingSampler
    import itertools
    full = list(itertools.islice(TrainingSampler(5, True, seed=42, rank=0, world_size=1), 10))
    assert sorted(full[:5]) == list(range(5)) and sorted(full[5:]) == list(range(5))
    r0 = list(itertools.islice(TrainingSampler(5, True, seed=42, rank=0, world_size=2), 5))
    r1 = list(itertools.islice(TrainingSampler(5, True, seed=42, rank=1, world_size=2), 5))
    assert r0 == full[0::2] and r1 == full[1::2]


def test_sampler_streams_equal_reference_golden():
    """tests/golden/samplers.npz: streams of the reference's own sampler classes (make_golden.py::gen_samplers) -- seeded,
    sharded, with and without shuffling; the inference shards as (start, length)."""
    import itertools
    import numpy as np
    from divergen_amd.data.samplers import InferenceSampler, RepeatFactorTrainingSampler, TrainingSampler
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "samplers.npz"))
    rf = torch.from_numpy(z["repeat_factors"])
    for ci, (seed, rank, world, shuffle) in enumerate(z["cases"].tolist()):
        s = RepeatFactorTrainingSampler(rf, shuffle=bool(shuffle), seed=seed, rank=rank, world_size=world)
        assert list(itertools.islice(iter(s), 300)) == z["rf_%d" % ci].tolist()
        s = TrainingSampler(23, bool(shuffle), seed=seed, rank=rank, world_size=world)
        assert list(itertools.islice(iter(s), 100)) == z["tr_%d" % ci].tolist()
    for tot, w, r, start, n in z["inference_shards"].tolist():
        got = InferenceSampler._get_local_indices(tot, w, r)
        assert len(got) == n and (n == 0 or got.start == start)


def test_inference_sampler_shards():
    # D2T/data/test_sampler.py:94-111
    from divergen_amd.data.samplers import InferenceSampler
    assert list(InferenceSampler._get_local_indices(100, 4, 0)) == list(range(25))
    sizes = [len(InferenceSampler._get_local_indices(10, 3, r)) for r in range(3)]
    assert sizes == [4, 3, 3]
    allidx = sum([list(InferenceSampler._get_local_indices(10, 3, r)) for r in range(3)], [])
    assert allidx == list(range(10))


def test_checkpointer_roundtrip_and_suffix_matching(tmp_path):
    from divergen_amd.checkpoint import DetectionCheckpointer, PeriodicCheckpointer, _match_keys
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    ck = DetectionCheckpointer(net, str(tmp_path), save_to_disk=True)
    per = PeriodicCheckpointer(ck, 2, max_iter=4)
    for it in range(4):
        per.step(it)
    files = sorted(os.listdir(tmp_path))
    assert files == ["last_checkpoint", "model_0000001.pth", "model_0000003.pth", "model_final.pth"]
    net2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    extra = DetectionCheckpointer(net2, str(tmp_path)).resume_or_load("", resume=True)
    assert extra["iteration"] == 3
    for a, b in zip(net.parameters(), net2.parameters()):
        assert torch.equal(a, b)
    m = _match_keys(["backbone.bottom_up.layers.0.blocks.0.attn.qkv.weight", "backbone.fpn_lateral3.weight"],
                    ["layers.0.blocks.0.attn.qkv.weight", "fpn_lateral3.weight", "head.weight"])
    assert m == {"layers.0.blocks.0.attn.qkv.weight": "backbone.bottom_up.layers.0.blocks.0.attn.qkv.weight",
                 "fpn_lateral3.weight": "backbone.fpn_lateral3.weight"}
    # a short checkpoint key ('norm.weight') listed FIRST must not claim a model key whose true owner is a longer suffix,
    # whatever the order of either list (the earlier greedy matcher depended on the checkpoint's key order)
    mk = ["backbone.bottom_up.patch_embed.norm.weight", "backbone.bottom_up.norm.weight"]
    for ckeys in (["norm.weight", "patch_embed.norm.weight"], ["patch_embed.norm.weight", "norm.weight"]):
        for mkeys in (mk, mk[::-1]):
            assert _match_keys(mkeys, ckeys) == {"patch_embed.norm.weight": mk[0], "norm.weight": mk[1]}
    with pytest.raises(ValueError):          # one checkpoint key, two owners: ambiguous, as in the reference
        _match_keys(["a.norm.weight", "b.norm.weight"], ["norm.weight"])


# ------------------------------------------------------------------ the real data path (divergen_amd/data/build.py)
def test_efficientdet_resize_crop_vs_reference_golden(golden):
    """EfficientDetResizeCrop + its transform against outputs of the reference's own files (tests/golden/make_golden.py
    gen_augment): same np.random stream -> same parameters, byte-identical image / segmentation, same coordinates."""
    import numpy as np
    from divergen_amd.data.build import EfficientDetResizeCrop
    g = golden("augment")
    ci = 0
    while "c%d_img" % ci in g.files:
        size, s0, s1, seed = g["c%d_cfg" % ci]
        np.random.seed(int(seed))
        t = EfficientDetResizeCrop(int(size), (float(s0), float(s1))).get_transform(g["c%d_img" % ci])
        got = np.array([t.scaled_h, t.scaled_w, t.offset_y, t.offset_x, t.img_scale, t.target_size[0], t.target_size[1]])
        assert np.array_equal(got, g["c%d_params" % ci]), (ci, got, g["c%d_params" % ci])
        assert np.array_equal(t.apply_image(g["c%d_img" % ci]), g["c%d_out" % ci])
        assert np.array_equal(t.apply_image(g["c%d_seg" % ci], nearest=True), g["c%d_seg_out" % ci])
        assert np.array_equal(t.apply_coords(g["c%d_pts" % ci].copy()), g["c%d_pts_out" % ci])
        ci += 1
    assert ci >= 5


def test_polygon_rasterisation_known_answers():
    """COCO polygon fill (pycocotools rleFrPoly restated): an axis-aligned box [x0, x1) x [y0, y1) fills exactly those pixels;
    a right triangle keeps the pixels whose centres lie inside; two polygons of one instance are OR-ed."""
    import numpy as np
    from divergen_amd.data.build import polygons_to_bitmask
    m = polygons_to_bitmask([[1, 1, 4, 1, 4, 3, 1, 3]], 5, 6)
    want = np.zeros((5, 6), bool)
    want[1:3, 1:4] = True
    assert np.array_equal(m, want)
    tri = polygons_to_bitmask([[0, 0, 5, 0, 0, 4]], 5, 6).astype(int)
    assert tri.tolist() == [[1, 1, 1, 1, 0, 0], [1, 1, 1, 0, 0, 0], [1, 1, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0]]
    both = polygons_to_bitmask([[1, 1, 4, 1, 4, 3, 1, 3], [4, 3, 6, 3, 6, 5, 4, 5]], 5, 6)
    want[3:5, 4:6] = True
    assert np.array_equal(both, want)
    assert not polygons_to_bitmask([[10, 10, 12, 10, 12, 12]], 5, 6).any()          # fully outside the canvas


def _tiny_lvis(tmp_path):
    import json
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(3)
    root = tmp_path / "datasets"
    (root / "coco" / "train2017").mkdir(parents=True)
    (root / "lvis").mkdir()
    images, anns = [], []
    for i in range(6):
        h, w = 60 + 7 * i, 80 + 5 * i
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(root / "coco" / "train2017" / ("%012d.jpg" % (i + 1)))
        images.append({"id": i + 1, "height": h, "width": w, "coco_url": "http://images.cocodataset.org/train2017/%012d.jpg" % (i + 1),
                       "neg_category_ids": [2], "not_exhaustive_category_ids": []})
        for k in range(0 if i == 5 else 2):
            x, y = 5 + 10 * k, 4 + 8 * k
            anns.append({"id": len(anns) + 1, "image_id": i + 1, "category_id": 1 + (i + k) % 3, "bbox": [x, y, 30, 20],
                         "segmentation": [[x, y, x + 30, y, x + 30, y + 20, x, y + 20]], "area": 600.0})
    cats = [{"id": c, "name": "c%d" % c, "frequency": "f"} for c in (1, 2, 3)]
    with open(root / "lvis" / "lvis_v1_train.json", "w") as f:
        json.dump({"images": images, "annotations": anns, "categories": cats}, f)
    return str(root)


def test_lvis_loader_end_to_end_on_cpu(tmp_path, monkeypatch):
    """json -> dataset dicts -> mapper (resize-crop, flip, bitmasks) -> repeat-factor sampler -> batches, no copy-paste."""
    import os
    import numpy as np
    from divergen_amd.config import get_cfg
    from divergen_amd.data import build as B
    monkeypatch.setenv("DETECTRON2_DATASETS", _tiny_lvis(tmp_path))
    dicts = B.get_detection_dataset_dicts(["lvis_v1_train"])
    assert len(dicts) == 5                                            # the image without annotations is dropped
    d0 = dicts[0]
    assert d0["file_name"].endswith("coco/train2017/000000000001.jpg") and d0["neg_category_ids"] == [1]
    assert {a["category_id"] for a in d0["annotations"]} <= {0, 1, 2} and d0["annotations"][0]["bbox_mode"] == "XYWH_ABS"
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "DiverGen_swinL.yaml"))
    cfg.merge_from_list(["INPUT.TRAIN_SIZE", 64, "INPUT.INST_POOL", False, "INPUT.USE_COPY_METHOD", "none", "DATALOADER.NUM_WORKERS", 0,
                         "DATALOADER.SAMPLER_TRAIN", "RepeatFactorTrainingSampler", "DATALOADER.REPEAT_THRESHOLD", 0.5])
    it = B.build_detection_train_loader(cfg, 2, "cpu", seed=7)
    for _ in range(3):
        batch = next(it)
        assert len(batch) == 2
        for d in batch:
            img, inst = d["image"], d["instances"]
            assert img.dtype.is_floating_point is False and img.shape[0] == 3 and max(img.shape[1:]) <= 64
            assert inst.image_size == tuple(img.shape[1:]) and inst.gt_masks.tensor.shape[1:] == img.shape[1:]
            assert len(inst) == len(inst.gt_classes) == inst.gt_masks.tensor.shape[0]
            if len(inst):
                b = inst.gt_boxes.tensor
                assert float(b.min()) >= 0 and float(b[:, 2].max()) <= img.shape[2] and float(b[:, 3].max()) <= img.shape[1]
                # a rasterised axis-aligned box polygon has the area of its (clipped) box up to the pixel-centre rule
                area_m = inst.gt_masks.tensor.flatten(1).sum(1).float()
                area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
                assert bool(((area_m - area_b).abs() <= 0.25 * area_b + 8).all())
    with __import__("pytest").raises(FileNotFoundError):
        B.load_lvis_json(str(tmp_path / "missing.json"), "x")


def test_train_loader_default_seed_and_missing_dataset(monkeypatch, tmp_path):
    """SEED -1 (the reference default) must not crash the synthetic loader, and a real dataset name without its
    files must raise instead of training on noise."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import train_net
    from divergen_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "baseline_swinL.yaml"))
    assert cfg.SEED < 0
    cfg.merge_from_list(["DATASETS.TRAIN", ("synthetic",), "INPUT.TRAIN_SIZE", 64, "SOLVER.IMS_PER_BATCH", 2])
    batch = next(train_net.build_train_loader(cfg, None))
    assert len(batch) == 2 and batch[0]["image"].shape == (3, 64, 64)
    cfg.merge_from_list(["DATASETS.TRAIN", ("lvis_v1_train",), "INPUT.INST_POOL", False, "INPUT.USE_COPY_METHOD", "none"])
    monkeypatch.setenv("DETECTRON2_DATASETS", str(tmp_path))
    with __import__("pytest").raises(FileNotFoundError):
        next(train_net.build_train_loader(cfg, None))
