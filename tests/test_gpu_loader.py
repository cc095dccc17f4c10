"""GPU: the product's REAL data path, end to end.
(a) InstPool (worker half + compositor kernel) on the PNG fixtures against the reference's own get_mix_result outputs
    (tests/golden/pool_draws.npz) and against oracle/compositor.py fed the same draws -- directly and through the shard store;
(b) build_detection_train_loader with worker processes: what comes out is what one process computes, composited one batch ahead;
(c) train_net.do_train on a generated LVIS-format split + PNG pool with DATALOADER.NUM_WORKERS 4: loss keys, metrics.json,
    checkpoint file set and contents, --resume.
Reference: DG/divergen/data/custom_build_copypaste_mapper.py:856-958,213-296,359-456,488-566; DG/train_net.py:164-239,248-304."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FREQ = os.path.join(ROOT, "configs", "metadata", "ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json")


@pytest.fixture()
def in_golden_dir(monkeypatch):
    monkeypatch.chdir(GOLD)


def _case_input(z, ci, H, W):
    from divergen_amd.structures import BitMasks, Boxes, Instances
    inst = Instances((H, W), gt_boxes=Boxes(torch.from_numpy(z["c%d_boxes" % ci])), gt_classes=torch.from_numpy(z["c%d_labels" % ci]),
                     gt_masks=BitMasks(torch.from_numpy(z["c%d_masks" % ci])))
    return {"image": torch.from_numpy(z["c%d_image" % ci]), "instances": inst, "file_name": "case%d" % ci}


@pytest.mark.parametrize("through_shards", [False, True])
def test_instpool_end_to_end_equals_reference_mix_result(in_golden_dir, tmp_path, through_shards):
    from oracle import compositor as OK
    from tests.test_host_pool_draws import cases, make_pool
    z = np.load(os.path.join(GOLD, "pool_draws.npz"))
    loader = None
    if through_shards:          # INPUT.INST_POOL_SHARDS: the same keys decoded once into mmap-ed shard files
        from divergen_amd.data import pool_store as PS
        keys = [str(k) for k in np.load(os.path.join(GOLD, "pool_decode.npz"))["keys"]]
        r = PS.build_shards({"0": keys}, str(tmp_path / "shards"))
        assert r["records"] == len(keys)
        loader = PS.PoolStore(str(tmp_path / "shards")).loader
    ip = make_pool(z, loader)
    H, W = (int(v) for v in z["hw"])
    n_pasted = 0
    for ci in cases(z):
        np.random.seed(int(z["c%d_seed" % ci]))
        out = ip(_case_input(z, ci, H, W))              # prepare (CPU) + composite (dgx_copy_paste)
        o = out["instances"]
        assert out["image"].is_cuda and (out["height"], out["width"]) == (H, W)
        assert np.array_equal(out["image"].cpu().numpy(), z["c%d_out_image" % ci]), ci
        assert np.array_equal(o.gt_boxes.tensor.cpu().numpy(), z["c%d_out_boxes" % ci])
        assert np.array_equal(o.gt_classes.cpu().numpy(), z["c%d_out_labels" % ci])
        assert np.array_equal(o.gt_masks.tensor.cpu().numpy().astype(np.uint8), z["c%d_out_masks" % ci])
        assert np.array_equal(o.instance_source.cpu().numpy(), z["c%d_out_source" % ci])
        n_pasted += int(o.instance_source.sum())
        # the same draws through the oracle compositor
        np.random.seed(int(z["c%d_seed" % ci]))
        pastes, _ = ip.draw((H, W))
        ref = OK.composite(z["c%d_image" % ci], z["c%d_masks" % ci], z["c%d_boxes" % ci], z["c%d_labels" % ci], pastes)
        assert np.array_equal(out["image"].cpu().numpy(), ref["image"]) and np.array_equal(o.gt_masks.tensor.cpu().numpy().astype(np.uint8), ref["masks"])
        assert out["paste_labels"] == [p[3] for p in pastes]
    assert n_pasted >= 10


def _mini_cfg(tmp_path, size, workers, extra=()):
    from divergen_amd.config import add_bsgal_config, add_centernet_config, add_divergen_config, get_cfg
    from divergen_amd.data.synthetic import write_mini_lvis
    info = write_mini_lvis(str(tmp_path / "data"), n_images=12, image_hw=(120, 160), n_obj=5, n_pool=10, pool_px=(40, 90), seed=5, poly_vertices=16)
    cfg = get_cfg()
    add_centernet_config(cfg), add_divergen_config(cfg), add_bsgal_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "DiverGen_swinL.yaml"))
    cfg.merge_from_list(["MODEL.SWIN.SIZE", "T", "MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", FREQ, "INPUT.TRAIN_SIZE", size,
                         "INPUT.INST_POOL_PATH", info["pool_json"], "INPUT.MEAN_STD2_PATH", os.path.join(ROOT, "configs", "metadata", "area_mean_std2.json"),
                         "INPUT.RANDOM_SCALE_MIN", 0.3, "INPUT.RANDOM_SCALE_MAX", 1.2, "DATALOADER.NUM_WORKERS", workers,
                         "SOLVER.IMS_PER_BATCH", 2, "MODEL.WEIGHTS", "", "OUTPUT_DIR", str(tmp_path / "out"), "MODEL.DEVICE", "cuda"] + list(extra))
    return cfg, info


def test_loader_with_workers_equals_one_process(tmp_path, monkeypatch):
    """4 worker processes + pin thread + side-stream compositor against the same mapper run inline with the workers' seeds:
    every batch bit-identical (a batch comes from ONE worker, seeded base + worker_id, visiting samples in sampler order)."""
    from divergen_amd.data import build as B
    cfg, info = _mini_cfg(tmp_path, 128, 4)
    monkeypatch.setenv("DETECTRON2_DATASETS", info["root"])
    seed, per_gpu, nb = 3, 2, 6
    it = B.build_detection_train_loader(cfg, per_gpu, "cuda", seed)
    got = [next(it) for _ in range(nb)]
    torch.cuda.synchronize()
    # inline: worker w handles batches w, w + 4, ... with np.random seeded (seed * 1009 + w)
    dicts = B.get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
    mapper = B.CopyPasteMapper(B.DatasetMapper(cfg, True), cfg)
    mapper.set_dataset(dicts)
    from divergen_amd.data.samplers import RepeatFactorTrainingSampler
    rf = RepeatFactorTrainingSampler.repeat_factors_from_category_frequency(dicts, cfg.DATALOADER.REPEAT_THRESHOLD)
    import itertools
    idx = list(itertools.islice(iter(RepeatFactorTrainingSampler(rf, seed=seed)), nb * per_gpu))
    pasted = 0
    for w in range(4):
        np.random.seed((seed * 1009 + w) % (2 ** 31))
        for b in range(w, nb, 4):
            for j in range(per_gpu):
                want = mapper.finish(mapper(dicts[idx[b * per_gpu + j]]), "cuda")
                have = got[b][j]
                assert have["file_name"] == want["file_name"]
                assert torch.equal(have["image"], want["image"]), (b, j)
                hi, wi = have["instances"], want["instances"]
                assert torch.equal(hi.gt_boxes.tensor, wi.gt_boxes.tensor) and torch.equal(hi.gt_classes, wi.gt_classes)
                assert torch.equal(hi.gt_masks.tensor, wi.gt_masks.tensor) and torch.equal(hi.instance_source, wi.instance_source)
                assert "paste_pack" not in have and have["paste_filename_list"] == want["paste_filename_list"]
                pasted += int(hi.instance_source.sum())
    assert pasted > 0 and it.side is not None


def test_do_train_through_the_real_loader_checkpoints_and_resumes(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import train_net
    from divergen_amd.modeling import build_model
    cfg, info = _mini_cfg(tmp_path, 128, 4, ["SOLVER.MAX_ITER", 30, "SOLVER.CHECKPOINT_PERIOD", 12, "SOLVER.WARMUP_ITERS", 5, "SEED", 7])
    monkeypatch.setenv("DETECTRON2_DATASETS", info["root"])
    os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)          # train_net.setup() does this for the CLI
    torch.manual_seed(7)
    opt = train_net.do_train(cfg, build_model(cfg))
    out = str(tmp_path / "out")
    files = sorted(f for f in os.listdir(out) if f.endswith(".pth") or f == "last_checkpoint")
    assert files == ["last_checkpoint", "model_0000011.pth", "model_0000023.pth", "model_final.pth"], files
    assert open(os.path.join(out, "last_checkpoint")).read().strip() == "model_final.pth"
    ck = torch.load(os.path.join(out, "model_final.pth"), map_location="cpu", weights_only=False)
    assert {"model", "optimizer", "scheduler", "model_ema", "iteration"} <= set(ck) and ck["iteration"] == 30      # DG/train_net.py:257 counts from 1; the final save fires at max_iter - 1 and max_iter
    rows = [json.loads(line) for line in open(os.path.join(out, "metrics.json"))]
    keys = {"loss_cls_stage0", "loss_box_reg_stage0", "loss_cls_stage1", "loss_box_reg_stage1", "loss_cls_stage2", "loss_box_reg_stage2",
            "loss_mask", "loss_centernet_loc", "loss_centernet_agn_pos", "loss_centernet_agn_neg", "total_loss", "lr", "data_time", "time", "iteration"}
    # (data_time is stored before storage.step(), i.e. under the previous iteration, as in DG/train_net.py:254-259: its own row)
    seen = set().union(*[set(r) for r in rows])
    assert len(rows) >= 2 and keys <= seen and keys - {"data_time"} <= set(rows[-1]), (sorted(seen), sorted(rows[-1]))
    assert all(np.isfinite(r["total_loss"]) for r in rows if "total_loss" in r)
    p_end = opt.arena.p.clone()
    # --resume: continues after iteration 30 of a longer schedule, with the saved optimizer / scheduler / EMA
    cfg2 = cfg.clone()
    cfg2.defrost()
    cfg2.merge_from_list(["SOLVER.MAX_ITER", 36])
    torch.manual_seed(8)
    opt2 = train_net.do_train(cfg2, build_model(cfg2), resume=True)
    ck2 = torch.load(os.path.join(out, "model_final.pth"), map_location="cpu", weights_only=False)
    assert ck2["iteration"] == 36
    assert float((opt2.arena.p - p_end).abs().max()) > 0                     # it trained on ...
    rows2 = [json.loads(line) for line in open(os.path.join(out, "metrics.json"))]
    assert rows2[-1]["iteration"] > rows[-1]["iteration"]


def test_generation_factory_to_training_step_chain(tmp_path, monkeypatch):
    """BASELINE config #5 end to end with stub networks: 'generated' images + two stub segmenters -> CLIP-style
    scoring of the masked samples (factory.clip_scores_for_category, a deterministic stand-in for the network) -> best-mask selection
    with score / area bars (factory.select_pool_entries = DG/filteration/clean_pool_if.py:157-213) -> largest-component crop written as
    pre-processed '*' RGBA files (factory.crop_instance = clean_pool_if.py:47-84) -> merge with a second pool
    (factory.merge_inst_pools = DG/tools/merge_inst_pool_json.py) -> shard store (data/pool_store.py) -> INPUT.INST_POOL_SHARDS ->
    InstPool in loader workers -> GPU compositor -> one optimizer step of the registry-built model.  Asserted: every stage hands the
    next what it expects, pasted instances in the batches carry exactly the factory's categories, the step's losses are finite."""
    sys.path.insert(0, ROOT)
    from PIL import Image
    from divergen_amd.data import build as B
    from divergen_amd.data import factory as F
    from divergen_amd.data import pool_store as PS
    from divergen_amd.engine import total_loss
    from divergen_amd.modeling import build_model
    from divergen_amd.solver import build_optimizer
    from divergen_amd.utils.events import EventStorage
    rng = np.random.default_rng(17)
    cats = [{"id": 4, "name": "alarm_clock", "image_count": 3}, {"id": 11, "name": "apple", "image_count": 7}]
    methods, n_samples, stage = ["sam", "u2net"], 4, "II"
    img_dir, seg_dir = tmp_path / "gen", tmp_path / "seg"
    yy, xx = np.mgrid[0:96, 0:80]
    for c in cats:
        (img_dir / stage / c["name"]).mkdir(parents=True)
        for k in range(n_samples):
            Image.fromarray(rng.integers(0, 256, (96, 80, 3), dtype=np.uint8)).save(img_dir / stage / c["name"] / ("%d_%07d.png" % (c["id"], k)))
            for mi, m in enumerate(methods):          # two "segmenters": an ellipse each, the second one smaller and with a stray blob
                (seg_dir / stage / m / c["name"]).mkdir(parents=True, exist_ok=True)
                r = 30 - 12 * mi + k
                mask = ((((xx - 40) / r) ** 2 + ((yy - 48) / (r + 6)) ** 2) <= 1).astype(np.uint8) * 255
                if mi == 1:
                    mask[2:6, 2:6] = 255
                Image.fromarray(mask).save(seg_dir / stage / m / c["name"] / ("%d_%07d.png" % (c["id"], k)))

    def score(images, text):                          # the stand-in for CLIP: deterministic in the masked pixels and the prompt
        return images.double().mean(dim=(1, 2, 3)) * 10.0 + 20.0 + len(text) * 0.01
    results = {m: [] for m in methods}
    for m in methods:
        for c in cats:
            paths = [str(img_dir / stage / c["name"] / ("%d_%07d.png" % (c["id"], k))) for k in range(n_samples)]
            mpaths = [str(seg_dir / stage / m / c["name"] / ("%d_%07d.png" % (c["id"], k))) for k in range(n_samples)]
            idx, sc, ar = F.clip_scores_for_category(paths, c["name"], score, F.clip_preprocess, 2, 0, 1, mpaths)
            assert idx == list(range(n_samples)) and len(sc) == len(ar) == n_samples
            results[m].append(dict(c, clip_scores=sc, areas=ar))
    selected = F.select_pool_entries([results[m] for m in methods], methods, str(img_dir), str(seg_dir), stage, min_clip=0.0, min_area=0.02,
                                     max_area=0.9, tolerance=50.0)
    assert set(selected) == {3, 10} and all(len(v) == n_samples for v in selected.values())
    pool_a = {}
    (tmp_path / "crops").mkdir()
    for cid, entries in selected.items():
        for j, e in enumerate(entries):
            ip, mp_ = e.split("|")
            crop = F.crop_instance(np.array(Image.open(ip).convert("RGBA")), np.array(Image.open(mp_)))
            assert crop is not None and crop.shape[2] == 4 and crop.shape[0] < 96        # cropped to the (largest) component
            out = tmp_path / "crops" / ("%d_%d.png" % (cid, j))
            Image.fromarray(crop).save(out)
            pool_a.setdefault(str(cid), []).append("*" + str(out))
    from divergen_amd.data.synthetic import write_mini_lvis
    info = write_mini_lvis(str(tmp_path / "data"), n_images=8, image_hw=(120, 160), n_obj=4, n_pool=6, pool_px=(40, 80), seed=9, poly_vertices=12)
    pool_b = json.load(open(info["pool_json"]))
    merged = F.merge_inst_pools([pool_a, pool_b])
    assert set(pool_a) <= set(merged) and sum(len(v) for v in merged.values()) == 2 * n_samples + 6
    json.dump(merged, open(tmp_path / "merged.json", "w"))
    r = PS.build_shards(merged, str(tmp_path / "shards"))
    assert r["records"] == 2 * n_samples + 6 and r["failed"] == []
    cfg, _ = _mini_cfg(tmp_path / "cfgdata", 128, 2, ["INPUT.INST_POOL_PATH", str(tmp_path / "merged.json"), "INPUT.INST_POOL_SHARDS", str(tmp_path / "shards"),
                                                      "INPUT.INST_POOL_MAX_SAMPLES", 12])
    monkeypatch.setenv("DETECTRON2_DATASETS", info["root"])
    loader = B.build_detection_train_loader(cfg, 2, "cuda", 5)
    torch.manual_seed(3)
    model = build_model(cfg).train()
    model.early_proposal_backward = True
    opt = build_optimizer(cfg, model)
    factory_labels, pasted = {3, 10}, set()
    with EventStorage(0):
        for it in range(4):
            batch = next(loader)
            for d in batch:
                src = d["instances"].instance_source.bool()
                pasted |= set(d["instances"].gt_classes[src].tolist())
                assert set(d["paste_labels"]) <= {int(k) for k in merged}
            opt.zero_grad()
            losses = model(batch)
            total = total_loss(losses)
            total.backward()
            opt.step()
            assert bool(torch.isfinite(total)), losses
    torch.cuda.synchronize()
    assert pasted & factory_labels, (pasted, "no instance of the factory's categories was pasted in 8 images")
