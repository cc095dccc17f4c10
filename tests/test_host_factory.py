"""divergen_amd/data/factory.py -- the reference-owned glue of the generation -> mask -> filter factory.
Goldens: tests/golden/factory.json = what the reference's OWN scripts wrote when run on a synthetic tree with stub networks
(tests/golden/make_golden_factory.py); the functions the reference cannot run as shipped are held on hand-worked cases."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from divergen_amd.data import factory as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, "tests", "golden", "factory.json")))


def stub_score(images, text):          # the stand-in for CLIP's logits_per_text used when the golden was made
    return images.double().mean(dim=(1, 2, 3)) * 100.0 + images.double()[:, 0, 3, 5] + len(text)


def _tree(tmp_path):
    for key, arr in G["clip_inputs"]["images"].items():
        for sub, src in (("in", arr), ("mask", G["clip_inputs"]["masks"][key])):
            p = tmp_path / sub / key
            p.parent.mkdir(parents=True, exist_ok=True)
            Image.fromarray(np.array(src, dtype=np.uint8)).save(p)
    return tmp_path


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_clip_scores_match_the_reference_script(tmp_path, masked, world):
    """get_clip_score.py:113-166 (+ :170-204 when sharded): scores and areas of every category in image-index order, for 1, 2 and 3
    ranks (the ranks' shares merged the way the script's all_gather + sort does)."""
    d = _tree(tmp_path)
    gold = G["clip_masked" if masked else "clip_plain"]
    for cat in gold["results"]:
        names = sorted(k for k in G["clip_inputs"]["images"] if k.startswith(cat["name"] + "/"))
        paths = [str(d / "in" / k) for k in names]
        mpaths = [str(d / "mask" / k) for k in names] if masked else None
        if len(paths) != 5:                       # the script skips a category with an unexpected sample count
            assert cat["clip_scores"] == []
            continue
        idx, sc, ar = [], [], []
        for r in range(world):
            i, s, a = F.clip_scores_for_category(paths, cat["name"], stub_score, F.clip_preprocess, gold["max_batch_size"], r, world, mpaths)
            assert i == F.shard_indices(len(paths), r, world)
            idx += i
            sc += s
            ar += a or []
        order = np.argsort(idx, kind="stable")
        np.testing.assert_allclose([sc[j] for j in order], cat["clip_scores"], rtol=1e-6, atol=1e-6)
        if masked:
            np.testing.assert_allclose([ar[j] for j in order], cat["areas"], rtol=0, atol=1e-12)


def test_merge_inst_pools_matches_the_reference_script():
    g = G["merge"]
    assert F.merge_inst_pools(g["pools"], g["before"], g["after"]) == g["merged"]
    assert F.merge_inst_pools(g["pools"])["7"] == ["/a/y/1.png", "/b/y/9.png"]
    with pytest.raises(ValueError):
        F.merge_inst_pools(g["pools"], ["/a/"], ["/x/"])


def test_masked_image_sets_background_to_one():
    img = np.full((2, 3, 3), 200, np.uint8)
    mask = np.array([[255, 128, 129], [0, 200, 10]], np.uint8)
    out, area = F.masked_image_and_area(img, mask)
    assert out[0, 0, 0] == 200 and out[0, 1, 0] == 1 and out[0, 2, 0] == 200 and out[1, 0, 0] == 1    # > 128 keeps, the rest becomes 1
    assert area == 3 / 6


def test_generation_plan_file_numbers_cover_every_sample_exactly_once():
    """txt2img_diffusers_stages_from_txt.py:123-131,213-262: over all ranks every prompt gets n_samples distinct numbers
    offset .. offset + n_samples - 1 (+ n_samples per earlier prompt)."""
    prompts = ["a photo of b", "a photo of a"]
    for n_samples, world, mb in [(8, 2, 3), (8, 4, 2), (6, 1, 4), (4, 2, 2)]:
        seen = {}
        for r in range(world):
            plan = F.generation_plan(prompts, "17", n_samples, mb, r, world, offset=1024)
            total = n_samples // world
            calls = -(-total // mb)
            assert [p for p, _, _ in plan] == sorted(calls * prompts)
            assert sum(n for _, n, _ in plan) == total * len(prompts)
            assert plan[0][1] == (total % mb or mb)                      # the FIRST call of a prompt takes the remainder
            for prompt, n, names in plan:
                assert len(names) == n
                for nm in names:
                    assert nm not in seen, nm
                    seen[nm] = prompt
        nums = sorted(int(k.split("_")[1].split(".")[0]) for k in seen)
        assert nums == list(range(1024, 1024 + n_samples * len(prompts)))
        assert {seen["17_%07d.png" % k] for k in range(1024, 1024 + n_samples)} == {"a photo of a"}
    with pytest.raises(ValueError):
        F.generation_plan(prompts, "1", 7, 2, 0, 2)
    assert F.rank_seed(42, 3) == 45


def test_sam_background_prompting_helpers():
    pts, lab = F.background_corner_points(100, 60, 5)
    assert pts.tolist() == [[5, 5], [0, 54], [94, 5], [94, 54]] and lab.tolist() == [1, 1, 1, 1]
    masks = np.zeros((3, 4, 4), bool)
    masks[2, :2] = True
    out = F.background_mask_from_sam(masks)
    assert out.dtype == np.uint8 and out[:2].max() == 0 and out[2:].min() == 255
    att = np.array([[0.1, 0.9], [0.5, 0.2]])
    assert F.check_point_in_foreground((0, 1), att, 0.5) and not F.check_point_in_foreground((1, 0), att, 0.5)


def test_select_pool_entries_matches_the_reference_script():
    """tests/golden/factory.json 'select': the selection DG/filteration/clean_pool_if.py:157-213 hands to its cropping pool, produced by
    running the script itself (make_golden_factory.py::gen_select: `enable_split = False` supplied to the namespace its parser forgets
    to define, the cv2 cropping pool replaced by a recorder) on three segmentation methods whose results.json files are in different
    orders, with and without the similarity csv, three threshold sets."""
    g = G["select"]
    for case in g["cases"]:
        keep = None
        if case["csv"] is not None:
            keep = {}
            for cat, fn in case["csv"]:
                keep.setdefault(cat, set()).add(fn)
        got = F.select_pool_entries([g["results"][m] for m in g["methods"]], g["methods"], "/img", "$D/seg", "II", min_clip=case["min_clip"],
                                    min_area=case["min_area"], max_area=case["max_area"], tolerance=case["tolerance"], keep_names=keep)
        assert {str(k): v for k, v in got.items()} == case["selected"], case
    assert sum(len(v) for c in g["cases"] for v in c["selected"].values()) >= 15


def test_select_pool_entries_hand_worked():
    """clean_pool_if.py:157-213 on two segmentation methods: per image the method with the higher CLIP score; bar = min(min_clip,
    best score of the category - tolerance); area window; csv filter."""
    a = [{"id": 2, "name": "cat", "image_count": 9, "clip_scores": [20.0, 25.0, 10.0, 24.0], "areas": [0.5, 0.01, 0.5, 0.99]},
         {"id": 1, "name": "ant", "image_count": 3, "clip_scores": [], "areas": []}]
    b = [{"id": 1, "name": "ant", "image_count": 3, "clip_scores": [], "areas": []},
         {"id": 2, "name": "cat", "image_count": 9, "clip_scores": [22.0, 21.0, 12.0, 23.0], "areas": [0.4, 0.3, 0.6, 0.5]}]
    out = F.select_pool_entries([a, b], ["sam", "u2"], "/img", "/seg", "II", min_clip=21.0, min_area=0.05, max_area=0.95, tolerance=1.0)
    # bar = min(21, 25 - 1) = 21.  k=0: best u2 (22 >= 21, area .4) keep; k=1: best sam (25) but area .01 < min -> drop (no fall-back to
    # the other method, as in the reference); k=2: best u2 12 < bar -> drop; k=3: best sam 24, area .99 > max -> drop
    assert out == {1: ["/img/II/cat/2_0000000.png|/seg/II/u2/cat/2_0000000.png"]}
    out = F.select_pool_entries([a, b], ["sam", "u2"], "/img", "/seg", "II", min_clip=0.0, tolerance=1.0,
                                keep_names={"cat": {"2_0000002.png", "2_0000003.png"}})
    assert sorted(out[1]) == ["/img/II/cat/2_0000002.png|/seg/II/u2/cat/2_0000002.png", "/img/II/cat/2_0000003.png|/seg/II/sam/cat/2_0000003.png"]
    with pytest.raises(ValueError):
        F.select_pool_entries([a, [dict(b[1], id=5), b[0]]], ["sam", "u2"], "/img", "/seg", "II")


def test_crop_instance_keeps_the_largest_component_with_holes_filled():
    rgba = np.zeros((12, 14, 4), np.uint8)
    rgba[..., :3] = 77
    rgba[2:9, 3:10, 3] = 255          # 7 x 7 square ...
    rgba[4:6, 5:7, 3] = 0             # ... with a hole (filled by fillPoly)
    rgba[10:12, 0:2, 3] = 255         # a smaller component: dropped
    rgba[0, 13, 3] = 100              # alpha <= 128: background
    out = F.crop_instance(rgba)
    assert out.shape == (7, 7, 4)
    assert out[..., 3].min() == 0 and out[0, 0, 3] == 255      # alpha multiplied by the 0/1 component mask: the hole keeps alpha 0
    assert (out[..., 3] > 0).sum() == 49 - 4
    seg = F.largest_component_filled((rgba[..., 3:] > 128).astype("uint8"))
    assert seg.shape == (12, 14, 1) and seg.sum() == 49 and seg[10:, :2].sum() == 0
    assert F.crop_instance(np.zeros((5, 5, 4), np.uint8)) is None
    one = np.zeros((5, 5, 4), np.uint8)
    one[2, 2, 3] = 255
    assert F.crop_instance(one) is None                        # y_max <= y_min: a single pixel is not an instance
    m = np.zeros((5, 5), np.uint8)
    m[1:4, 1:4] = 255
    assert F.crop_instance(np.zeros((5, 5, 4), np.uint8), mask=m).shape == (3, 3, 4)


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    idx = F.shard_indices(7, rank, world)                       # 4 images on rank 0, 3 on rank 1: unequal shares
    sc, ar = [10.0 * i + 0.5 for i in idx], [i / 8.0 for i in idx]
    q.put((rank, F.gather_by_index(idx, sc, ar)))
    dist.destroy_process_group()


def test_gather_by_index_world2_gloo():
    """get_clip_score.py:170-204 over two ranks: every rank ends up with all scores / areas in image-index order."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    for r in range(2):
        sc, ar = got[r]
        assert sc == [10.0 * i + 0.5 for i in range(7)] and ar == [i / 8.0 for i in range(7)]
    assert F.gather_by_index([2, 0, 1], [5.0, 3.0, 4.0]) == [[3.0, 4.0, 5.0]]
