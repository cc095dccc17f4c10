"""Goldens for divergen_amd/data/factory.py, produced by RUNNING the reference's own scripts (AUTHORING CONTAINER ONLY; nothing of the
reference is copied -- only the synthetic inputs and what the scripts wrote):

  DG/filteration/get_clip_score.py    on a synthetic sample tree, with a stub `clip` (a deterministic scorer shared with the test) and a
                                      stub `torchvision.transforms` (the preprocess is divergen_amd.data.factory.clip_preprocess on both
                                      sides: what is pinned is the sharding, batching, masking, area and ordering logic, not CLIP)
  DG/tools/merge_inst_pool_json.py    on two synthetic pools with prefix replacement

    python tests/golden/make_golden_factory.py      -> tests/golden/factory.json
"""
import json
import os
import runpy
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/DiverGen"
from divergen_amd.data import factory as F  # noqa: E402


def stub_score(images, text):
    """Deterministic stand-in for CLIP's logits_per_text (shared with tests/test_host_factory.py)."""
    return images.double().mean(dim=(1, 2, 3)) * 100.0 + images.double()[:, 0, 3, 5] + len(text)


def run_script(path, argv, stubs):
    old_argv, old_mods, old_to = sys.argv, {k: sys.modules.get(k) for k in stubs}, torch.Tensor.to
    sys.argv = [path] + argv
    sys.modules.update(stubs)

    def to(self, *a, **k):          # the scripts move tensors to 'cuda:0'; there is no GPU in the authoring container
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return old_to(self, *a, **k)
    torch.Tensor.to = to
    try:
        runpy.run_path(path, run_name="__main__")
    finally:
        torch.Tensor.to = old_to
        sys.argv = old_argv
        for k, v in old_mods.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        torch.set_grad_enabled(True)


def clip_stubs():
    clip = types.ModuleType("clip")

    class Model:
        def float(self):
            return self

        def __call__(self, images, text):
            return None, stub_score(images, text._text).float()

    class Tok(torch.Tensor):
        pass

    def tokenize(text):
        t = torch.zeros(1)
        t._text = text
        t.to = lambda *a, **k: t
        return t
    clip.load = lambda *a, **k: (Model(), F.clip_preprocess)
    clip.tokenize = tokenize
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tr.Compose = lambda steps: F.clip_preprocess
    for n in ("Resize", "CenterCrop", "ToTensor", "Normalize"):
        setattr(tr, n, lambda *a, **k: None)
    tr.InterpolationMode = types.SimpleNamespace(BICUBIC=Image.BICUBIC)
    tv.transforms = tr
    return {"clip": clip, "torchvision": tv, "torchvision.transforms": tr}


def gen_clip(out):
    rng = np.random.default_rng(7)
    cats = [{"id": 3, "name": "alarm_clock", "image_count": 5}, {"id": 1, "name": "apple", "image_count": 2},
            {"id": 2, "name": "short", "image_count": 9}]
    images, masks = {}, {}
    for use_mask, bs in ((False, 2), (True, 3)):
        with tempfile.TemporaryDirectory() as d:
            for c in cats:
                n = 5 if c["name"] != "short" else 3          # 'short' has the wrong sample count: skipped by the script
                os.makedirs(os.path.join(d, "in", c["name"]))
                os.makedirs(os.path.join(d, "mask", "sam", c["name"]))
                for k in range(n):
                    key = "%s/%d_%07d.png" % (c["name"], c["id"], k)
                    if key not in images:
                        images[key] = rng.integers(0, 256, (26 + k, 31, 3), dtype=np.uint8)
                        masks[key] = (rng.integers(0, 2, (26 + k, 31), dtype=np.uint8) * 255)
                    Image.fromarray(images[key]).save(os.path.join(d, "in", key))
                    Image.fromarray(masks[key]).save(os.path.join(d, "mask", "sam", key))
            lvis = os.path.join(d, "lvis.json")
            json.dump(cats, open(lvis, "w"))
            os.makedirs(os.path.join(d, "out", "sam"))
            argv = ["--indir", os.path.join(d, "in"), "--outdir", os.path.join(d, "out"), "--n_samples", "5", "--max_batch_size", str(bs),
                    "--in_lvis_json_path", lvis, "--stages", "sd"]
            if use_mask:
                argv += ["--use_mask", "--in_mask_dir", os.path.join(d, "mask"), "--seg_name", "sam"]
            run_script(os.path.join(REF, "filteration", "get_clip_score.py"), argv, clip_stubs())
            res = json.load(open(os.path.join(d, "out", "sam" if use_mask else "", "results.json")))
        out["clip_masked" if use_mask else "clip_plain"] = {"max_batch_size": bs, "results": res}
    out["clip_inputs"] = {"categories": cats, "images": {k: v.tolist() for k, v in images.items()}, "masks": {k: v.tolist() for k, v in masks.items()}}


def gen_merge(out):
    pools = [{"0": ["/a/x/1.png", "/a/x/2.png"], "7": ["/a/y/1.png"]}, {"7": ["/b/y/9.png"], "12": ["/b/z/3.png", "/b/z/4.png"]}]
    with tempfile.TemporaryDirectory() as d:
        argv = []
        for i, p in enumerate(pools):
            json.dump(p, open(os.path.join(d, "p%d.json" % i), "w"))
            argv += ["--inst_pool_path", os.path.join(d, "p%d.json" % i)]
        argv += ["--enable_replace", "--before_prefix", "/a/", "--after_prefix", "/data/A/", "--before_prefix", "/b/", "--after_prefix", "/data/B/",
                 "--out_inst_pool_path", os.path.join(d, "o", "merged.json")]
        run_script(os.path.join(REF, "tools", "merge_inst_pool_json.py"), argv, {"oss2": types.ModuleType("oss2")})
        merged = json.load(open(os.path.join(d, "o", "merged.json")))
    out["merge"] = {"pools": pools, "before": ["/a/", "/b/"], "after": ["/data/A/", "/data/B/"], "merged": merged}


def gen_select(out):
    """DG/filteration/clean_pool_if.py run as __main__ on synthetic results.json trees.  As shipped the script dies at :177
    (`args.enable_split` is read but its parser, :103-115, never defines it): the namespace its parser returns gets
    `enable_split = False` added -- the behaviour of the branch the authors ran -- and nothing else is changed.  cv2 is absent, so the
    128-process pool that crops the selected instances (:214-235 -> work / subwork, cv2.findContours) is replaced by a recorder: the
    golden is the SELECTION the script hands to that pool (category -> 'image|mask' paths), with and without the similarity csv."""
    import argparse
    import csv
    import multiprocessing
    rng = np.random.default_rng(11)
    cats = [{"id": 5, "name": "bee", "image_count": 4}, {"id": 2, "name": "cat", "image_count": 9}, {"id": 9, "name": "empty", "image_count": 1}]
    methods = ["sam", "u2net", "selfreformer"]
    n = {"bee": 6, "cat": 5, "empty": 0}
    res = {}
    for m in methods:
        res[m] = [dict(c, clip_scores=[round(float(v), 3) for v in rng.uniform(15, 30, n[c["name"]])],
                       areas=[round(float(v), 4) for v in rng.uniform(0.0, 1.0, n[c["name"]])]) for c in cats]
        rng.shuffle(res[m])                       # every results.json in its own order: the script sorts by image_count
    keep = [("bee", "5_0000001.png"), ("bee", "5_0000004.png"), ("bee", "5_0000005.png"), ("cat", "2_0000000.png"), ("cat", "2_0000003.png")]
    cases = []
    for use_csv, min_clip, min_area, max_area, tol in ((False, 21.0, 0.05, 0.95, 1.0), (True, 0.0, 0.0, 1.0, 1.0), (False, 40.0, 0.2, 0.8, 6.0)):
        with tempfile.TemporaryDirectory() as d:
            for m in methods:
                os.makedirs(os.path.join(d, "seg", "II", m))
                json.dump(res[m], open(os.path.join(d, "seg", "II", m, "results.json"), "w"))
            argv = ["--input_dir", os.path.join(d, "seg"), "--image_dir", "/img", "--output_file", os.path.join(d, "o", "pool.json"),
                    "--min_clip", str(min_clip), "--min_area", str(min_area), "--max_area", str(max_area), "--tolerance", str(tol),
                    "--seg_method"] + methods + ["--stages", "II"]
            if use_csv:
                with open(os.path.join(d, "keep.csv"), "w", newline="") as f:
                    w = csv.writer(f)
                    w.writerow(["category", "file", "similarity"])
                    for row in keep:
                        w.writerow(list(row) + [0.9])
                argv += ["--filter_image_csv_path", os.path.join(d, "keep.csv")]
            os.makedirs(os.path.join(d, "o"))
            seen = {}

            class Pool:
                def __init__(self, processes=None):
                    pass

                def map(self, fn, parts, chunk=None):
                    for part in parts:
                        seen.update({str(k): [v.replace(d, "$D") for v in vs] for k, vs in part.items() if k != "output"})
                    return [{}]
            old_parse, old_pool, old_ssm = argparse.ArgumentParser.parse_args, multiprocessing.Pool, multiprocessing.set_start_method

            def parse(self, *a, **k):
                ns = old_parse(self, *a, **k)
                ns.enable_split = False
                return ns
            argparse.ArgumentParser.parse_args, multiprocessing.Pool, multiprocessing.set_start_method = parse, Pool, (lambda *a, **k: None)
            try:
                run_script(os.path.join(REF, "filteration", "clean_pool_if.py"), argv, {"cv2": types.ModuleType("cv2")})
            finally:
                argparse.ArgumentParser.parse_args, multiprocessing.Pool, multiprocessing.set_start_method = old_parse, old_pool, old_ssm
        cases.append({"min_clip": min_clip, "min_area": min_area, "max_area": max_area, "tolerance": tol, "csv": keep if use_csv else None,
                      "selected": seen})
    out["select"] = {"methods": methods, "results": res, "cases": cases}


if __name__ == "__main__":
    out = {}
    gen_clip(out)
    gen_merge(out)
    gen_select(out)
    path = os.path.join(ROOT, "tests", "golden", "factory.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")
