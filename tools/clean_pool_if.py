"""CLI of the pool cleaner (DG/filteration/clean_pool_if.py's arguments): CLIP-score / area selection over the segmentation methods'
results.json files, largest-component crop of every kept instance, pool json out.  Logic: divergen_amd/data/factory.py.
    python tools/clean_pool_if.py --input_dir SEG --image_dir IMG --output_file OUT/pool.json --seg_method sam u2 --stages II --min_clip 21"""
import argparse
import csv
import json
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from divergen_amd.data import factory as F  # noqa: E402

if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir")
    p.add_argument("--image_dir")
    p.add_argument("--output_file")
    p.add_argument("--filter_image_csv_path", default=None)
    p.add_argument("--min_clip", type=float, default=0)
    p.add_argument("--min_area", type=float, default=0.0)
    p.add_argument("--max_area", type=float, default=1.0)
    p.add_argument("--tolerance", type=float, default=1)
    p.add_argument("--seg_method", nargs="+")
    p.add_argument("--stages", nargs="+", default=["I", "II"])
    a = p.parse_args()
    keep = None
    if a.filter_image_csv_path is not None:          # first column category, second file name (data/filtration.py writes it)
        keep = {}
        with open(a.filter_image_csv_path) as f:
            rows = csv.reader(f)
            next(rows)
            for row in rows:
                keep.setdefault(row[0], set()).add(row[1])
    out_dir = os.path.dirname(a.output_file)
    result = {}
    for stage in a.stages:
        res = [json.load(open(os.path.join(a.input_dir, stage, m, "results.json"))) for m in a.seg_method]
        picked = F.select_pool_entries(res, a.seg_method, a.image_dir, a.input_dir, stage, a.min_clip, a.min_area, a.max_area, a.tolerance, keep)
        for cid, entries in picked.items():
            os.makedirs(os.path.join(out_dir, "images", str(cid)), exist_ok=True)
            done = []
            for c, e in enumerate(entries):
                img_path, mask_path = e.split("|")
                try:
                    inst = F.crop_instance(np.array(Image.open(img_path).convert("RGBA")), np.array(Image.open(mask_path)))
                except Exception:
                    inst = None
                if inst is None:
                    continue
                dst = os.path.join(out_dir, "images", str(cid), "%d.png" % c)
                Image.fromarray(inst).save(dst)
                done.append("*" + dst)
            result[cid] = done
    with open(a.output_file, "w") as f:
        json.dump(result, f)
    print("kept", {k: len(v) for k, v in result.items()})
