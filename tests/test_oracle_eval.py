"""Evaluation row: the matching / accumulation arithmetic of divergen_amd/evaluation/lvis_eval.py against the
reference's own C++ evaluator, D2/layers/csrc/cocoeval/cocoeval.cpp (what COCOeval_opt calls, D2/evaluation/fast_eval_api.py:88,109),
compiled where it lies into oracle/_ref/dgref.so (oracle/build.py).

lvis-api itself is not vendored by the reference ("parity unpinned" for the LVIS-only rules: federated filtering, not-exhaustive
categories, the r / c / f split -- those stay on the hand-worked cases of tests/test_host_eval.py).  What LVISEval shares with
COCOeval -- greedy matching of score-sorted detections per (image, category, area range, IoU threshold) with ignored ground truth
last, the cross-image precision / recall curve made monotone from the right and sampled at 101 recall points -- is identical in
both algorithms when no ground truth is a crowd and no image holds more than max_dets detections, so on such a COCO-style set
the two must produce THE SAME precision and recall tensors.  IoUs are handed to both sides (box_iou_xywh), so this pins
matching and accumulation, not the IoU routine (pinned by known answers in test_host_eval.py).
"""
import types

import numpy as np
import pytest

from divergen_amd.evaluation import lvis_eval as LE


def _dataset(seed, n_img=14, n_cat=5, crowd_free=True):
    rng = np.random.RandomState(seed)
    images = [{"id": i + 1, "height": 480, "width": 640} for i in range(n_img)]
    cats = [{"id": c + 1, "frequency": "rcf"[c % 3]} for c in range(n_cat)]
    anns, dets = [], []
    for im in images:
        for c in cats:
            ng = rng.randint(0, 5)
            for _ in range(ng):
                side = float(rng.choice([rng.uniform(4, 30), rng.uniform(33, 95), rng.uniform(97, 300)]))
                w, h = side * rng.uniform(0.7, 1.4), side * rng.uniform(0.7, 1.4)
                x, y = rng.uniform(0, 640 - w), rng.uniform(0, 480 - h)
                anns.append({"id": len(anns) + 1, "image_id": im["id"], "category_id": c["id"], "bbox": [x, y, w, h],
                             "area": w * h, "iscrowd": 0, "ignore": int(rng.rand() < 0.1)})
                if rng.rand() < 0.8:                                       # a detection near it (IoU anywhere in 0.3 .. 1)
                    j = rng.uniform(-0.25, 0.25, 4) * np.array([w, h, w, h])
                    dets.append({"image_id": im["id"], "category_id": c["id"],
                                 "bbox": [x + j[0], y + j[1], max(w + j[2], 1.0), max(h + j[3], 1.0)], "score": float(rng.rand())})
                if rng.rand() < 0.3:                                       # a duplicate detection of the same object
                    dets.append({"image_id": im["id"], "category_id": c["id"], "bbox": [x + 1.0, y - 1.0, w, h],
                                 "score": float(rng.rand())})
            for _ in range(rng.randint(0, 3)):                             # false positives (also on categories without ground truth)
                w, h = rng.uniform(5, 200), rng.uniform(5, 200)
                dets.append({"image_id": im["id"], "category_id": c["id"], "bbox": [rng.uniform(0, 400), rng.uniform(0, 250), w, h],
                             "score": float(rng.rand())})
    # a few exact score ties: both sides use stable sorts
    for k in range(0, len(dets) - 1, 7):
        dets[k + 1]["score"] = dets[k]["score"]
    present = {}
    for a in anns:
        present.setdefault(a["image_id"], set()).add(a["category_id"])
    for im in images:                                                      # nothing is filtered by the federated rule
        im["neg_category_ids"] = [c["id"] for c in cats if c["id"] not in present.get(im["id"], set())]
        im["not_exhaustive_category_ids"] = []
    return {"images": images, "categories": cats, "annotations": anns}, dets


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_lvis_eval_matching_and_accumulation_equal_reference_cocoeval_cpp(seed):
    from oracle import build as OB
    ref = OB.load_ref()
    if ref is None or not hasattr(ref, "cocoeval_evaluate_images"):
        pytest.skip("oracle/_ref/dgref.so (reference C++ compiled in place) not available")
    gt, dets = _dataset(seed)
    ev = LE.LVISEval(gt, dets, iou_type="bbox", max_dets=300)
    ev.run()
    # ---- the same problem in the C++ evaluator's input form (fast_eval_api.py:55-90)
    img_ids, cat_ids = ev.img_ids, ev.cat_ids
    gts = {(i, c): [] for i in img_ids for c in cat_ids}
    for a in gt["annotations"]:
        gts[a["image_id"], a["category_id"]].append(a)
    dts = {(i, c): [] for i in img_ids for c in cat_ids}
    for k, d in enumerate(dets):
        e = dict(d)
        e["id"], e["area"] = k + 1, d["bbox"][2] * d["bbox"][3]
        dts[d["image_id"], d["category_id"]].append(e)
    G, D, I = [], [], []
    for i in img_ids:
        g_row, d_row, i_row = [], [], []
        for c in cat_ids:
            g, d = gts[i, c], dts[i, c]
            g_row.append([ref.InstanceAnnotation(int(x["id"]), 0.0, float(x["area"]), False, bool(x["ignore"])) for x in g])
            d_row.append([ref.InstanceAnnotation(int(x["id"]), float(x["score"]), float(x["area"]), False, False) for x in d])
            ds = sorted(d, key=lambda x: -x["score"])                       # computeIoU: rows in score order (stable), columns as stored
            i_row.append([[LE.box_iou_xywh(x["bbox"], y["bbox"]) for y in g] for x in ds] if g and ds else [])
        G.append(g_row), D.append(d_row), I.append(i_row)
    P = types.SimpleNamespace(recThrs=ev.rec_thrs.tolist(), iouThrs=ev.iou_thrs.tolist(), maxDets=[300], useCats=1,
                              areaRng=[list(map(float, r)) for r in ev.area_rng], catIds=cat_ids, imgIds=img_ids)
    E = ref.cocoeval_evaluate_images([list(map(float, r)) for r in ev.area_rng], 300, ev.iou_thrs.tolist(), I, G, D)
    acc = ref.cocoeval_accumulate(P, E)
    counts = list(acc["counts"])                                           # T, R, K, A, M
    prec = np.array(acc["precision"]).reshape(counts)[..., 0]
    rec = np.array(acc["recall"]).reshape(counts[:1] + counts[2:])[..., 0]
    assert prec.shape == ev.precision.shape and rec.shape == ev.recall.shape
    assert (prec > -1).sum() > 0.5 * prec.size                              # the comparison is not vacuous
    np.testing.assert_array_equal(prec > -1, ev.precision > -1)
    np.testing.assert_allclose(ev.precision, prec, rtol=0, atol=1e-12)
    np.testing.assert_allclose(ev.recall, rec, rtol=0, atol=1e-12)
    # and the summary numbers follow from those tensors the way COCOeval.summarize / LVISEval._summarize average them
    valid = prec[:, :, :, 0][prec[:, :, :, 0] > -1]
    assert abs(ev.summarize()["AP"] - float(valid.mean())) < 1e-12
