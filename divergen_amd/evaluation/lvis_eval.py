"""LVIS average precision: the arithmetic behind LVISEvaluator._eval_predictions
(D2/evaluation/lvis_evaluation.py:127-178 -> _evaluate_predictions_on_lvis :331-380), which the reference delegates to
lvis-api (`LVIS`, `LVISResults`, `LVISEval`; pip package `lvis`, NOT vendored and absent from this image).

Restated from the published lvis-api algorithm (lvis/eval.py, v0.5.x) -- PARITY UNPINNED by reference vectors; pinned by
hand-computable cases in tests/test_host_eval.py:
  * federated ground truth: a detection of category c on image I counts only if c has ground truth on I or c is in I's
    neg_category_ids; unmatched detections of categories in not_exhaustive_category_ids are ignored;
  * per (image, category, area range): greedy matching of score-sorted detections (<= max_dets per image, already enforced
    by the caller's top-300) to ground truth at IoU thresholds 0.50:0.05:0.95, ignored ground truth matched last;
  * per (category, area range): precision at 101 recall points from the score-sorted, cross-image detection list, made
    monotone from the right; AP = mean over valid (> -1) entries; APr / APc / APf average the categories of one frequency.
Masks are compared as COCO run-length encodings (uncompressed counts or the compressed string form libdgx writes).
"""
import json
from collections import defaultdict

import numpy as np


# ----------------------------------------------------------------------------------------------- RLE helpers
def rle_string_to_counts(s):
    """pycocotools maskApi.c rleFrString: the compressed ascii form -> run lengths."""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts, p, n = [], 0, len(s)
    while p < n:
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def _counts(rle):
    c = rle["counts"]
    return rle_string_to_counts(c) if isinstance(c, (str, bytes)) else list(c)


def rle_area(rle):
    return int(sum(_counts(rle)[1::2]))


def rle_runs(rle):
    """Run-length mask -> (starts, ends, cumulative lengths) of its foreground runs over the column-major pixel index, as int64
    arrays: the form `runs_iou` intersects.  LVISEval keeps one per annotation / detection, so an RLE string is decoded once."""
    c = np.asarray(_counts(rle), dtype=np.int64)
    e = np.cumsum(c)
    n = c.size // 2
    s, t = e[0:2 * n:2], e[1:2 * n:2]
    return s, t, np.concatenate([np.zeros(1, dtype=np.int64), np.cumsum(t - s)])


def _covered(runs, x):
    """Foreground pixels of `runs` in [0, x) for every x of an array."""
    s, t, cum = runs
    k = np.searchsorted(t, x, side="right")                 # runs that end at or before x
    kk = np.minimum(k, s.size - 1)
    return cum[k] + np.where(k < s.size, np.maximum(x - s[kk], 0), 0)


def runs_iou(a, b):
    """maskApi.c rleIou with iscrowd = 0 on two `rle_runs` triples: the intersection is, for every run of a, b's coverage of
    [start, end) -- two binary searches per run instead of a merge walk over both run lists in the interpreter."""
    inter = int((_covered(b, a[1]) - _covered(b, a[0])).sum()) if a[0].size and b[0].size else 0
    union = int(a[2][-1]) + int(b[2][-1]) - inter
    return inter / union if union > 0 else 0.0


def rle_iou(a, b):
    """Intersection over union of two run-length masks of the same size (maskApi.c rleIou with iscrowd = 0)."""
    return runs_iou(rle_runs(a), rle_runs(b))


def box_iou_xywh(a, b):
    ix = max(0.0, min(a[0] + a[2], b[0] + b[2]) - max(a[0], b[0]))
    iy = max(0.0, min(a[1] + a[3], b[1] + b[3]) - max(a[1], b[1]))
    inter = ix * iy
    union = a[2] * a[3] + b[2] * b[3] - inter
    return inter / union if union > 0 else 0.0


def ann_to_rle(ann, h, w):
    """LVIS.ann_to_rle: polygons -> merged RLE; dict segmentations pass through."""
    seg = ann["segmentation"]
    if isinstance(seg, dict):
        return seg
    from ..data.build import polygons_to_bitmask
    m = polygons_to_bitmask(seg, h, w)
    flat = m.T.reshape(-1)            # column-major
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(edges).tolist()
    if flat.size and flat[0]:
        counts = [0] + counts
    return {"size": [h, w], "counts": counts}


# ----------------------------------------------------------------------------------------------- LVISEval
class LVISEval:
    def __init__(self, gt_json, results, iou_type="segm", max_dets=300):
        gt = json.load(open(gt_json)) if isinstance(gt_json, str) else gt_json
        self.iou_type, self.max_dets = iou_type, max_dets
        self.iou_thrs = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
        self.rec_thrs = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
        self.area_rng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.area_rng_lbl = ["all", "small", "medium", "large"]
        self.imgs = {im["id"]: im for im in gt["images"]}
        self.cats = {c["id"]: c for c in gt["categories"]}
        self.img_ids = sorted(self.imgs)
        self.cat_ids = sorted(self.cats)
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        img_pl = defaultdict(set)
        for ann in gt["annotations"]:
            a = dict(ann)
            a.setdefault("ignore", 0)
            if iou_type == "segm":
                im = self.imgs[a["image_id"]]
                a["segmentation"] = ann_to_rle(a, im["height"], im["width"])
            self._gts[a["image_id"], a["category_id"]].append(a)
            img_pl[a["image_id"]].add(a["category_id"])
        img_nl = {i: set(im.get("neg_category_ids", [])) for i, im in self.imgs.items()}
        self.img_nel = {i: set(im.get("not_exhaustive_category_ids", [])) for i, im in self.imgs.items()}
        # LVISResults: ids, areas; at most max_dets per image by score
        per_img = defaultdict(list)
        for k, r in enumerate(results):
            d = dict(r)
            d["id"] = k + 1
            if iou_type == "segm":
                d["area"] = rle_area(d["segmentation"])
            else:
                d["area"] = d["bbox"][2] * d["bbox"][3]
            per_img[d["image_id"]].append(d)
        for img_id, dts in per_img.items():
            if max_dets >= 0 and len(dts) > max_dets:
                dts = sorted(dts, key=lambda x: -x["score"])[:max_dets]
            for d in dts:
                c = d["category_id"]
                if img_id not in self.imgs or (c not in img_nl[img_id] and c not in img_pl[img_id]):
                    continue
                self._dts[img_id, c].append(d)
        self.freq_groups = [[], [], []]
        for idx, c in enumerate(self.cat_ids):
            f = self.cats[c].get("frequency", "f")
            self.freq_groups[["r", "c", "f"].index(f)].append(idx)

    def _iou(self, d, g):
        if self.iou_type != "segm":
            return box_iou_xywh(d["bbox"], g["bbox"])
        for x in (d, g):                 # d, g are this evaluator's own copies of the records: the decoded runs are kept on them
            if "_runs" not in x:
                x["_runs"] = rle_runs(x["segmentation"])
        return runs_iou(d["_runs"], g["_runs"])

    def _evaluate_img(self, img_id, cat_id, area_rng, ious):
        gt, dt = self._gts[img_id, cat_id], self._dts[img_id, cat_id]
        if len(gt) == 0 and len(dt) == 0:
            return None
        ig = np.array([1 if (g["ignore"] or g["area"] < area_rng[0] or g["area"] > area_rng[1]) else 0 for g in gt], dtype=int)
        gt_idx = np.argsort(ig, kind="mergesort")
        gt = [gt[i] for i in gt_idx]
        gt_ig = ig[gt_idx]
        dt_idx = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in dt_idx]
        iou = ious[:, gt_idx] if len(ious) > 0 else ious
        T, G, D = len(self.iou_thrs), len(gt), len(dt)
        gt_m, dt_m, dt_ig = np.zeros((T, G)), np.zeros((T, D)), np.zeros((T, D))
        for t, thr in enumerate(self.iou_thrs):
            if len(iou) == 0:
                break
            for di in range(D):
                best = min(thr, 1 - 1e-10)
                m = -1
                for gi in range(G):
                    if gt_m[t, gi] > 0:
                        continue
                    if m > -1 and gt_ig[m] == 0 and gt_ig[gi] == 1:
                        break
                    if iou[di, gi] < best:
                        continue
                    best = iou[di, gi]
                    m = gi
                if m == -1:
                    continue
                dt_ig[t, di] = gt_ig[m]
                dt_m[t, di] = gt[m]["id"]
                gt_m[t, m] = dt[di]["id"]
        mask = np.array([d["area"] < area_rng[0] or d["area"] > area_rng[1] or d["category_id"] in self.img_nel[d["image_id"]]
                         for d in dt], dtype=bool).reshape(1, D)
        dt_ig = np.logical_or(dt_ig, np.logical_and(dt_m == 0, np.repeat(mask, T, 0)))
        return {"dt_matches": dt_m, "dt_scores": [d["score"] for d in dt], "gt_ignore": gt_ig, "dt_ignore": dt_ig}

    def run(self):
        T, R, K, A = len(self.iou_thrs), len(self.rec_thrs), len(self.cat_ids), len(self.area_rng)
        precision = -np.ones((T, R, K, A))
        recall = -np.ones((T, K, A))
        for k, cat in enumerate(self.cat_ids):
            ious = {}
            for img in self.img_ids:
                gt, dt = self._gts[img, cat], self._dts[img, cat]
                if not gt and not dt:
                    continue
                dts = sorted(dt, key=lambda x: -x["score"])
                self._dts[img, cat] = dts
                ious[img] = np.array([[self._iou(d, g) for g in gt] for d in dts]).reshape(len(dts), len(gt)) if gt and dts else []
            for a, rng in enumerate(self.area_rng):
                E = [self._evaluate_img(img, cat, rng, ious[img]) for img in self.img_ids if img in ious]
                E = [e for e in E if e is not None]
                if not E:
                    continue
                scores = np.concatenate([e["dt_scores"] for e in E], axis=0)
                order = np.argsort(-scores, kind="mergesort")
                dt_m = np.concatenate([e["dt_matches"] for e in E], axis=1)[:, order]
                dt_ig = np.concatenate([e["dt_ignore"] for e in E], axis=1)[:, order]
                gt_ig = np.concatenate([e["gt_ignore"] for e in E])
                num_gt = np.count_nonzero(gt_ig == 0)
                if num_gt == 0:
                    continue
                tps = np.logical_and(dt_m, np.logical_not(dt_ig))
                fps = np.logical_and(np.logical_not(dt_m), np.logical_not(dt_ig))
                tp_sum, fp_sum = np.cumsum(tps, axis=1).astype(float), np.cumsum(fps, axis=1).astype(float)
                for t in range(T):
                    tp, fp = tp_sum[t], fp_sum[t]
                    n = len(tp)
                    rc = tp / num_gt
                    recall[t, k, a] = rc[-1] if n else 0
                    pr = (tp / (fp + tp + np.spacing(1))).tolist()
                    for i in range(n - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    idx = np.searchsorted(rc, self.rec_thrs, side="left")
                    at = [0.0] * R
                    for ri, pi in enumerate(idx):
                        if pi < n:
                            at[ri] = pr[pi]
                    precision[t, :, k, a] = np.array(at)
        self.precision, self.recall = precision, recall
        return self.summarize()

    def _ap(self, iou_thr=None, area="all", freq=None):
        a = self.area_rng_lbl.index(area)
        p = self.precision
        if freq is not None:
            p = p[:, :, self.freq_groups[freq], :]
        if iou_thr is not None:
            p = p[np.where(np.isclose(self.iou_thrs, iou_thr))[0]]
        p = p[:, :, :, a]
        p = p[p > -1]
        return float(np.mean(p)) if p.size else -1.0

    def summarize(self):
        r = {"AP": self._ap(), "AP50": self._ap(0.5), "AP75": self._ap(0.75), "APs": self._ap(area="small"),
             "APm": self._ap(area="medium"), "APl": self._ap(area="large"), "APr": self._ap(freq=0), "APc": self._ap(freq=1),
             "APf": self._ap(freq=2)}
        rc = self.recall[:, :, 0]
        rc = rc[rc > -1]
        r["AR@%d" % self.max_dets] = float(np.mean(rc)) if rc.size else -1.0
        return r


def evaluate_predictions_on_lvis(gt_json, results, iou_type, max_dets_per_image=300):
    """_evaluate_predictions_on_lvis (lvis_evaluation.py:331-380): metrics x 100; segm results are scored without their boxes."""
    if len(results) == 0:
        return {m: float("nan") for m in ["AP", "AP50", "AP75", "APs", "APm", "APl", "APr", "APc", "APf"]}
    if iou_type == "segm":
        results = [{k: v for k, v in r.items() if k != "bbox"} for r in results]
    out = LVISEval(gt_json, results, iou_type, max_dets_per_image).run()
    return {m: float(out[m] * 100) for m in ["AP", "AP50", "AP75", "APs", "APm", "APl", "APr", "APc", "APf"]}
