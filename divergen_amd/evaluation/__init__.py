"""Results writer of the evaluation loop: host mirror of `instances_to_coco_json`
(D2/evaluation/coco_evaluation.py:380-440, called per image from LVISEvaluator.process, lvis_evaluation.py:80-97).

The reference moves N x H x W bitmasks to the host and run-length encodes them one by one with pycocotools.  Here the masks
are encoded on the GPU (dgx_paste_rle straight from the SxS probabilities when `detector_postprocess(..., mask_format="rle")`
was used, dgx_rle_encode for bitmasks) and only run lengths cross PCIe."""
from ..layers.mask_ops import rle_encode_bitmasks


def instances_to_coco_json(instances, img_id):
    """list[dict] in COCO results format: image_id, category_id, bbox (XYWH), score, segmentation {"size", "counts": str}."""
    n = len(instances)
    if n == 0:
        return []
    b = instances.pred_boxes.tensor.detach().float().cpu()
    b = b.clone()
    b[:, 2] -= b[:, 0]           # BoxMode XYXY_ABS -> XYWH_ABS
    b[:, 3] -= b[:, 1]
    boxes = b.tolist()
    scores = instances.scores.tolist()
    classes = instances.pred_classes.tolist()
    rles = None
    if instances.has("pred_masks_rle"):
        rles = instances.pred_masks_rle
    elif instances.has("pred_masks"):
        rles = rle_encode_bitmasks(instances.pred_masks)
    if rles is not None:
        rles = [{"size": r["size"], "counts": r["counts"].decode("utf-8") if isinstance(r["counts"], bytes) else r["counts"]}
                for r in rles]
    out = []
    for k in range(n):
        r = {"image_id": img_id, "category_id": classes[k], "bbox": boxes[k], "score": scores[k]}
        if rles is not None:
            r["segmentation"] = rles[k]
        out.append(r)
    return out


class LVISEvaluator:
    """LVISEvaluator (D2/evaluation/lvis_evaluation.py:22-178): collects per-image results like LVISResultsWriter and, when the
    split's annotation json is available, scores them -- box AP and mask AP with the r / c / f breakdown -- with this build's
    restatement of lvis-api's LVISEval (evaluation/lvis_eval.py).  `evaluate()` returns {"bbox": {...}, "segm": {...}}."""

    def __init__(self, dataset_name, cfg=None, distributed=True, output_dir=None, max_dets_per_image=300, tasks=("bbox", "segm")):
        from ..data.build import dataset_files
        self._json, _ = dataset_files(dataset_name)
        self._writer = LVISResultsWriter(output_dir, distributed)
        self._max_dets, self._tasks = max_dets_per_image, tasks
        if cfg is not None and not cfg.MODEL.MASK_ON:
            self._tasks = tuple(t for t in tasks if t != "segm")

    def reset(self):
        self._writer.reset()

    def process(self, inputs, outputs):
        self._writer.process(inputs, outputs)

    def evaluate(self):
        import os
        from collections import OrderedDict
        from ..utils import comm
        from .lvis_eval import evaluate_predictions_on_lvis
        if self._writer.evaluate() is None or not comm.is_main_process():
            return {}
        results = getattr(self._writer, "results", [])
        out = OrderedDict()
        if not os.path.isfile(self._json):
            return out
        import json
        with open(self._json) as f:
            gt = json.load(f)
        if "annotations" not in gt:                       # test-dev style split: predictions only (lvis_evaluation.py:162-164)
            return out
        for task in self._tasks:
            if task == "segm" and not any("segmentation" in r for r in results):
                continue
            out[task] = evaluate_predictions_on_lvis(gt, results, task, self._max_dets)
        return out


def print_csv_format(results):
    """D2/evaluation/testing.py:print_csv_format."""
    import logging
    log = logging.getLogger("divergen_amd")
    for task, res in results.items():
        if isinstance(res, dict):
            important = [(k, v) for k, v in res.items() if "-" not in k]
            log.info("copypaste: Task: {}".format(task))
            log.info("copypaste: " + ",".join([k[0] for k in important]))
            log.info("copypaste: " + ",".join(["{0:.4f}".format(k[1]) for k in important]))


class LVISResultsWriter:
    """The prediction-collecting half of LVISEvaluator (D2/evaluation/lvis_evaluation.py: reset :76, process :79-97,
    evaluate :99-125 and the json dump of _eval_predictions :132-160): per-image results in LVIS format, gathered to rank 0
    and written to `lvis_instances_results.json` with 1-indexed category ids.  The AP computation itself
    (`_evaluate_predictions_on_lvis`, lvis-api) is not available in this image: `evaluate()` returns {} after writing the
    file, which is what the reference does when the split has no annotations (:162-164)."""

    def __init__(self, output_dir=None, distributed=True, reverse_id_mapping=None, mask_on_device=True):
        self._output_dir, self._distributed = output_dir, distributed
        self._reverse = reverse_id_mapping
        self._predictions = []

    def reset(self):
        self._predictions = []

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            pred = {"image_id": inp["image_id"]}
            if "instances" in out:
                # masks are run-length encoded on the GPU (no bitmask transfer); boxes / scores / classes go to the host
                pred["instances"] = instances_to_coco_json(out["instances"], inp["image_id"])
            self._predictions.append(pred)

    def evaluate(self):
        import itertools
        import json
        import os
        from ..utils import comm
        preds = self._predictions
        if self._distributed and comm.get_world_size() > 1:
            comm.synchronize()
            gathered = comm.gather(preds, dst=0)
            if not comm.is_main_process():
                return
            preds = list(itertools.chain(*gathered))
        if len(preds) == 0:
            return {}
        results = list(itertools.chain(*[p["instances"] for p in preds if "instances" in p]))
        for r in results:
            r["category_id"] = self._reverse[r["category_id"]] if self._reverse is not None else r["category_id"] + 1
        if self._output_dir:
            os.makedirs(self._output_dir, exist_ok=True)
            with open(os.path.join(self._output_dir, "lvis_instances_results.json"), "w") as f:
                f.write(json.dumps(results))
        self.results = results
        return {}


def inference_on_dataset(model, data_loader, evaluator, mask_format="rle"):
    """D2/evaluation/evaluator.py:103-190 without the timing log: model in eval mode under no_grad, evaluator.process per
    batch, evaluator.evaluate() at the end.  With mask_format="rle" the model's post-processing leaves COCO run-length
    dicts (`pred_masks_rle`) instead of (N,H,W) bitmasks -- see modeling/meta_arch/custom_rcnn.detector_postprocess."""
    import torch
    from ..modeling.meta_arch.custom_rcnn import detector_postprocess
    was_training = model.training
    model.eval()
    evaluator.reset()
    try:
        with torch.no_grad():
            for inputs in data_loader:
                if mask_format == "rle":
                    raw = model.inference(inputs, do_postprocess=False)
                    outputs = [{"instances": detector_postprocess(r, inp.get("height", r.image_size[0]),
                                                                  inp.get("width", r.image_size[1]), mask_format="rle")}
                               for r, inp in zip(raw, inputs)]
                else:
                    outputs = model(inputs)
                evaluator.process(inputs, outputs)
    finally:
        model.train(was_training)
    return evaluator.evaluate() or {}
