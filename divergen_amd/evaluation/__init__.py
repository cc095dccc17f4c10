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
