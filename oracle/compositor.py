"""Oracle: DiverGen instance copy-paste compositor, numpy, sequential like the reference.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
DG/divergen/data/custom_build_copypaste_mapper.py:
  pad_to_hw :38-43 (cv2.warpAffine with an integer translation == shifted copy, zero border),
  start_xy :59-66, get_updated_masks :73-77, get_bboxes :79-92, InstPool._copy_paste :510-566,
  InstPool._cat_a_new_image :488-507 (the loop over pastes);
DG/divergen/data/transforms/custom_cp_method.py:5-9 ('basic' blend, integer arithmetic).
cv2 is absent here and on the GPU box; decode/resize stay outside the compositor contract.
"""
import numpy as np

BBOX_OCCLUDED_THR = 10   # mapper.py:164
MASK_OCCLUDED_THR = 300  # mapper.py:165


def get_bboxes(masks):
    """(n,H,W) -> (n,4) float32 [x0,y0,x1+1,y1+1]; empty mask -> zeros.  :79-92."""
    n = len(masks)
    boxes = np.zeros((n, 4), np.float32)
    cols = masks.any(axis=1)
    rows = masks.any(axis=2)
    for i in range(n):
        x = np.flatnonzero(cols[i])
        y = np.flatnonzero(rows[i])
        if len(x) and len(y):
            boxes[i] = (x[0], y[0], x[-1] + 1, y[-1] + 1)
    return boxes


def place(rgba, x0, y0, H, W):
    """Translate an (h,w,4) RGBA patch onto an HxW canvas -> image (4,H,W) u8, mask (1,H,W) u8.
    Restates pad_to_hw/start_xy for integer (x0,y0) (may be negative / overhang)."""
    h, w = rgba.shape[:2]
    img = np.zeros((4, H, W), np.uint8)
    m = np.zeros((1, H, W), np.uint8)
    ys, xs, ye, xe = max(y0, 0), max(x0, 0), min(y0 + h, H), min(x0 + w, W)
    if ye > ys and xe > xs:
        sub = rgba[ys - y0:ye - y0, xs - x0:xe - x0]
        img[:, ys:ye, xs:xe] = sub.transpose(2, 0, 1)
        m[0, ys:ye, xs:xe] = sub[..., 3] > 0
    return img, m


def copy_paste(dst, src_img, src_mask, src_label):
    """One paste.  dst: dict(image (3,H,W) u8, masks (n,H,W) u8, boxes (n,4) f32, labels (n,) i64,
    source (n,) i64).  :510-566."""
    src_boxes = get_bboxes(src_mask)
    if len(src_boxes) == 0:
        return dst
    composed = np.where(np.any(src_mask, axis=0), 1, 0)
    upd = np.where(composed, 0, dst["masks"])
    upd_boxes = get_bboxes(upd)
    ok_box = np.all(np.abs(upd_boxes - dst["boxes"]) <= BBOX_OCCLUDED_THR, axis=-1)
    ok_area = upd.sum(axis=(1, 2)) > MASK_OCCLUDED_THR
    valid = ok_box | ok_area
    img = (dst["image"] * (1 - composed) + src_img[:3] * composed).astype(dst["image"].dtype)
    return dict(image=img,
                masks=np.concatenate([upd[valid], src_mask]),
                boxes=np.concatenate([upd_boxes[valid], src_boxes]),
                labels=np.concatenate([dst["labels"][valid], np.atleast_1d(src_label)]),
                source=np.concatenate([dst["source"][valid], [1]]))


def composite(image, masks, boxes, labels, pastes, H=None, W=None):
    """pastes: list of (rgba (h,w,4) u8, x0, y0, label).  Returns the final dict."""
    H = H or image.shape[1]
    W = W or image.shape[2]
    dst = dict(image=image.copy(), masks=masks.copy(), boxes=boxes.copy(), labels=labels.copy(),
               source=np.zeros(len(labels), np.int64))
    for rgba, x0, y0, lab in pastes:
        si, sm = place(rgba, int(x0), int(y0), H, W)
        dst = copy_paste(dst, si, sm, lab)
    return dst
