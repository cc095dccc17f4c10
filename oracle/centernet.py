"""Oracle: CenterNet proposal generator targets / losses / decoding on torch-CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
CN/modeling/dense_heads/centernet.py (CN = BSGAL/third_party/CenterNet2/projects/CenterNet2/centernet):
  compute_grids :317-335, _get_ground_truth :338-436, _get_label_inds :439-483,
  assign_fpn_level :486-502, assign_reg_fpn :505-516, _get_reg_targets :519-530,
  _create_agn_heatmaps_from_dist :551-562, get_center3x3 :576-592, losses :237-314,
  predict_single_level :643-708, nms_and_topK :711-737;
CN/modeling/layers/heatmap_focal_loss.py:51-85, CN/modeling/layers/iou_loss.py:10-63.
Only the configuration the shipped YAMLs select is restated (ONLY_PROPOSAL + WITH_AGN_HM,
NOT_NORM_REG, no MORE_POS, giou).
"""
import torch

from . import roi

INF = 100000000
SOI = [[0, 80], [64, 160], [128, 320], [256, 640], [512, 10000000]]
STRIDES = (8, 16, 32, 64, 128)


def compute_grids(shapes, strides=STRIDES):
    """shapes: [(h,w)] per level -> list of (h*w, 2) (x,y) fp32 centres.  :317-335."""
    grids = []
    for (h, w), s in zip(shapes, strides):
        xs = torch.arange(0, w * s, step=s, dtype=torch.float32)
        ys = torch.arange(0, h * s, step=s, dtype=torch.float32)
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        grids.append(torch.stack([xx.reshape(-1), yy.reshape(-1)], 1) + s // 2)
    return grids


def label_inds(gt_boxes_list, shapes, strides=STRIDES, soi=SOI):
    """pos_inds (N',) int64 into the level-major flattened (level, image, y, x) layout.  :439-483."""
    L, B = len(strides), len(gt_boxes_list)
    hw = torch.tensor(shapes, dtype=torch.int64)
    loc = hw[:, 0] * hw[:, 1]
    bases, s = [], 0
    for l in range(L):
        bases.append(s)
        s += B * int(loc[l])
    bases = torch.tensor(bases, dtype=torch.int64)
    st = torch.tensor(strides, dtype=torch.float32)
    sr = torch.tensor(soi, dtype=torch.float32)
    out = []
    for i, bx in enumerate(gt_boxes_list):
        n = bx.shape[0]
        c = (bx[:, [0, 1]] + bx[:, [2, 3]]) / 2  # n,2
        ci = (c[:, None, :] / st[None, :, None]).long()  # n,L,2 (trunc)
        ind = bases[None] + i * loc[None] + ci[:, :, 1] * hw[None, :, 1] + ci[:, :, 0]
        crit = ((bx[:, 2:] - bx[:, :2]) ** 2).sum(1) ** 0.5 / 2  # :496
        cared = (crit[:, None] >= sr[None, :, 0]) & (crit[:, None] <= sr[None, :, 1])
        out.append(ind[cared].reshape(-1))
    return torch.cat(out) if out else torch.zeros(0, dtype=torch.int64)


def ground_truth(gt_boxes_list, shapes, strides=STRIDES, soi=SOI, hm_min_overlap=0.8, min_radius=4):
    """-> pos_inds, reg_targets (M*B,4), flattened_hms (M*B,1), level-major.  :338-436."""
    delta = (1 - hm_min_overlap) / (1 + hm_min_overlap)
    grids_l = compute_grids(shapes, strides)
    nloc = [g.shape[0] for g in grids_l]
    grids = torch.cat(grids_l)
    M = grids.shape[0]
    st = torch.cat([torch.full((n,), float(s)) for n, s in zip(nloc, strides)])
    rng = torch.cat([torch.tensor(r, dtype=torch.float32).reshape(1, 2).expand(n, 2)
                     for n, r in zip(nloc, soi)])
    regs, hms = [], []
    for bx in gt_boxes_list:
        N = bx.shape[0]
        if N == 0:
            regs.append(torch.zeros(M, 4) - INF)
            hms.append(torch.zeros(M, 1))
            continue
        area = roi.box_area(bx)
        gx, gy = grids[:, 0:1], grids[:, 1:2]
        l, t = gx - bx[None, :, 0], gy - bx[None, :, 1]
        r, b = bx[None, :, 2] - gx, bx[None, :, 3] - gy
        reg = torch.stack([l, t, r, b], 2)  # M,N,4
        ctr = (bx[:, [0, 1]] + bx[:, [2, 3]]) / 2  # N,2
        se = st[:, None, None]
        disc = ((ctr[None] / se).int() * se).float() + se / 2  # M,N,2 (:395-396)
        gxy = grids[:, None, :]
        is_peak = ((gxy - disc) ** 2).sum(2) == 0
        in_box = reg.min(2)[0] > 0
        c33 = ((gxy[..., 0] - disc[..., 0]).abs() <= st[:, None]) & \
              ((gxy[..., 1] - disc[..., 1]).abs() <= st[:, None]) & in_box  # :576-592
        crit = ((reg[:, :, :2] + reg[:, :, 2:]) ** 2).sum(2) ** 0.5 / 2  # :512-513
        cared = (crit >= rng[:, [0]]) & (crit <= rng[:, [1]])
        mask = c33 & cared
        d2 = ((gxy - ctr[None]) ** 2).sum(2)
        d2[is_peak] = 0
        rad2 = torch.clamp(delta ** 2 * 2 * area, min=min_radius ** 2)
        wd = d2 / rad2[None]
        dist = wd.clone()
        dist[mask == 0] = INF * 1.0
        mn, mi = dist.min(1)
        rt = reg[torch.arange(M), mi]
        rt[mn == INF] = -INF
        hm = torch.exp(-wd.min(1)[0])[:, None]
        hm[hm < 1e-4] = 0
        regs.append(rt)
        hms.append(hm)
    # image-first -> level-first (CN/modeling/dense_heads/utils.py:16-29), reg / stride
    reg_lv, hm_lv = [], []
    for l in range(len(strides)):
        a, bnd = sum(nloc[:l]), sum(nloc[:l + 1])
        reg_lv.append(torch.cat([r[a:bnd] for r in regs]) / float(strides[l]))
        hm_lv.append(torch.cat([h[a:bnd] for h in hms]))
    pos = label_inds(gt_boxes_list, shapes, strides, soi)
    return pos, torch.cat(reg_lv), torch.cat(hm_lv)


def giou_loss(pred, target, weight=None, reduction="sum"):
    """iou_loss.py:10-63 (loc_loss_type='giou'; l/t/r/b parametrisation, +1 smoothing)."""
    pl, pt, pr, pb = pred.unbind(1)
    tl, tt, tr, tb = target.unbind(1)
    ta, pa = (tl + tr) * (tt + tb), (pl + pr) * (pt + pb)
    wi = torch.min(pl, tl) + torch.min(pr, tr)
    hi = torch.min(pb, tb) + torch.min(pt, tt)
    gw = torch.max(pl, tl) + torch.max(pr, tr)
    gh = torch.max(pb, tb) + torch.max(pt, tt)
    ac = gw * gh
    ai = wi * hi
    au = ta + pa - ai
    ious = (ai + 1.0) / (au + 1.0)
    losses = 1 - (ious - (ac - au) / ac)
    if weight is not None:
        losses = losses * weight
    return losses.sum() if reduction == "sum" else losses


def binary_focal(logits, targets, pos_inds, alpha=0.25, beta=4, gamma=2, clamp=1e-4, ignore_high_fp=0.85):
    """heatmap_focal_loss.py:51-85 (sigmoid applied out of place here)."""
    pred = torch.clamp(torch.sigmoid(logits), min=clamp, max=1 - clamp)
    negw = torch.pow(1 - targets, beta)
    pp = pred[pos_inds]
    pos = torch.log(pp) * torch.pow(1 - pp, gamma)
    neg = torch.log(1 - pred) * torch.pow(pred, gamma) * negw
    if ignore_high_fp > 0:
        neg = (pred < ignore_high_fp).float() * neg
    pos, neg = -pos.sum(), -neg.sum()
    if alpha >= 0:
        pos, neg = alpha * pos, (1 - alpha) * neg
    return pos, neg


def losses(pos_inds, reg_targets, hms, reg_pred, agn_logits, world=1, reg_weight=1.0,
           pos_weight=0.5, neg_weight=0.5, ignore_high_fp=0.85):
    """:237-314 with ONLY_PROPOSAL / WITH_AGN_HM / NOT_NORM_REG, single process."""
    num_pos_avg = max(pos_inds.numel() * world / world, 1.0)
    ri = torch.nonzero(reg_targets.max(1)[0] >= 0).squeeze(1)
    wmap = hms.max(1)[0][ri] * 0 + 1
    reg_norm = max(float(wmap.sum()), 1)
    loc = reg_weight * giou_loss(reg_pred[ri], reg_targets[ri], wmap, "sum") / reg_norm
    p, n = binary_focal(agn_logits, hms.max(1)[0], pos_inds, ignore_high_fp=ignore_high_fp)
    return {"loss_centernet_loc": loc,
            "loss_centernet_agn_pos": pos_weight * p / num_pos_avg,
            "loss_centernet_agn_neg": neg_weight * n / num_pos_avg}


def predict_level(grids, hm, reg, thresh, topk):
    """One level, one image.  hm (HW,) sigmoid heat-map, reg (HW,4) already * stride.
    -> boxes (n,4), scores (n,) = sqrt(hm).  :643-708.  Order: descending score (the reference's
    topk(sorted=False) order is unspecified; callers compare as sets / sorted)."""
    cand = torch.nonzero(hm > thresh).squeeze(1)
    sc = hm[cand]
    if cand.numel() > topk:
        sc, ti = sc.topk(topk, sorted=False)
        cand = cand[ti]
    g, r = grids[cand], reg[cand]
    det = torch.stack([g[:, 0] - r[:, 0], g[:, 1] - r[:, 1], g[:, 0] + r[:, 2], g[:, 1] + r[:, 3]], 1)
    det[:, 2] = torch.max(det[:, 2], det[:, 0] + 0.01)
    det[:, 3] = torch.max(det[:, 3], det[:, 1] + 0.01)
    return det, torch.sqrt(sc)


def nms_and_topk(boxes, scores, nms_thresh, post_topk):
    """:711-737: class-agnostic NMS then keep everything >= the k-th largest score (ties kept)."""
    keep = roi.nms(boxes, scores, nms_thresh)
    boxes, scores = boxes[keep], scores[keep]
    n = len(scores)
    if n > post_topk:
        th, _ = torch.kthvalue(scores.float(), n - post_topk + 1)
        k = torch.nonzero(scores >= th.item()).squeeze(1)
        boxes, scores = boxes[k], scores[k]
    return boxes, scores
