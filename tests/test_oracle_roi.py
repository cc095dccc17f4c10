"""Oracle ROIAlign/NMS/box utilities vs the reference's known-answer tests, the compiled in-tree
reference sources (oracle/_ref, angle 0) and golden vectors of D2's matcher/sampling/box2box."""
import numpy as np
import pytest
import torch

from oracle import roi as O
from oracle.build import load_ref


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_roi_align_kat_d2t():
    # expected matrices: D2T/layers/test_roi_align.py:14-47
    inp = torch.arange(25, dtype=torch.float32).reshape(1, 1, 5, 5)
    rois = torch.tensor([[0, 1, 1, 3, 3]], dtype=torch.float32)
    old = O.roi_align(inp, rois, 1.0, 4, 0, aligned=False)[0, 0]
    new = O.roi_align(inp, rois, 1.0, 4, 0, aligned=True)[0, 0]
    assert np.allclose(old.numpy(), [[7.5, 8, 8.5, 9], [10, 10.5, 11, 11.5], [12.5, 13, 13.5, 14], [15, 15.5, 16, 16.5]])
    assert np.allclose(new.numpy(), [[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0], [12.0, 12.5, 13.0, 13.5]])


def test_roi_align_empty_and_grad_properties():
    # D2T/layers/test_roi_align.py:111-128: empty box -> zeros & zero grad; empty batch shape
    inp = torch.rand(1, 3, 10, 10)
    rois = torch.tensor([[0, 3, 3, 3, 3]], dtype=torch.float32)
    out = O.roi_align(inp, rois, 1.0, 7, 0, True)
    assert out.shape == (1, 3, 7, 7) and (out == 0).all()
    gi = O.roi_align_backward(torch.ones_like(out), rois, 1.0, (1, 3, 10, 10), 0, True)
    assert (gi == 0).all()
    assert O.roi_align(inp, torch.zeros(0, 5), 1.0, 7).shape == (0, 3, 7, 7)


def _rand_rois(g, n, B, H, W):
    xy = torch.rand(n, 2, generator=g) * torch.tensor([W * 0.7, H * 0.7])
    wh = torch.rand(n, 2, generator=g) * torch.tensor([W * 0.6, H * 0.6]) + 0.5
    b = torch.randint(0, B, (n, 1), generator=g).float()
    return torch.cat([b, xy, xy + wh], 1)


def test_roi_align_vs_compiled_reference():
    """D2T/modeling/test_roi_pooler.py:14-59: ROIAlignV2 == in-tree ROIAlignRotated @ angle 0 (atol 1e-4)."""
    ref = load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference and no prebuilt .so)")
    g = torch.Generator().manual_seed(5)
    feat = torch.rand(2, 4, 10, 8, generator=g)
    rois = _rand_rois(g, 40, 2, 10 * 4, 8 * 4)
    scale = 0.25
    mine = O.roi_align(feat, rois, scale, 14, 0, True)
    cx, cy = (rois[:, 1] + rois[:, 3]) / 2, (rois[:, 2] + rois[:, 4]) / 2
    w, h = rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]
    rr = torch.stack([rois[:, 0], cx, cy, w, h, torch.zeros_like(w)], 1)
    theirs = ref.roi_align_rotated_forward(feat, rr, scale, 14, 14, 0)
    torch.testing.assert_close(mine, theirs, atol=1e-4, rtol=0)
    go = torch.rand(mine.shape, generator=g)
    gmine = O.roi_align_backward(go, rois, scale, tuple(feat.shape), 0, True)
    gtheirs = ref.roi_align_rotated_backward(go, rr, scale, 14, 14, 2, 4, 10, 8, 0)
    torch.testing.assert_close(gmine, gtheirs, atol=1e-3, rtol=1e-4)


def test_nms_vs_compiled_reference_and_bruteforce():
    g = torch.Generator().manual_seed(9)
    xy = torch.rand(300, 2, generator=g) * 100
    wh = torch.rand(300, 2, generator=g) * 40 + 2
    boxes = torch.cat([xy, xy + wh], 1)
    scores = torch.rand(300, generator=g)
    for thr in (0.3, 0.5, 0.9):
        keep = O.nms(boxes, scores, thr)
        # brute force restatement in torch
        order = torch.sort(scores, descending=True, stable=True)[1]
        iou = O.pairwise_iou(boxes, boxes)
        dead = torch.zeros(300, dtype=torch.bool)
        bf = []
        for i in order.tolist():
            if dead[i]:
                continue
            bf.append(i)
            dead |= iou[i] > thr
        assert keep.tolist() == bf
        ref = load_ref()
        if ref is not None:
            cx, cy = (boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2
            r5 = torch.stack([cx, cy, wh[:, 0], wh[:, 1], torch.zeros(300)], 1)
            kr = ref.nms_rotated(r5, scores, thr)
            # polygon-clip IoU may differ from axis-aligned IoU in the last ulp near the threshold
            a, b = set(keep.tolist()), set(kr.tolist())
            assert len(a ^ b) <= 2


def test_batched_nms_per_class():
    g = torch.Generator().manual_seed(11)
    xy = torch.rand(200, 2, generator=g) * 50
    boxes = torch.cat([xy, xy + 10], 1)
    scores = torch.rand(200, generator=g)
    idxs = torch.randint(0, 3, (200,), generator=g)
    keep = O.batched_nms(boxes, scores, idxs, 0.5)
    # coordinate-offset trick (the other torchvision strategy) must give the same set
    off = idxs.float() * (boxes.max() + 1)
    keep2 = O.nms(boxes + off[:, None], scores, 0.5)
    assert keep.tolist() == keep2.tolist()
    assert (scores[keep][:-1] >= scores[keep][1:]).all()


def test_matcher_kat_d2t():
    # D2T/modeling/test_matcher.py:16-29 uses thresholds [0.3,0.7], labels [0,-1,1] + low quality
    # matches; the path here has no low-quality matches, so pin the threshold part only:
    q = torch.tensor([[0.15, 0.45, 0.2, 0.6], [0.3, 0.65, 0.05, 0.1], [0.05, 0.4, 0.25, 0.4]])
    idx, lab = O.matcher(q, [0.3, 0.7], [0, -1, 1])
    assert idx.tolist() == [1, 1, 2, 0]
    assert lab.tolist() == [-1, -1, 0, -1]  # before set_low_quality_matches_ (which flips 1 and 3 to 1)


def test_pairwise_iou_kat_d2t():
    # D2T/structures/test_boxes.py:154-178
    b1 = torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 1.0]])
    b2 = torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 0.5, 1.0], [0.0, 0.0, 1.0, 0.5],
                       [0.0, 0.0, 0.5, 0.5], [0.5, 0.5, 1.0, 1.0], [0.5, 0.5, 1.5, 1.5]])
    exp = torch.tensor([[1.0, 0.5, 0.5, 0.25, 0.25, 0.25 / (2 - 0.25)]] * 2)
    torch.testing.assert_close(O.pairwise_iou(b1, b2), exp)


def test_match_sample_box2box_levels_golden(golden):
    g = golden("roi_match")
    gt, pr = T(g["gt"]), T(g["proposals"])
    iou = O.pairwise_iou(gt, pr)
    assert torch.equal(iou, T(g["iou"]))
    for thr in (0.6, 0.7, 0.8):
        idx, lab = O.matcher(iou, [thr], [0, 1])
        assert torch.equal(idx, T(g["match_idx_%d" % int(thr * 10)]))
        assert torch.equal(lab, T(g["match_lab_%d" % int(thr * 10)]))
    torch.manual_seed(int(g["seed"]))
    pos, neg = O.subsample_labels(T(g["cls"]), 64, 0.25, 20)
    assert torch.equal(pos, T(g["pos_idx"])) and torch.equal(neg, T(g["neg_idx"]))
    w = (10.0, 10.0, 5.0, 5.0)
    d = O.get_deltas(pr[-18:], torch.cat([gt, gt]), w)
    torch.testing.assert_close(d, T(g["deltas"]), atol=0, rtol=0)
    torch.testing.assert_close(O.apply_deltas(T(g["deltas_in"]), pr[-18:], w), T(g["applied"]), atol=0, rtol=0)
    lv = O.assign_boxes_to_levels([pr[:200], pr[200:]], 3, 5)
    assert torch.equal(lv, T(g["levels"]))


def test_crop_and_resize_properties():
    m = torch.zeros(2, 40, 50, dtype=torch.bool)
    m[0, 10:30, 5:25] = True
    m[1] = True
    boxes = torch.tensor([[5.0, 10.0, 25.0, 30.0], [0.0, 0.0, 50.0, 40.0]])
    out = O.crop_and_resize(m, boxes, 28)
    assert out.dtype == torch.bool and out.all()
    out2 = O.crop_and_resize(m, torch.tensor([[0.0, 0.0, 50.0, 40.0], [60.0, 60.0, 70.0, 70.0]]), 28)
    assert 0.15 < out2[0].float().mean() < 0.25 and not out2[1].any()
