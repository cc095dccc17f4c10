"""Generate tests/golden/*.npz by running the REFERENCE's own source files (read-only at
/root/reference) on seeded inputs.  Run in the authoring container only:

    python tests/golden/make_golden.py

Each fixture stores inputs (or the seed recipe that regenerates them) and the outputs the
reference produced.  No reference source text is stored.  The oracle (oracle/*.py) is
checked against these in tests/test_oracle_*.py; the HIP path is checked against both.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

torch.set_num_threads(4)


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print("wrote %-28s %7.1f KB  keys=%d" % (name, os.path.getsize(path) / 1024, len(arrs)))


def fill_params(module, seed, scale=0.05):
    """Deterministic parameter recipe shared with the tests (tests/_recipes.py)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            v = torch.randn(p.shape, generator=g) * scale
            if name.endswith("norm.weight") or ".norm" in name and name.endswith("weight") \
                    or name.startswith("norm") and name.endswith("weight"):
                v = v + 1.0
            p.copy_(v)


def checksum(module):
    return float(sum(p.double().abs().sum() for p in module.parameters()))


# --------------------------------------------------------------------------------------
def gen_swin():
    sw = R.ref("divergen.modeling.backbone.swintransformer")
    # G1: WindowAttention fwd + bwd, window 7 and 12, head_dim 32
    for ws in (7, 12):
        torch.manual_seed(100 + ws)
        N = ws * ws
        attn = sw.WindowAttention(64, (ws, ws), 2)
        fill_params(attn, 7 + ws, 0.1)
        nW, B = 3, 2
        x = torch.randn(nW * B, N, 64, requires_grad=True)
        # 0/-100 mask with the reference's structure: region ids per token
        ids = torch.randint(0, 3, (nW, N)).float()
        mask = ids.unsqueeze(1) - ids.unsqueeze(2)
        mask = mask.masked_fill(mask != 0, -100.0).masked_fill(mask == 0, 0.0)
        mask[0] = 0
        out = attn(x, mask)
        g = torch.randn_like(out)
        out.backward(g)
        out_nomask = attn(x.detach(), None)
        save("swin_attn_w%d" % ws,
             x=x, mask=mask, g=g, out=out, out_nomask=out_nomask, dx=x.grad,
             qkv_w=attn.qkv.weight, qkv_b=attn.qkv.bias, proj_w=attn.proj.weight,
             proj_b=attn.proj.bias, table=attn.relative_position_bias_table,
             index=attn.relative_position_index,
             d_qkv_w=attn.qkv.weight.grad, d_qkv_b=attn.qkv.bias.grad,
             d_proj_w=attn.proj.weight.grad, d_proj_b=attn.proj.bias.grad,
             d_table=attn.relative_position_bias_table.grad)

    # G2: BasicLayer (W-MSA + SW-MSA blocks, padding, PatchMerging with odd W)
    for ws, H, W in ((7, 10, 13), (12, 14, 25)):
        torch.manual_seed(200 + ws)
        layer = sw.BasicLayer(dim=64, depth=2, num_heads=2, window_size=ws,
                              drop_path=0.0, downsample=sw.PatchMerging)
        fill_params(layer, 21 + ws, 0.08)
        x = torch.randn(2, H * W, 64, requires_grad=True)
        x_out, h, w, x_down, wh, ww = layer(x, H, W)
        g1 = torch.randn_like(x_out)
        g2 = torch.randn_like(x_down)
        (x_out * g1).sum().add((x_down * g2).sum()).backward()
        sd = {("p." + k): v for k, v in layer.state_dict().items()}
        gr = {("g." + k): p.grad for k, p in layer.named_parameters()}
        save("swin_layer_w%d" % ws, x=x, H=H, W=W, x_out=x_out, x_down=x_down, Wh=wh, Ww=ww,
             g1=g1, g2=g2, dx=x.grad, **sd, **gr)

    # G3: whole backbone, params by recipe
    torch.manual_seed(300)
    net = sw.SwinTransformer(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8],
                             window_size=7, drop_path_rate=0.0, out_indices=(1, 2, 3))
    fill_params(net, 31, 0.05)
    img = torch.randn(1, 3, 75, 110)
    outs = net(img)
    save("swin_full", img=img, param_seed=31, param_scale=0.05, param_checksum=checksum(net),
         **{k: v for k, v in outs.items()})

    # drop path (timm algorithm) under a fixed torch seed
    blk = sw.SwinTransformerBlock(64, 2, window_size=7, shift_size=0, drop_path=0.3)
    fill_params(blk, 41, 0.08)
    blk.train()
    blk.H, blk.W = 7, 7
    x = torch.randn(8, 49, 64)
    torch.manual_seed(4242)
    y = blk(x, None)
    save("swin_droppath", x=x, y=y, torch_seed=4242, rate=0.3,
         **{("p." + k): v for k, v in blk.state_dict().items()})


# --------------------------------------------------------------------------------------
def _make_gt(n, h, w, ncls, gen, Instances, Boxes):
    x1 = torch.rand(n, generator=gen) * (w - 40)
    y1 = torch.rand(n, generator=gen) * (h - 40)
    bw = 8 + torch.rand(n, generator=gen) * (w * 0.6)
    bh = 8 + torch.rand(n, generator=gen) * (h * 0.6)
    boxes = torch.stack([x1, y1, (x1 + bw).clamp(max=w), (y1 + bh).clamp(max=h)], 1)
    inst = Instances((h, w))
    inst.gt_boxes = Boxes(boxes)
    inst.gt_classes = torch.randint(0, ncls, (n,), generator=gen)
    return inst


def gen_centernet():
    cn = R.ref("centernet.modeling.dense_heads.centernet")
    st = sys.modules["detectron2.structures"]
    gen = torch.Generator().manual_seed(500)
    net = cn.CenterNet(in_channels=16, num_classes=7, with_agn_hm=True, only_proposal=True,
                       score_thresh=0.0001, reg_weight=1.0, not_norm_reg=True,
                       pos_weight=0.5, neg_weight=0.5, ignore_high_fp=0.85,
                       pre_nms_topk_train=60, post_nms_topk_train=40,
                       nms_thresh_train=0.9, nms_thresh_test=0.9,
                       centernet_head=nn.Identity())
    net.train()
    H, W = 256, 320
    strides = (8, 16, 32, 64, 128)
    feats = [torch.zeros(2, 16, -(-H // s), -(-W // s)) for s in strides]
    grids = net.compute_grids(feats)
    shapes = grids[0].new_tensor([(f.shape[2], f.shape[3]) for f in feats])
    gts = [_make_gt(6, H, W, 7, gen, st.Instances, st.Boxes),
           _make_gt(0, H, W, 7, gen, st.Instances, st.Boxes),
           ]
    # a third image reuses image 0 shapes: B must equal feature batch (2)
    pos_inds, labels, reg_targets, hms = net._get_ground_truth(grids, shapes, gts)
    gts2 = [_make_gt(9, H, W, 7, gen, st.Instances, st.Boxes),
            _make_gt(3, H, W, 7, gen, st.Instances, st.Boxes)]
    # include a tiny box and a huge one to hit level-range edges
    gts2[1].gt_boxes.tensor[0] = torch.tensor([10.0, 12.0, 14.5, 15.0])
    gts2[1].gt_boxes.tensor[1] = torch.tensor([0.0, 0.0, 320.0, 256.0])
    pos2, lab2, reg2, hm2 = net._get_ground_truth(grids, shapes, gts2)
    M = sum(g.shape[0] for g in grids) * 2
    reg_pred = (torch.rand(M, 4, generator=gen) * 6).requires_grad_(True)
    agn_logit = (torch.randn(M, generator=gen) * 2 - 2).requires_grad_(True)
    losses = net.losses(pos2, lab2, reg2, hm2, None, reg_pred, agn_logit.clone())
    total = sum(losses.values())
    total.backward()
    save("centernet_targets",
         H=H, W=W, strides=np.array(strides),
         gt0_boxes=gts[0].gt_boxes.tensor, gt0_classes=gts[0].gt_classes,
         pos_inds=pos_inds, labels=labels, reg_targets=reg_targets, hms=hms,
         gt2a_boxes=gts2[0].gt_boxes.tensor, gt2a_classes=gts2[0].gt_classes,
         gt2b_boxes=gts2[1].gt_boxes.tensor, gt2b_classes=gts2[1].gt_classes,
         pos2=pos2, lab2=lab2, reg2=reg2, hm2=hm2,
         reg_pred=reg_pred, agn_logit=agn_logit,
         loss_loc=losses["loss_centernet_loc"], loss_pos=losses["loss_centernet_agn_pos"],
         loss_neg=losses["loss_centernet_agn_neg"],
         d_reg_pred=reg_pred.grad, d_agn_logit=agn_logit.grad)

    # predict_single_level (pre-NMS candidates): one level with > topk candidates
    hm = torch.rand(2, 1, 16, 20, generator=gen)
    hm[1] *= 1e-5  # second image: nothing above threshold except a few
    hm[1, 0, 3, 4] = 0.5
    reg = torch.rand(2, 4, 16, 20, generator=gen) * 5
    res = net.predict_single_level(grids[1], hm, reg * 16, [(H, W), (H, W)], None, 1)
    # topk(sorted=False) order is an implementation detail: store sorted by score
    out = {}
    for i, r in enumerate(res):
        o = torch.argsort(r.scores, descending=True, stable=True)
        out["boxes%d" % i] = r.pred_boxes.tensor[o]
        out["scores%d" % i] = r.scores[o]
    save("centernet_predict", hm=hm, reg=reg, stride=16, grids=grids[1], topk=60, thresh=0.0001, **out)

    il = R.ref("centernet.modeling.layers.iou_loss")
    p = torch.rand(50, 4, generator=gen) * 10
    t = torch.rand(50, 4, generator=gen) * 10
    w = torch.rand(50, generator=gen)
    save("iou_loss", pred=p, target=t, weight=w,
         giou_none=il.IOULoss("giou")(p, t, None, reduction="none"),
         giou_sum_w=il.IOULoss("giou")(p, t, w, reduction="sum"))


# --------------------------------------------------------------------------------------
def gen_roi():
    st = sys.modules["detectron2.structures"]
    Boxes, pairwise_iou = st.Boxes, st.pairwise_iou
    mt = R.ref("detectron2.modeling.matcher")
    sm = R.ref("detectron2.modeling.sampling")
    br = R.ref("detectron2.modeling.box_regression")
    pl = R.ref("detectron2.modeling.poolers")
    ut = R.ref("divergen.modeling.utils")
    fr = R.ref("divergen.modeling.roi_heads.detic_fast_rcnn")
    gen = torch.Generator().manual_seed(700)

    def rboxes(n, s=300.0):
        a = torch.rand(n, 2, generator=gen) * s
        wh = 4 + torch.rand(n, 2, generator=gen) * s * 0.5
        return torch.cat([a, a + wh], 1)

    gt = rboxes(9)
    pr = torch.cat([rboxes(400), gt + torch.randn(9, 4, generator=gen) * 3, gt])
    iou = pairwise_iou(Boxes(gt), Boxes(pr))
    outs = {}
    for thr in (0.6, 0.7, 0.8):
        m = mt.Matcher([thr], [0, 1], allow_low_quality_matches=False)
        idx, lab = m(iou)
        outs["match_idx_%d" % int(thr * 10)] = idx
        outs["match_lab_%d" % int(thr * 10)] = lab
    # subsample_labels under a fixed torch seed (randperm stream)
    gt_classes = torch.randint(0, 20, (9,), generator=gen)
    lab6 = outs["match_lab_6"]
    cls = gt_classes[outs["match_idx_6"]].clone()
    cls[lab6 == 0] = 20
    torch.manual_seed(777)
    pos_idx, neg_idx = sm.subsample_labels(cls, 64, 0.25, 20)
    # box2box
    tr = br.Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))
    src = pr[-18:]
    tgt = torch.cat([gt, gt])
    deltas = tr.get_deltas(src, tgt)
    big = deltas.clone()
    big[0, 2] = 50.0  # exercise scale clamp
    applied = tr.apply_deltas(big, src)
    # level assignment
    lv = pl.assign_boxes_to_levels([Boxes(pr[:200]), Boxes(pr[200:])], 3, 5, 224, 4)
    save("roi_match", gt=gt, proposals=pr, iou=iou, gt_classes=gt_classes, cls=cls,
         seed=777, pos_idx=pos_idx, neg_idx=neg_idx, deltas=deltas, deltas_in=big,
         applied=applied, levels=lv, **outs)

    # federated loss class sampling + sigmoid CE + box reg loss
    C = 40
    freq = (torch.rand(C, generator=gen) * 1000 + 1).float() ** 0.5
    R_ = 96
    logits = (torch.randn(R_, C + 1, generator=gen) * 2).requires_grad_(True)
    gtc = torch.randint(0, C + 1, (R_,), generator=gen)
    gtc[:40] = C  # plenty of background
    fake = types.SimpleNamespace(use_fed_loss=True, freq_weight=freq, fed_loss_num_cat=10,
                                 ignore_zero_cats=False, num_classes=C,
                                 box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0,
                                 box2box_transform=tr)
    torch.manual_seed(888)
    appeared = ut.get_fed_loss_inds(gtc, 10, C, freq)
    torch.manual_seed(888)
    loss_cls = fr.DeticFastRCNNOutputLayers.sigmoid_cross_entropy_loss(fake, logits, gtc)
    loss_cls.backward()
    pb = rboxes(R_)
    gb = pb + torch.randn(R_, 4, generator=gen) * 4
    gb[:, 2:] = torch.max(gb[:, 2:], gb[:, :2] + 1)
    pd = (torch.randn(R_, 4, generator=gen)).requires_grad_(True)
    loss_box = fr.DeticFastRCNNOutputLayers.box_reg_loss(fake, pb, gb, pd, gtc, None, num_classes=C)
    loss_box.backward()
    save("roi_losses", freq=freq, logits=logits, gt_classes=gtc, seed=888, appeared=appeared,
         loss_cls=loss_cls, d_logits=logits.grad, prop_boxes=pb, gt_boxes=gb, pred_deltas=pd,
         loss_box=loss_box, d_pred_deltas=pd.grad, weights=np.array([10.0, 10.0, 5.0, 5.0]))


# --------------------------------------------------------------------------------------
def gen_compositor():
    mp = R.ref("divergen.data.custom_build_copypaste_mapper")
    rng = np.random.default_rng(7)
    H, W = 96, 128

    def ellipse_mask(x0, y0, x1, y1):
        yy, xx = np.mgrid[0:H, 0:W]
        cx, cy, rx, ry = (x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0) / 2, (y1 - y0) / 2
        return ((((xx - cx) / max(rx, 1)) ** 2 + ((yy - cy) / max(ry, 1)) ** 2) <= 1).astype(np.uint8)

    img = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
    n = 5
    masks = []
    for i in range(n):
        x0, y0 = rng.integers(0, W - 30), rng.integers(0, H - 30)
        masks.append(ellipse_mask(x0, y0, x0 + rng.integers(6, 50), y0 + rng.integers(6, 40)))
    masks = np.stack(masks)
    dst = {"image": img.copy(), "gt_masks": masks.copy(), "gt_bboxes": mp.get_bboxes(masks),
           "gt_labels": rng.integers(0, 100, n).astype(np.int64),
           "instance_source": np.zeros(n, dtype=np.int64)}
    fake = types.SimpleNamespace(bbox_occluded_thr=10, mask_occluded_thr=300, cp_method=["basic"])
    K = 7
    store = {"dst_image": img, "dst_masks": masks, "dst_boxes": dst["gt_bboxes"],
             "dst_labels": dst["gt_labels"], "K": K}
    for k in range(K):
        h, w = int(rng.integers(8, 60)), int(rng.integers(8, 70))
        rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        m = ((((xx - w / 2) / (w / 2)) ** 2 + ((yy - h / 2) / (h / 2)) ** 2) <= 1).astype(np.uint8)
        rgba[..., 3] *= m
        x0, y0 = int(rng.integers(-w // 2, W - w // 2)), int(rng.integers(-h // 2, H - h // 2))
        # what pad_to_hw's integer-translate warpAffine produces: shifted copy, zero border
        canvas = np.zeros((4, H, W), np.uint8)
        cm = np.zeros((1, H, W), np.uint8)
        ys, xs = max(y0, 0), max(x0, 0)
        ye, xe = min(y0 + h, H), min(x0 + w, W)
        canvas[:, ys:ye, xs:xe] = rgba[ys - y0:ye - y0, xs - x0:xe - x0].transpose(2, 0, 1)
        cm[0, ys:ye, xs:xe] = (rgba[ys - y0:ye - y0, xs - x0:xe - x0, 3] > 0)
        src = {"image": canvas, "gt_masks": cm, "gt_bboxes": mp.get_bboxes(cm),
               "gt_labels": np.array([1000 + k], dtype=np.int64)}
        store["src%d_rgba" % k] = rgba
        store["src%d_xy" % k] = np.array([x0, y0])
        store["src%d_label" % k] = src["gt_labels"]
        dst = mp.InstPool._copy_paste(fake, dst, src)
    store.update(out_image=dst["image"], out_masks=dst["gt_masks"], out_boxes=dst["gt_bboxes"],
                 out_labels=dst["gt_labels"], out_source=dst["instance_source"])
    save("compositor", **store)


# --------------------------------------------------------------------------------------
def gen_solver():
    ema_m = R.ref("divergen.ema")
    lr = R.ref("detectron2.solver.lr_scheduler")
    torch.manual_seed(900)
    net = nn.Sequential(nn.Linear(6, 5), nn.LayerNorm(5), nn.Linear(5, 3))
    ema = ema_m.ModelEma(net, 0.999)
    opt = torch.optim.AdamW([{"params": [p], "lr": 1e-2} for p in net.parameters()], 1e-2,
                            weight_decay=1e-4)
    sched = lr.WarmupCosineLR(opt, 100, warmup_factor=1e-4, warmup_iters=10)
    lrs = []
    x = torch.randn(16, 6)
    init = {("init." + k): v.clone() for k, v in net.state_dict().items()}
    for it in range(12):
        loss = (net(x) ** 2).sum() * 30
        ema.update(net)
        opt.zero_grad()
        loss.backward()
        for p in net.parameters():
            torch.nn.utils.clip_grad_value_(p, 1.0)
        opt.step()
        lrs.append(opt.param_groups[0]["lr"])
        sched.step()
    save("solver", x=x, lrs=np.array(lrs), **init,
         **{("final." + k): v for k, v in net.state_dict().items()},
         **{("ema." + k): v for k, v in ema.state_dict().items()})


def gen_heads():
    fpn_m = sys.modules["detectron2.modeling.backbone.fpn"]
    f5 = R.ref("centernet.modeling.backbone.fpn_p5")
    ch = R.ref("centernet.modeling.dense_heads.centernet_head")
    ss = sys.modules["detectron2.layers"].ShapeSpec
    Backbone = sys.modules["detectron2.modeling.backbone"].Backbone

    class Dummy(Backbone):
        _out_features = ["swin1", "swin2", "swin3"]
        _out_feature_channels = {"swin1": 8, "swin2": 16, "swin3": 32}
        _out_feature_strides = {"swin1": 8, "swin2": 16, "swin3": 32}

        def forward(self, x):
            return x

    torch.manual_seed(1000)
    fpn = fpn_m.FPN(Dummy(), ["swin1", "swin2", "swin3"], 16, norm="",
                    top_block=f5.LastLevelP6P7_P5(16, 16), fuse_type="sum")
    fill_params(fpn, 51, 0.1)
    feats = {"swin1": torch.randn(2, 8, 16, 24), "swin2": torch.randn(2, 16, 8, 12),
             "swin3": torch.randn(2, 32, 4, 6)}
    out = fpn(feats)
    save("fpn", **{("in." + k): v for k, v in feats.items()},
         **{("out." + k): v for k, v in out.items()},
         **{("p." + k): v for k, v in fpn.state_dict().items()})

    head = ch.CenterNetHead(in_channels=32, num_levels=2, num_classes=5, with_agn_hm=True,
                            only_proposal=True, norm="GN", num_cls_convs=4, num_box_convs=4,
                            num_share_convs=0, use_deformable=False, prior_prob=0.01)
    fill_params(head, 61, 0.05)
    xs = [torch.randn(2, 32, 12, 10), torch.randn(2, 32, 6, 5)]
    clss, regs, hms = head(xs)
    save("centernet_head", x0=xs[0], x1=xs[1], reg0=regs[0], reg1=regs[1], hm0=hms[0], hm1=hms[1],
         **{("p." + k): v for k, v in head.state_dict().items()})


def fill_by_name(module, seed, scale):
    """Parameters from a per-NAME seeded stream, rounded to bf16 (so the product's bf16 shadow weights ARE these weights):
    independent of registration order; mirrored by tests/_recipes.py:fill_by_name."""
    import zlib
    with torch.no_grad():
        for name, p in module.named_parameters():
            g = torch.Generator().manual_seed(seed + (zlib.crc32(name.encode()) & 0xFFFFFF))
            v = torch.randn(p.shape, generator=g) * scale
            if p.dim() == 1 and name.endswith("weight"):          # norm scales
                v = v + 1.0
            p.copy_(v.to(torch.bfloat16).float())


def bf16r(t):
    return t.to(torch.bfloat16).float()


def gen_heads_wide():
    """FPN, CenterNetHead, mask head and box head of the reference at the widths the product's HIP path is built for
    (256 channels: implicit-GEMM convolutions, 8-channel GroupNorm groups), fp32 on bf16-exact weights and inputs: values and
    gradients.  Fixtures hold inputs / outputs only; the parameters are regenerated from their names (fill_by_name)."""
    fpn_m = sys.modules.get("detectron2.modeling.backbone.fpn") or R.ref("detectron2.modeling.backbone.fpn")
    f5 = R.ref("centernet.modeling.backbone.fpn_p5")
    ch = R.ref("centernet.modeling.dense_heads.centernet_head")
    mh = R.ref("detectron2.modeling.roi_heads.mask_head")
    bh = R.ref("detectron2.modeling.roi_heads.box_head")
    ss = sys.modules["detectron2.layers"].ShapeSpec
    Backbone = sys.modules["detectron2.modeling.backbone"].Backbone

    class Dummy(Backbone):
        _out_features = ["swin1", "swin2", "swin3"]
        _out_feature_channels = {"swin1": 64, "swin2": 128, "swin3": 256}
        _out_feature_strides = {"swin1": 8, "swin2": 16, "swin3": 32}

        def forward(self, x):
            return x

    torch.manual_seed(2000)
    fpn = fpn_m.FPN(Dummy(), ["swin1", "swin2", "swin3"], 256, norm="", top_block=f5.LastLevelP6P7_P5(256, 256), fuse_type="sum")
    fill_by_name(fpn, 71, 0.03)
    feats = {"swin1": bf16r(torch.randn(1, 64, 16, 12)).requires_grad_(True), "swin2": bf16r(torch.randn(1, 128, 8, 6)).requires_grad_(True),
             "swin3": bf16r(torch.randn(1, 256, 4, 3)).requires_grad_(True)}
    out = fpn(feats)
    gos = {k: bf16r(torch.randn(v.shape)) for k, v in out.items()}
    sum((out[k] * gos[k]).sum() for k in out).backward()
    sd = dict(fpn.named_parameters())
    save("fpn_wide", **{("in." + k): v for k, v in feats.items()}, **{("out." + k): v for k, v in out.items()},
         **{("go." + k): v for k, v in gos.items()}, **{("din." + k): v.grad for k, v in feats.items()},
         **{"g.fpn_lateral5.weight": sd["fpn_lateral5.weight"].grad, "g.fpn_output3.bias": sd["fpn_output3.bias"].grad,
            "g.fpn_output4.weight.rows8": sd["fpn_output4.weight"].grad[:8], "g.top_block.p6.weight.rows8": sd["top_block.p6.weight"].grad[:8]})

    head = ch.CenterNetHead(in_channels=256, num_levels=2, num_classes=5, with_agn_hm=True, only_proposal=True, norm="GN",
                            num_cls_convs=4, num_box_convs=4, num_share_convs=0, use_deformable=False, prior_prob=0.01)
    fill_by_name(head, 72, 0.02)
    with torch.no_grad():
        head.bbox_pred.bias.fill_(2.0)          # keep the ReLU of the regression maps away from zero, as the trained bias (8.0) does
    xs = [bf16r(torch.randn(1, 256, 12, 10)).requires_grad_(True), bf16r(torch.randn(1, 256, 6, 5)).requires_grad_(True)]
    clss, regs, hms = head(xs)
    gr = [bf16r(torch.randn(r.shape)) for r in regs]
    gh = [bf16r(torch.randn(h.shape)) for h in hms]
    (sum((r * g).sum() for r, g in zip(regs, gr)) + sum((h * g).sum() for h, g in zip(hms, gh))).backward()
    hp = dict(head.named_parameters())
    save("centernet_head_wide", x0=xs[0], x1=xs[1], reg0=regs[0], reg1=regs[1], hm0=hms[0], hm1=hms[1], gr0=gr[0], gr1=gr[1],
         gh0=gh[0], gh1=gh[1], dx0=xs[0].grad, dx1=xs[1].grad,
         **{"g.bbox_tower.0.weight.rows8": hp["bbox_tower.0.weight"].grad[:8], "g.bbox_tower.1.weight": hp["bbox_tower.1.weight"].grad,
            "g.bbox_tower.1.bias": hp["bbox_tower.1.bias"].grad, "g.bbox_pred.weight": hp["bbox_pred.weight"].grad,
            "g.agn_hm.bias": hp["agn_hm.bias"].grad, "g.scales.1.scale": hp["scales.1.scale"].grad})

    mask = mh.MaskRCNNConvUpsampleHead(ss(channels=256, height=14, width=14), num_classes=1, conv_dims=[256] * 5, conv_norm="")
    fill_by_name(mask, 73, 0.02)
    xm = bf16r(torch.randn(3, 256, 14, 14)).requires_grad_(True)
    logits = mask.layers(xm)
    gm = bf16r(torch.randn(logits.shape))
    (logits * gm).sum().backward()
    mp = dict(mask.named_parameters())
    save("mask_head_wide", x=xm, logits=logits, go=gm, dx=xm.grad,
         **{"g.mask_fcn1.weight.rows8": mp["mask_fcn1.weight"].grad[:8], "g.deconv.weight.rows8": mp["deconv.weight"].grad[:8],
            "g.deconv.bias": mp["deconv.bias"].grad, "g.predictor.weight": mp["predictor.weight"].grad, "g.predictor.bias": mp["predictor.bias"].grad})

    box = bh.FastRCNNConvFCHead(ss(channels=256, height=7, width=7), conv_dims=[], fc_dims=[1024, 1024])
    fill_by_name(box, 74, 0.01)
    xb = bf16r(torch.randn(8, 256, 7, 7)).requires_grad_(True)
    yb = box(xb)
    gb = bf16r(torch.randn(yb.shape))
    (yb * gb).sum().backward()
    bp = dict(box.named_parameters())
    save("box_head_wide", x=xb, y=yb, go=gb, dx=xb.grad,
         **{"g.fc1.weight.rows4": bp["fc1.weight"].grad[:4], "g.fc2.weight.rows16": bp["fc2.weight"].grad[:16], "g.fc2.bias": bp["fc2.bias"].grad})


def gen_postprocess():
    """paste_masks_in_image (D2/layers/mask_ops.py:73) on seeded detections; output stored bit-packed."""
    mo = R.ref("detectron2.layers.mask_ops")
    g = torch.Generator().manual_seed(77)
    H, W, S, N = 97, 131, 28, 24
    # smooth blobs like sigmoid mask logits: low-res noise upsampled
    low = torch.rand(N, 1, 5, 5, generator=g)
    masks = torch.nn.functional.interpolate(low, size=(S, S), mode="bicubic", align_corners=False)[:, 0].clamp(0, 1)
    cx, cy = torch.rand(N, generator=g) * W, torch.rand(N, generator=g) * H
    bw, bh = torch.rand(N, generator=g) * 80 + 0.7, torch.rand(N, generator=g) * 60 + 0.7
    boxes = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
    boxes[0] = torch.tensor([0., 0., W, H])                      # whole image
    boxes[1] = torch.tensor([-20.5, -10.25, 40.5, 30.0])          # partly outside (un-clipped boxes are legal inputs)
    boxes[2] = torch.tensor([100.0, 50.0, 160.0, 120.0])
    boxes[3] = torch.tensor([10.0, 10.0, 10.9, 10.9])             # sub-pixel box
    masks[4] = 1.0                                                # solid
    masks[5] = 0.0                                                # empty
    out = mo.paste_masks_in_image(masks, boxes, (H, W), 0.5)
    assert out.dtype == torch.bool and tuple(out.shape) == (N, H, W)
    save("paste_masks", masks=masks, boxes=boxes, image_shape=np.array([H, W]),
         out_bits=np.packbits(out.numpy().reshape(N, -1), axis=1), threshold=np.float32(0.5))


def gen_inference():
    """fast_rcnn_inference_single_image (D2/modeling/roi_heads/fast_rcnn.py:117-170) on seeded box / score tensors.  The file is
    torch-only except for `batched_nms` (torchvision, not vendored): the reference function runs with the oracle's per-class
    greedy NMS in that slot (oracle/roi.py, itself cross-checked against the reference's nms_rotated_cpu.cpp), so what this
    fixture pins is everything around it -- the finite-row filter, clipping, score threshold, the (row, class) index pairs,
    per-class suppression, top-k, output order."""
    fr = R.ref("detectron2.modeling.roi_heads.fast_rcnn")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import roi as OR
    fr.batched_nms = OR.batched_nms
    g = torch.Generator().manual_seed(4242)
    store = {}
    # (name, R proposals, K classes, class-specific regression?, image (h, w), score threshold, NMS threshold, top-k)
    cases = [("agnostic", 400, 23, False, (333, 500), 0.02, 0.5, 100), ("perclass", 160, 7, True, (240, 200), 0.05, 0.6, 50),
             ("nonfinite", 120, 11, False, (128, 160), 0.0001, 0.5, 300), ("empty", 40, 5, False, (64, 64), 1.5, 0.5, 10),
             ("swinL", 256, 1203, False, (800, 1216), 0.0001, 0.5, 300)]
    for name, Rn, K, per_class, (h, w), st, nt, topk in cases:
        # clustered boxes (so that suppression happens), some reaching outside the image (so that clipping matters)
        ctr = torch.rand(Rn // 8 + 1, 2, generator=g) * torch.tensor([w, h])
        c = ctr[torch.randint(0, len(ctr), (Rn,), generator=g)] + torch.randn(Rn, 2, generator=g) * 6.0
        wh = (torch.rand(Rn, 2, generator=g) * 0.5 + 0.08) * torch.tensor([w, h])
        nreg = K if per_class else 1
        base = torch.cat([c - wh / 2, c + wh / 2], 1)
        boxes = (base[:, None, :] + torch.randn(Rn, nreg, 4, generator=g) * 3.0).reshape(Rn, nreg * 4)
        if name == "swinL":                                    # sigmoid scores of a federated classifier: most classes near 0
            scores = torch.sigmoid(torch.randn(Rn, K + 1, generator=g) * 2.5 - 6.0)
        else:
            scores = torch.rand(Rn, K + 1, generator=g) ** 3
        if name == "nonfinite":
            boxes[5, 2] = float("inf"); boxes[17, 0] = float("nan"); scores[33, 4] = float("nan")
        res, kept = fr.fast_rcnn_inference_single_image(boxes.clone(), scores.clone(), (h, w), st, nt, topk)
        store[name + "_boxes"], store[name + "_scores"] = boxes, scores
        store[name + "_cfg"] = np.array([h, w, st, nt, topk], dtype=np.float64)
        store[name + "_out_boxes"], store[name + "_out_scores"] = res.pred_boxes.tensor, res.scores
        store[name + "_out_classes"], store[name + "_out_rows"] = res.pred_classes, kept
        print("  %-10s %4d x %4d classes -> %3d detections" % (name, Rn, K, len(kept)))
    save("fast_rcnn_inference", **store)


def gen_pool():
    """Instance-pool decode: PNG fixtures under tests/golden/pool/ + what the reference's
    InstPool._load_RGBA (mapper.py:359-444) hands to cv2.resize for each key (array and target size), captured by
    replacing the stubbed cv2.resize with a recorder.  Keys are stored relative to tests/golden/."""
    from PIL import Image
    mp = R.ref("divergen.data.custom_build_copypaste_mapper")
    pdir = os.path.join(HERE, "pool")
    os.makedirs(pdir, exist_ok=True)
    rng = np.random.default_rng(21)
    names = []
    for i, (h, w) in enumerate([(40, 56), (33, 21), (64, 64), (17, 90), (48, 30), (6, 7)]):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        soft = 255 * np.clip(1.3 - (((xx - w / 2) / (w / 2.4)) ** 2 + ((yy - h / 2) / (h / 2.4)) ** 2), 0, 1)
        alpha = soft.astype(np.uint8)
        if i % 2 == 0:          # "path|maskpath" pair: RGB png + single-channel mask png
            Image.fromarray(rgb, "RGB").save(os.path.join(pdir, "inst%d.png" % i))
            Image.fromarray(alpha, "L").save(os.path.join(pdir, "inst%d_mask.png" % i))
            names.append("pool/inst%d.png|pool/inst%d_mask.png" % (i, i))
        elif i == 1:            # plain RGBA png
            Image.fromarray(np.dstack([rgb, alpha]), "RGBA").save(os.path.join(pdir, "inst%d.png" % i))
            names.append("pool/inst%d.png" % i)
        else:                   # '*' = pre-processed RGBA
            Image.fromarray(np.dstack([rgb, alpha]), "RGBA").save(os.path.join(pdir, "inst%d.png" % i))
            names.append("*pool/inst%d.png" % i)

    class _Stop(Exception):
        pass
    captured = {}

    def recorder(img, size):
        captured["img"], captured["size"] = img.copy(), tuple(int(v) for v in size)
        raise _Stop()
    mp.cv2.resize = recorder
    cwd = os.getcwd()
    os.chdir(HERE)
    store = {"keys": np.array(names), "labels": np.arange(len(names)) + 3, "train_hw": np.array([256, 320])}
    try:
        for mode in ("random_scale", "area_prior"):
            fake = types.SimpleNamespace(
                data_to_cat={k: int(3 + i) for i, k in enumerate(names)},
                HWms={} if mode == "random_scale" else {str(4 + i): [0.12, 0.03] for i in range(len(names))},
                random_scale=(mode == "random_scale"), random_scale_min=0.5, random_scale_max=2.0, random_scale_min_size=5,
                scale_min=10, scale_max=0.5, mask_threshold=128, use_largest_part=False, instance_filter_min=0.01,
                instance_filter_max=1.0, shape_jitter=0.2)
            for i, k in enumerate(names):
                np.random.seed(1000 + i)
                captured.clear()
                try:
                    r = mp.InstPool._load_RGBA(fake, k, (256, 320))
                    assert r is None
                    store["%s_%d_rejected" % (mode, i)] = np.array(1)
                except _Stop:
                    store["%s_%d_img" % (mode, i)] = captured["img"]
                    store["%s_%d_size" % (mode, i)] = np.array(captured["size"])
    finally:
        os.chdir(cwd)
    save("pool_decode", **store)


def _install_fvcore_transforms():
    """fvcore.transforms.transform is third-party and absent: the four members the reference's D2 augmentation files use here,
    restated from fvcore's published semantics (Transform.apply_box through the four corners; HFlipTransform flips axis 1 and
    maps x -> width - x; NoOpTransform; TransformList).  Everything else they import is a bare name (never called)."""
    class Transform:
        def _set_attributes(self, params=None):
            if params:
                for k, v in params.items():
                    if k != "self" and not k.startswith("_"):
                        setattr(self, k, v)

        def apply_box(self, box):
            idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
            coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
            coords = self.apply_coords(coords).reshape((-1, 4, 2))
            return np.concatenate((coords.min(axis=1), coords.max(axis=1)), axis=1)

        def apply_segmentation(self, segmentation):
            return self.apply_image(segmentation)

        @classmethod
        def register_type(cls, data_type, func):          # D2's transform.py registers rotated-box handlers at import
            setattr(cls, "apply_" + data_type, func)

    class HFlipTransform(Transform):
        def __init__(self, width):
            self.width = width

        def apply_image(self, img):
            return np.flip(img, axis=1) if img.ndim <= 3 else np.flip(img, axis=-2)

        def apply_coords(self, coords):
            coords[:, 0] = self.width - coords[:, 0]
            return coords

    class NoOpTransform(Transform):
        def apply_image(self, img):
            return img

        def apply_coords(self, coords):
            return coords

        def apply_segmentation(self, seg):
            return seg

    class TransformList(Transform):
        def __init__(self, transforms):
            self.transforms = list(transforms)

    ft = types.ModuleType("fvcore.transforms.transform")
    for nm in ("BlendTransform", "CropTransform", "PadTransform", "VFlipTransform"):
        setattr(ft, nm, type(nm, (Transform,), {}))
    ft.Transform, ft.HFlipTransform, ft.NoOpTransform, ft.TransformList = Transform, HFlipTransform, NoOpTransform, TransformList
    pkg = types.ModuleType("fvcore.transforms")
    pkg.transform = ft
    sys.modules["fvcore.transforms"] = pkg
    sys.modules["fvcore.transforms.transform"] = ft


def gen_pool_draws():
    """The reference's InstPool.get_mix_result('cas_random') END TO END (mapper.py:213-261 -> _get_cls_balanced_random_samples
    :263-296 -> _cat_a_new_image :488-507 -> _load_RGBA :359-456 -> random_start_xy :45-66 -> _copy_paste :510-566) on the PNG
    fixtures under tests/golden/pool/, from seeded np.random streams, with the reference's own D2 RandomFlip / AugInput /
    AugmentationList (D2/data/transforms/augmentation*.py run from their files).  Recorded: the order and values of every draw
    that reaches a side effect (key, target size, flip, placement) and the final image / boxes / classes / masks /
    instance_source.  cv2 is absent: cv2.resize is a PIL-bilinear stand-in (the build's own `_resize`, so the PIXELS of a resized
    patch are not pinned by this file, their position / size / order are), cv2.warpAffine with the integer translation
    pad_to_hw builds is a shifted copy with a zero border."""
    import importlib.util
    from PIL import Image
    mp = R.ref("divergen.data.custom_build_copypaste_mapper")
    _install_fvcore_transforms()
    if not hasattr(Image, "LINEAR"):          # removed from Pillow 10; D2's transform.py names it in a default argument
        Image.LINEAR = Image.BILINEAR
    for nm in ("detectron2.data.transforms", "detectron2.data.transforms.augmentation", "detectron2.data.transforms.transform",
               "detectron2.data.transforms.augmentation_impl"):
        sys.modules.pop(nm, None)
    pkg = types.ModuleType("detectron2.data.transforms")
    pkg.__path__ = [R.D2 + "/data/transforms"]
    sys.modules["detectron2.data.transforms"] = pkg
    for leaf in ("augmentation", "transform", "augmentation_impl"):
        spec = importlib.util.spec_from_file_location("detectron2.data.transforms." + leaf, R.D2 + "/data/transforms/%s.py" % leaf)
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        setattr(pkg, leaf, m)
    T = types.SimpleNamespace(AugInput=pkg.augmentation.AugInput, AugmentationList=pkg.augmentation.AugmentationList,
                              RandomFlip=pkg.augmentation_impl.RandomFlip)
    mp.T = T
    log = []

    def resize(img, size):
        tw, th = int(size[0]), int(size[1])
        log.append(("resize", img.shape[0], img.shape[1], tw, th))
        return np.array(Image.fromarray(np.ascontiguousarray(img), "RGBA").resize((tw, th), Image.BILINEAR))

    def warp_affine(data, M, wh):
        w, h = wh
        x0, y0 = int(M[0][2]), int(M[1][2])
        assert float(M[0][2]) == x0 and float(M[1][2]) == y0 and M[0][0] == 1 and M[1][1] == 1
        src = data if data.ndim == 3 else data[:, :, None]
        out = np.zeros((h, w, src.shape[2]), dtype=data.dtype)
        sh, sw = src.shape[:2]
        ys, xs, ye, xe = max(y0, 0), max(x0, 0), min(y0 + sh, h), min(x0 + sw, w)
        if ye > ys and xe > xs:
            out[ys:ye, xs:xe] = src[ys - y0:ye - y0, xs - x0:xe - x0]
        return out if data.ndim == 3 else out[:, :, 0]
    mp.cv2.resize, mp.cv2.warpAffine = resize, warp_affine
    real_start_xy = mp.start_xy

    def start_xy(data_dict, bb, train_size):
        log.append(("place", int(bb[0]), int(bb[1]), int(data_dict["gt_labels"][0])))
        return real_start_xy(data_dict, bb, train_size)
    mp.start_xy = start_xy

    class _Masks:
        def __init__(self, t):
            self.tensor = torch.as_tensor(t)

        def __len__(self):
            return self.tensor.shape[0]
    mp.BitMasks = _Masks

    names = [str(k) for k in np.load(os.path.join(HERE, "pool_decode.npz"))["keys"]]
    pool = {3: names[0:2], 7: names[2:3], 11: names[3:6]}
    ip = mp.InstPool.__new__(mp.InstPool)
    ip.dataset, ip.data_to_cat, ip.per_cat_pool = [], {}, {}
    for c, keys in pool.items():
        ip.per_cat_pool[c] = list(range(len(ip.dataset), len(ip.dataset) + len(keys)))
        ip.dataset += keys
        for k in keys:
            ip.data_to_cat[k] = c
    ip.cats = list(ip.per_cat_pool.keys())
    ip.HWms = {"4": [0.16, 0.05], "12": [0.2, 0.08]}             # category 7 (key "8") has no statistics: uniform-scale branch
    ip.augmentations = T.AugmentationList([T.RandomFlip()])
    ip.cumstom_augmentations = lambda image: {"image": image}    # albumentations.Compose([]) (COLOR_AUG false): identity, no draws
    ip.order_seed_state_dict = None
    ip.image_format, ip.cp_method, ip.max_samples = "RGBA", ["basic"], 8
    ip.bbox_occluded_thr, ip.mask_occluded_thr = 10, 300
    ip.scale_min, ip.scale_max, ip.instance_filter_min, ip.instance_filter_max = 10, 0.5, 0.01, 1.0
    ip.mask_threshold, ip.use_largest_part, ip.shape_jitter = 128, False, 0.2
    ip.random_scale, ip.random_scale_min, ip.random_scale_max, ip.random_scale_min_size = False, 0.5, 2.0, 5
    H, W = 160, 192
    rng = np.random.default_rng(33)
    store = {"pool_cats": np.array([3, 3, 7, 11, 11, 11]), "hw": np.array([H, W]), "max_samples": np.array(8),
             "HWms_keys": np.array(["4", "12"]), "HWms_vals": np.array([[0.16, 0.05], [0.2, 0.08]])}
    cwd = os.getcwd()
    os.chdir(HERE)
    try:
        for ci, seed in enumerate([11, 12, 13, 14, 15, 16, 17, 18]):
            n0 = int(rng.integers(0, 5)) if ci else 3
            img = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
            yy, xx = np.mgrid[0:H, 0:W]
            masks = np.zeros((n0, H, W), np.uint8)
            for i in range(n0):
                cx, cy, rx, ry = rng.uniform(20, W - 20), rng.uniform(20, H - 20), rng.uniform(6, 40), rng.uniform(6, 40)
                masks[i] = ((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1)
            boxes = mp.get_bboxes(masks)
            labels = rng.integers(0, 1203, n0).astype(np.int64)
            inst = mp.Instances((H, W))
            inst.gt_boxes, inst.gt_classes, inst.gt_masks = mp.Boxes(torch.from_numpy(boxes)), torch.from_numpy(labels), _Masks(masks)
            d = {"image": torch.from_numpy(img.copy()), "instances": inst, "file_name": "case%d" % ci}
            del log[:]
            np.random.seed(seed)
            out = ip.get_mix_result("cas_random", None, data_dict=d)
            after = np.random.randint(0, 2 ** 31 - 1)            # the stream position after the call: one more draw
            o = out["instances"]
            store.update({"c%d_seed" % ci: np.array(seed), "c%d_image" % ci: img, "c%d_masks" % ci: masks, "c%d_boxes" % ci: boxes,
                          "c%d_labels" % ci: labels, "c%d_after" % ci: np.array(after),
                          "c%d_resize" % ci: np.array([e[1:] for e in log if e[0] == "resize"], dtype=np.int64).reshape(-1, 4),
                          "c%d_place" % ci: np.array([e[1:] for e in log if e[0] == "place"], dtype=np.int64).reshape(-1, 3),
                          "c%d_out_image" % ci: npy(out["image"]), "c%d_out_boxes" % ci: npy(o.gt_boxes.tensor),
                          "c%d_out_labels" % ci: npy(o.gt_classes), "c%d_out_masks" % ci: npy(o.gt_masks.tensor).astype(np.uint8),
                          "c%d_out_source" % ci: npy(o.instance_source)})
    finally:
        os.chdir(cwd)
    save("pool_draws", **store)


def gen_bsgal():
    """BSGAL gradient bank: update_grad_bank / compute_grad_sim of the reference's CustomRCNN
    (BS/bsgal/modeling/meta_arch/custom_rcnn.py:1046-1086) called unbound on a stand-in `self`."""
    R.install()
    BS = R.REF + "/BSGAL/bsgal"
    R._ns("bsgal", BS)
    R._ns("bsgal.modeling", BS + "/modeling")
    R._ns("bsgal.modeling.meta_arch", BS + "/modeling/meta_arch")
    for nm in ("bsgal.modeling.text", "bsgal.modeling.text.text_encoder", "bsgal.modeling.utils", "torchshow"):
        sys.modules[nm] = R._Permissive(nm)
    rc = R._ns("detectron2.modeling.meta_arch.rcnn")
    rc.GeneralizedRCNN = nn.Module
    sys.modules["detectron2.utils.comm"].all_gather = lambda x: [x]
    sys.modules["detectron2.structures"].ROIMasks = R._PermissiveObj("ROIMasks")
    m = importlib.import_module("bsgal.modeling.meta_arch.custom_rcnn")
    C = m.CustomRCNN
    g = torch.Generator().manual_seed(31)
    n = 2051                                    # not a multiple of 4: exercises the tail
    store = {}
    for mode in ("AVERAGE", "MOMENTUM0.9"):
        bank = nn.Embedding(n, 1)
        bank.weight.requires_grad = False
        bank.weight.data.fill_(0)
        fake = types.SimpleNamespace(active_grad_save=True, active_grad_update=mode, iter=1, grad_bank=bank,
                                     output_dir="/tmp", rank="0", active_grad_norm=True)
        grads = [torch.randn(n, generator=g) * (0.1 + 0.3 * i) for i in range(4)]
        for it, gr in enumerate(grads):
            fake.iter = it + 1                  # iter % 10000 != 0: no checkpoint write
            out = C.update_grad_bank(fake, gr)
            store["%s_bank_%d" % (mode, it)] = out.clone()
        store["%s_grads" % mode] = torch.stack(grads)
        probe = torch.randn(n, generator=g)
        store["%s_probe" % mode] = probe
        store["%s_sim_norm" % mode] = C.compute_grad_sim(fake, probe, out)
        store["%s_sim_raw" % mode] = C.compute_grad_sim(fake, probe, out, norm=False)
    save("bsgal_bank", **store)


def gen_augment():
    """EfficientDetResizeCrop (DG/divergen/data/transforms/custom_augmentation_impl.py:24-72) + its transform
    (custom_transform.py:27-91) run from the reference's own files on seeded np.random streams: the transform parameters,
    the resized-and-cropped uint8 image, the nearest-neighbour segmentation path and transformed coordinates."""
    import importlib.util
    R.install()

    class _T:                                          # fvcore Transform: only the attribute helper is used
        def _set_attributes(self, params=None):
            if params:
                for k, v in params.items():
                    if k != "self" and not k.startswith("_"):
                        setattr(self, k, v)
    ft = types.ModuleType("fvcore.transforms.transform")
    for nm in ("BlendTransform", "CropTransform", "HFlipTransform", "NoOpTransform", "VFlipTransform", "TransformList"):
        setattr(ft, nm, type(nm, (_T,), {}))
    ft.Transform = _T
    sys.modules["fvcore.transforms"] = types.ModuleType("fvcore.transforms")
    sys.modules["fvcore.transforms.transform"] = ft
    am = types.ModuleType("detectron2.data.transforms.augmentation")
    am.Augmentation = type("Augmentation", (), {"__init__": lambda self: None})
    for nm in ("detectron2", "detectron2.data", "detectron2.data.transforms"):
        sys.modules.setdefault(nm, types.ModuleType(nm))
    sys.modules["detectron2.data.transforms.augmentation"] = am

    def load(name, path):
        sys.modules.pop(name, None)
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m
    load("divergen.data.transforms.custom_transform", R.DG + "/data/transforms/custom_transform.py")
    aug = load("divergen.data.transforms.custom_augmentation_impl", R.DG + "/data/transforms/custom_augmentation_impl.py")
    store = {}
    rng = np.random.default_rng(5)
    cases = [((96, 128), 160, (0.1, 2.0), 11), ((85, 128), 160, (0.1, 2.0), 12), ((66, 100), 128, (0.5, 1.5), 13),
             ((120, 80), 160, (1.0, 1.0), 14), ((64, 48), -1, (0.8, 1.2), 15), ((96, 128), 160, (0.1, 2.0), 16)]
    for ci, ((h, w), size, scale, seed) in enumerate(cases):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        seg = (rng.integers(0, 4, (h, w)) * 60).astype(np.uint8)
        np.random.seed(seed)
        t = aug.EfficientDetResizeCrop(size, scale).get_transform(img)
        pts = rng.uniform(0, min(h, w), (7, 2))
        store["c%d_img" % ci], store["c%d_seg" % ci], store["c%d_pts" % ci] = img, seg, pts
        store["c%d_cfg" % ci] = np.array([size, scale[0], scale[1], seed], dtype=np.float64)
        store["c%d_params" % ci] = np.array([t.scaled_h, t.scaled_w, t.offset_y, t.offset_x, t.img_scale,
                                              t.target_size[0], t.target_size[1]], dtype=np.float64)
        store["c%d_out" % ci] = t.apply_image(img)
        store["c%d_seg_out" % ci] = t.apply_segmentation(seg)
        store["c%d_pts_out" % ci] = t.apply_coords(pts.copy())
    save("augment", **store)


def gen_samplers():
    """Index streams of the reference's own samplers (D2/data/samplers/distributed_sampler.py) for seeded generators: what
    divergen_amd/data/samplers.py has to reproduce so that a run visits the images in the reference's order."""
    import importlib.util
    import itertools
    comm = types.ModuleType("detectron2.utils.comm")
    comm.get_rank, comm.get_world_size, comm.shared_random_seed = (lambda: 0), (lambda: 1), (lambda: 1)
    for nm in ("detectron2", "detectron2.utils"):
        sys.modules.setdefault(nm, types.ModuleType(nm))
    sys.modules["detectron2.utils.comm"] = comm
    sys.modules["detectron2.utils"].comm = comm
    spec = importlib.util.spec_from_file_location("ref_distributed_sampler", R.D2 + "/data/samplers/distributed_sampler.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    g = torch.Generator().manual_seed(3)
    rf = 1.0 + 2.5 * torch.rand(37, generator=g) ** 3
    store = {"repeat_factors": npy(rf)}
    cases = [(7, 0, 1, True), (7, 1, 3, True), (11, 2, 4, True), (5, 1, 2, False)]
    store["cases"] = np.array([[a, b, c, int(d)] for a, b, c, d in cases], dtype=np.int64)
    for ci, (seed, rank, world, shuffle) in enumerate(cases):
        s = m.RepeatFactorTrainingSampler(rf, shuffle=shuffle, seed=seed)
        s._rank, s._world_size = rank, world
        store["rf_%d" % ci] = np.array(list(itertools.islice(iter(s), 300)), dtype=np.int64)
        s = m.TrainingSampler(23, shuffle, seed)
        s._rank, s._world_size = rank, world
        store["tr_%d" % ci] = np.array(list(itertools.islice(iter(s), 100)), dtype=np.int64)
    shards = []
    for tot, w in ((100, 4), (10, 3), (5, 8), (0, 2)):
        for r in range(w):
            rg = m.InferenceSampler._get_local_indices(tot, w, r)
            shards.append([tot, w, r, rg.start if len(rg) else 0, len(rg)])
    store["inference_shards"] = np.array(shards, dtype=np.int64)
    save("samplers", **store)


if __name__ == "__main__":
    which = sys.argv[1:] or ["swin", "centernet", "roi", "compositor", "solver", "heads", "heads_wide", "postprocess", "inference", "pool", "bsgal", "augment", "samplers", "pool_draws"]
    for w in which:
        globals()["gen_" + w]()
