"""GPU parity of the host-mirror MODULES (built on libdgx) against golden outputs of the reference's
own source files (tests/golden/*.npz) and the CPU oracle.  The product has one precision (bf16 operands, fp32
accumulation, on its own HIP kernels): tolerances are stated relative to each tensor's scale."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

from oracle import heads as OH  # noqa: E402
from oracle import roi as OR  # noqa: E402
from tests._recipes import fill_state, swin_param_shapes  # noqa: E402


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_close(a, b, frac):
    err = float((a - b).abs().max())
    scale = float(b.abs().max())
    assert err <= frac * scale, (err, scale)


def rel_l2(a, b, frac):
    """||a - b|| / ||b||: the metric for gradients that passed ReLUs.  A pre-activation within bf16 rounding (2^-9) of zero flips
    its mask bit against the fp32 reference, which moves that element's gradient by its full magnitude: ~0.3 % of the elements
    per ReLU layer = ~5 % relative L2 per layer, independent of the arithmetic (the reference's own fp16 autocast has the same
    property at 2^-11).  `frac` below a stack of k ReLU layers is therefore budgeted as ~0.05 sqrt(k), not as rounding."""
    err = float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))
    assert err <= frac, err


def test_swin_backbone_vs_reference_golden(golden):
    """Whole SwinTransformer (reference golden, fp32) vs the HIP-backed module under bf16 autocast."""
    from divergen_amd.modeling.backbone.swintransformer import SwinTransformer
    g = golden("swin_full")
    net = SwinTransformer(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=7,
                          drop_path_rate=0.0, out_indices=(1, 2, 3))
    p = fill_state(swin_param_shapes(32, [2, 2, 2, 2], [1, 2, 4, 8], 7), int(g["param_seed"]), float(g["param_scale"]))
    missing, unexpected = net.load_state_dict(p, strict=False)
    assert not unexpected and all("relative_position_index" in m for m in missing)
    net = net.to(DEV).train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outs = net(T(g["img"]).to(DEV))
    for k in ("swin1", "swin2", "swin3"):
        rel_close(outs[k].float().cpu(), T(g[k]), 0.04)      # 8 blocks of bf16 GEMMs + bf16 residual stream


@pytest.mark.parametrize("ws", [7, 12])
def test_basic_layer_vs_reference_golden(golden, ws):
    from divergen_amd.modeling.backbone.swintransformer import BasicLayer, PatchMerging
    g = golden("swin_layer_w%d" % ws)
    layer = BasicLayer(dim=64, depth=2, num_heads=2, window_size=ws, drop_path=0.0, downsample=PatchMerging)
    sd = {k[2:]: T(g[k]) for k in g.files if k.startswith("p.")}
    layer.load_state_dict(sd, strict=False)
    layer = layer.to(DEV).train()
    H, W = int(g["H"]), int(g["W"])
    x = T(g["x"]).to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        x_out, _, _, x_down, wh, ww = layer(x, H, W)
    assert (wh, ww) == (int(g["Wh"]), int(g["Ww"]))
    rel_close(x_out.float().cpu(), T(g["x_out"]), 0.03)
    rel_close(x_down.float().cpu(), T(g["x_down"]), 0.03)
    ((x_out.float() * T(g["g1"]).to(DEV)).sum() + (x_down.float() * T(g["g2"]).to(DEV)).sum()).backward()
    rel_close(x.grad.float().cpu(), T(g["dx"]), 0.05)
    gq = dict(layer.named_parameters())["blocks.1.attn.qkv.weight"].grad
    rel_close(gq.float().cpu(), T(g["g.blocks.1.attn.qkv.weight"]), 0.05)
    gt = dict(layer.named_parameters())["blocks.1.attn.relative_position_bias_table"].grad
    rel_close(gt.float().cpu(), T(g["g.blocks.1.attn.relative_position_bias_table"]), 0.06)


def _grad_of(p):
    """fp32 gradient of a parameter in the reference's layout (3x3 weights live (Cout, kh, kw, Cin) in an arena)."""
    return p.grad.float().cpu()


def test_fpn_vs_reference_golden_hip_path(golden):
    """D2 FPN + LastLevelP6P7_P5 at the product's width (laterals 64 / 128 / 256 -> 256: 1x1 = MFMA GEMM, 3x3 stride 1 =
    implicit GEMM, P6 / P7 stride 2 = im2col + GEMM) against the reference's own modules run in fp32 on the same bf16-exact
    weights and inputs (tests/golden/make_golden.py:gen_heads_wide): values, input gradients, weight / bias gradients.
    Tolerance = bf16 rounding of the intermediate maps: 1 % of each tensor's scale."""
    from divergen_amd.modeling.backbone.fpn import FPN, LastLevelP6P7_P5
    from divergen_amd.modeling.backbone.swintransformer import Backbone
    from tests._recipes import fill_by_name

    class Dummy(Backbone):
        _out_features = ["swin1", "swin2", "swin3"]
        _out_feature_channels = {"swin1": 64, "swin2": 128, "swin3": 256}
        _out_feature_strides = {"swin1": 8, "swin2": 16, "swin3": 32}

        def forward(self, x):
            return x

    g = golden("fpn_wide")
    fpn = FPN(Dummy(), ["swin1", "swin2", "swin3"], 256, top_block=LastLevelP6P7_P5(256, 256))
    fill_by_name(fpn, 71, 0.03)
    fpn = fpn.to(DEV).train()
    ins = {k[3:]: T(g[k]).to(DEV).to(torch.bfloat16).to(memory_format=torch.channels_last).requires_grad_(True)
           for k in g.files if k.startswith("in.")}
    out = fpn(ins)
    sum((out[k].float() * T(g["go." + k]).to(DEV)).sum() for k in out).backward()
    for k in ("p3", "p4", "p5", "p6", "p7"):
        assert out[k].dtype == torch.bfloat16
        rel_close(out[k].float().cpu(), T(g["out." + k]), 0.01)
    for k in ("swin1", "swin2", "swin3"):
        rel_close(ins[k].grad.float().cpu(), T(g["din." + k]), 0.015)
    P = dict(fpn.named_parameters())
    rel_close(_grad_of(P["fpn_lateral5.weight"]), T(g["g.fpn_lateral5.weight"]), 0.015)
    rel_close(_grad_of(P["fpn_output3.bias"]), T(g["g.fpn_output3.bias"]), 0.015)
    rel_close(_grad_of(P["fpn_output4.weight"])[:8], T(g["g.fpn_output4.weight.rows8"]), 0.015)
    rel_close(_grad_of(P["top_block.p6.weight"])[:8], T(g["g.top_block.p6.weight.rows8"]), 0.015)


def test_centernet_head_vs_reference_golden_hip_path(golden):
    """CenterNetHead (4 x [3x3 conv, GroupNorm(32), ReLU] tower, 4 + 1 channel predictors, per-level Scale + ReLU) at 256
    channels on the HIP path against the reference's own module in fp32 on the same bf16-exact weights / inputs."""
    from divergen_amd.modeling.dense_heads.centernet_head import CenterNetHead
    from tests._recipes import fill_by_name
    g = golden("centernet_head_wide")
    head = CenterNetHead(in_channels=256, num_levels=2, num_classes=5, with_agn_hm=True, only_proposal=True)
    fill_by_name(head, 72, 0.02)
    with torch.no_grad():
        head.bbox_pred.bias.fill_(2.0)
    head = head.to(DEV).train()
    xs = [T(g["x%d" % i]).to(DEV).to(torch.bfloat16).to(memory_format=torch.channels_last).requires_grad_(True) for i in range(2)]
    _, regs, hms = head(xs)
    (sum((regs[i].float() * T(g["gr%d" % i]).to(DEV)).sum() for i in range(2))
     + sum((hms[i].float() * T(g["gh%d" % i]).to(DEV)).sum() for i in range(2))).backward()
    for i in range(2):
        rel_close(regs[i].float().cpu(), T(g["reg%d" % i]), 0.015)          # four conv + GroupNorm rounds in bf16
        rel_close(hms[i].float().cpu(), T(g["hm%d" % i]), 0.015)
        rel_l2(xs[i].grad.float().cpu(), T(g["dx%d" % i]), 0.12)
    P = dict(head.named_parameters())
    rel_l2(_grad_of(P["bbox_tower.0.weight"])[:8], T(g["g.bbox_tower.0.weight.rows8"]), 0.12)
    rel_l2(_grad_of(P["bbox_tower.1.weight"]), T(g["g.bbox_tower.1.weight"]), 0.12)
    rel_l2(_grad_of(P["bbox_tower.1.bias"]), T(g["g.bbox_tower.1.bias"]), 0.12)
    rel_close(_grad_of(P["bbox_pred.weight"]), T(g["g.bbox_pred.weight"]), 0.02)
    rel_close(_grad_of(P["agn_hm.bias"]), T(g["g.agn_hm.bias"]), 0.02)
    rel_close(_grad_of(P["scales.1.scale"]).reshape(-1), T(g["g.scales.1.scale"]).reshape(-1), 0.02)


def test_mask_and_box_head_vs_reference_golden_hip_path(golden):
    """MaskRCNNConvUpsampleHead (4 x 3x3 conv + ReLU, ConvTranspose2d(2, 2) + ReLU, 1x1 predictor) and FastRCNNConvFCHead
    (2 x Linear + ReLU over the 256 x 7 x 7 RoI features) on the HIP path against the reference's modules (fp32, same
    bf16-exact weights / inputs): values, input gradients, weight / bias gradients."""
    from divergen_amd.modeling import ShapeSpec
    from divergen_amd.modeling.roi_heads.box_head import FastRCNNConvFCHead
    from divergen_amd.modeling.roi_heads.mask_head import MaskRCNNConvUpsampleHead
    from tests._recipes import fill_by_name
    g = golden("mask_head_wide")
    mask = MaskRCNNConvUpsampleHead(ShapeSpec(channels=256, height=14, width=14), num_classes=1, conv_dims=[256] * 5, conv_norm="")
    fill_by_name(mask, 73, 0.02)
    mask = mask.to(DEV).train()
    x = T(g["x"]).to(DEV).to(torch.bfloat16).to(memory_format=torch.channels_last).requires_grad_(True)
    logits = mask.layers(x)
    (logits.float() * T(g["go"]).to(DEV)).sum().backward()
    rel_close(logits.float().cpu(), T(g["logits"]), 0.015)
    rel_l2(x.grad.float().cpu(), T(g["dx"]), 0.12)
    P = dict(mask.named_parameters())
    rel_l2(_grad_of(P["mask_fcn1.weight"])[:8], T(g["g.mask_fcn1.weight.rows8"]), 0.12)
    rel_l2(_grad_of(P["deconv.weight"])[:8], T(g["g.deconv.weight.rows8"]), 0.12)
    rel_l2(_grad_of(P["deconv.bias"]), T(g["g.deconv.bias"]), 0.12)
    rel_close(_grad_of(P["predictor.weight"]), T(g["g.predictor.weight"]), 0.02)
    rel_close(_grad_of(P["predictor.bias"]), T(g["g.predictor.bias"]), 0.02)

    g = golden("box_head_wide")
    box = FastRCNNConvFCHead(ShapeSpec(channels=256, height=7, width=7), conv_dims=[], fc_dims=[1024, 1024])
    fill_by_name(box, 74, 0.01)
    box = box.to(DEV).train()
    xb = T(g["x"]).to(DEV).to(torch.bfloat16).requires_grad_(True)
    yb = box(xb)
    (yb.float() * T(g["go"]).to(DEV)).sum().backward()
    rel_close(yb.float().cpu(), T(g["y"]), 0.01)
    rel_l2(xb.grad.float().cpu(), T(g["dx"]), 0.08)
    Pb = dict(box.named_parameters())
    # fc1's columns are stored in (h, w, c) order; the golden gradient rows are in the reference's (c, h, w) order
    rel_l2(box._cols_to_chw(_grad_of(Pb["fc1.weight"]).to(DEV))[:4].cpu(), T(g["g.fc1.weight.rows4"]), 0.08)
    rel_l2(_grad_of(Pb["fc2.weight"])[:16], T(g["g.fc2.weight.rows16"]), 0.08)
    rel_l2(_grad_of(Pb["fc2.bias"]), T(g["g.fc2.bias"]), 0.08)


def test_centernet_targets_and_losses_vs_reference_golden(golden):
    from divergen_amd.modeling.dense_heads.centernet import CenterNet
    from divergen_amd.structures import Boxes, Instances
    from divergen_amd.utils.events import EventStorage
    g = golden("centernet_targets")
    net = CenterNet(in_channels=16, num_classes=7, with_agn_hm=True, only_proposal=True, score_thresh=0.0001,
                    reg_weight=1.0, not_norm_reg=True, pos_weight=0.5, neg_weight=0.5, ignore_high_fp=0.85,
                    centernet_head=torch.nn.Identity()).to(DEV).train()
    H, W, st = int(g["H"]), int(g["W"]), [int(s) for s in g["strides"]]
    shapes = [(-(-H // s), -(-W // s)) for s in st]
    gts = []
    for k in ("gt2a_boxes", "gt2b_boxes"):
        inst = Instances((H, W))
        inst.gt_boxes = Boxes(T(g[k]).to(DEV))
        gts.append(inst)
    pos, reg, hm = net._get_ground_truth(shapes, gts)
    # the product path keeps (indices, cared) at fixed length; the reference's compacted list is indices[cared]
    assert isinstance(pos, tuple) and pos[0].shape == pos[1].shape
    assert torch.equal(pos[0][pos[1]].cpu(), T(g["pos2"]))           # index tensor: bit-exact
    assert torch.equal(net._get_label_inds([x.gt_boxes.tensor for x in gts], shapes).cpu(), T(g["pos2"]))
    assert torch.equal(reg.cpu(), T(g["reg2"]))
    torch.testing.assert_close(hm.cpu(), T(g["hm2"]), atol=1e-6, rtol=1e-5)
    rp = T(g["reg_pred"]).to(DEV).requires_grad_(True)
    al = T(g["agn_logit"]).to(DEV).requires_grad_(True)
    with EventStorage(0):
        losses = net.losses(pos, reg, hm, rp, al)
    # reference: fp32 losses within 1e-3 (north_star); here ~1e-6
    torch.testing.assert_close(losses["loss_centernet_loc"].cpu(), T(g["loss_loc"]), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(losses["loss_centernet_agn_pos"].cpu(), T(g["loss_pos"]), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(losses["loss_centernet_agn_neg"].cpu(), T(g["loss_neg"]), atol=1e-5, rtol=1e-5)
    sum(losses.values()).backward()
    torch.testing.assert_close(rp.grad.cpu(), T(g["d_reg_pred"]), atol=1e-6, rtol=1e-4)
    torch.testing.assert_close(al.grad.cpu(), T(g["d_agn_logit"]), atol=1e-6, rtol=1e-4)


def test_detic_losses_vs_reference_golden(golden, monkeypatch):
    import divergen_amd.modeling.roi_heads.detic_fast_rcnn as M
    from divergen_amd.modeling.box_regression import Box2BoxTransform
    g = golden("roi_losses")
    appeared = T(g["appeared"]).to(DEV)
    monkeypatch.setattr(M, "get_fed_loss_inds", lambda *a, **k: appeared)   # RNG-drawn class set from the golden
    fake = types.SimpleNamespace(use_fed_loss=True, freq_weight=T(g["freq"]).to(DEV), fed_loss_num_cat=10,
                                 ignore_zero_cats=False, num_classes=40, smooth_l1_beta=0.0,
                                 box2box_transform=Box2BoxTransform(tuple(g["weights"].tolist())))
    logits = T(g["logits"]).to(DEV).requires_grad_(True)
    gtc = T(g["gt_classes"]).to(DEV)
    l = M.DeticFastRCNNOutputLayers.sigmoid_cross_entropy_loss(fake, logits, gtc)
    torch.testing.assert_close(l.cpu(), T(g["loss_cls"]), atol=1e-5, rtol=1e-5)
    l.backward()
    torch.testing.assert_close(logits.grad.cpu(), T(g["d_logits"]), atol=1e-7, rtol=1e-4)
    pd = T(g["pred_deltas"]).to(DEV).requires_grad_(True)
    lb = M.DeticFastRCNNOutputLayers.box_reg_loss(fake, T(g["prop_boxes"]).to(DEV), T(g["gt_boxes"]).to(DEV), pd, gtc)
    torch.testing.assert_close(lb.cpu(), T(g["loss_box"]), atol=1e-5, rtol=1e-5)


def test_mask_loss_vs_oracle():
    from divergen_amd.modeling.roi_heads.mask_head import mask_rcnn_loss
    from divergen_amd.structures import BitMasks, Boxes, Instances
    from divergen_amd.utils.events import EventStorage
    g = torch.Generator().manual_seed(91)
    H, W = 120, 160
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    insts, masks_l, boxes_l = [], [], []
    for n in (5, 3):
        m = torch.zeros(n, H, W, dtype=torch.bool)
        for i in range(n):
            cx, cy = torch.rand(2, generator=g) * torch.tensor([W, H])
            m[i] = ((xx - cx) / 30) ** 2 + ((yy - cy) / 20) ** 2 <= 1
        xy = torch.rand(n, 2, generator=g) * 80
        b = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 60 + 4], 1)
        inst = Instances((H, W))
        inst.proposal_boxes, inst.gt_masks = Boxes(b.to(DEV)), BitMasks(m.to(DEV))
        insts.append(inst)
        masks_l.append(m)
        boxes_l.append(b)
    logits = torch.randn(8, 1, 28, 28, generator=g)
    ref = OH.mask_loss(logits, masks_l, boxes_l)
    with EventStorage(0):
        got = mask_rcnn_loss(logits.to(DEV), insts)
    torch.testing.assert_close(got.cpu(), ref, atol=1e-6, rtol=1e-5)   # identical bool targets -> identical BCE


def test_fused_optimizer_trajectory_vs_reference_golden(golden):
    """12 steps of AdamW + value clip + EMA + WarmupCosineLR on the golden's tiny net (reference:
    torch.optim.AdamW + clip_grad_value_ + ModelEma + D2 WarmupCosineLR)."""
    from divergen_amd.solver import FlatArena, FusedAdamWEMA, WarmupCosineLR
    g = golden("solver")
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LayerNorm(5), torch.nn.Linear(5, 3))
    net.load_state_dict({k[5:]: T(g[k]) for k in g.files if k.startswith("init.")})
    net = net.to(DEV)
    arena = FlatArena(net)
    opt = FusedAdamWEMA(arena, 1e-2, weight_decay=1e-4, clip_value=1.0, ema_decay=0.999)
    sched = WarmupCosineLR(opt, 100, warmup_factor=1e-4, warmup_iters=10)
    x = T(g["x"]).to(DEV)
    for it in range(12):
        assert abs(opt.param_groups[0]["lr"] - float(g["lrs"][it])) < 1e-12
        opt.zero_grad()
        ((net(x) ** 2).sum() * 30).backward()
        opt.step()
        sched.step()
    sd = net.state_dict()
    ema = opt.ema_state_dict(net)
    for k in sd:
        torch.testing.assert_close(sd[k].cpu(), T(g["final." + k]), atol=2e-6, rtol=2e-5)
        torch.testing.assert_close(ema[k].cpu(), T(g["ema." + k]), atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("xdt,shift", [(torch.float32, 0), (torch.bfloat16, 6), (torch.float32, 6)])
def test_fused_swin_block_equals_composed_path(xdt, shift, monkeypatch):
    """The single-node Swin block (divergen_amd/layers/swin_block.py) runs the same kernels in the same order
    as the composed path: outputs and input gradients must be bitwise equal, parameter gradients equal up to the
    order of fp32 atomic / split-M additions."""
    from divergen_amd.modeling.backbone import swintransformer as S
    from divergen_amd.solver import FlatArena
    torch.manual_seed(5)
    dim, nH, ws, B, H, W = 192, 6, 12, 2, 30, 26
    blk = S.SwinTransformerBlock(dim, nH, window_size=ws, shift_size=shift, drop_path=0.2).to(DEV).train()
    for p in blk.parameters():
        torch.nn.init.normal_(p, std=0.05)
    blk.H, blk.W = H, W
    arena = FlatArena(blk)
    from divergen_amd.layers import shift_regions
    region = shift_regions(H, W, ws).to(DEV) if shift else None
    x0 = torch.randn(B, H * W, dim, device=DEV).to(xdt)
    go = torch.randn(B, H * W, dim, device=DEV).to(xdt)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(S, "_FUSED_BLOCK", fused)
        arena.zero_grad()
        torch.manual_seed(11)           # same DropPath draw
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(x, region)
        y.backward(go)
        res[fused] = (y.detach().clone(), x.grad.clone(), arena.g.clone())
    assert res[True][0].dtype == res[False][0].dtype
    # same kernels in the same order except the activation: the fused path evaluates GELU inside the fc1 GEMM's epilogue
    # (erfc polynomial, |error| <= 1.5e-7) where the composed path calls torch's erf GELU -- a bf16 ulp on a few
    # activations, carried through fc2 into the block output
    yt, yf = res[True][0].float(), res[False][0].float()
    assert torch.equal(res[True][0], res[False][0]) or (yt - yf).abs().max() <= (2e-4 if xdt == torch.float32 else 2e-2) * yf.abs().max()
    assert torch.equal(res[True][1], res[False][1]) or (res[True][1].float() - res[False][1].float()).abs().max() <= (
        2e-4 if xdt == torch.float32 else 2e-2) * res[False][1].float().abs().max()
    ga, gb = res[True][2], res[False][2]
    assert ga.abs().max() > 0
    assert (ga - gb).abs().max() <= 2e-3 * gb.abs().max()


@pytest.mark.parametrize("dim,nH,H,W", [(192, 6, 30, 26), (768, 24, 32, 32), (384, 12, 13, 40), (192, 6, 24, 24)])
@pytest.mark.parametrize("shift", [0, 6])
def test_swin_block_compact_equals_padded(dim, nH, H, W, shift, monkeypatch):
    """layers/swin_block.COMPACT: the rows between LayerNorm-1 and proj without the padding tokens (csrc/winmap.h) against the
    padded form of rounds 1-5 -- the reference's own layout (swintransformer.py:216-251).  Exact work removal: block output and input
    gradient equal (bitwise unless the GEMM plan changes with M), every parameter gradient equal to fp32 summation order, the
    relative-position table's included (its sum runs over the same (query, key) pairs: padding keys stay).  (24, 24) has no padding:
    both settings are the same code path."""
    from divergen_amd.layers import shift_regions
    from divergen_amd.layers import swin_block as SB
    from divergen_amd.modeling.backbone import swintransformer as S
    from divergen_amd.solver import FlatArena
    torch.manual_seed(5)
    ws, B = 12, 2
    blk = S.SwinTransformerBlock(dim, nH, window_size=ws, shift_size=shift, drop_path=0.2).to(DEV).train()
    for p in blk.parameters():
        torch.nn.init.normal_(p, std=0.05)
    blk.H, blk.W = H, W
    arena = FlatArena(blk)
    region = shift_regions(H, W, ws).to(DEV) if shift else None
    x0 = torch.randn(B, H * W, dim, device=DEV).bfloat16()
    go = torch.randn(B, H * W, dim, device=DEV).bfloat16()
    res = {}
    for compact in (True, False):
        monkeypatch.setattr(SB, "COMPACT", compact)
        assert SB.compact_ok(H, W, ws, shift) == (compact and (H % ws != 0 or W % ws != 0))
        arena.zero_grad()
        torch.manual_seed(11)           # same DropPath draw
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(x, region)
        y.backward(go)
        SB.flush_wgrads()
        torch.cuda.synchronize()
        res[compact] = (y.detach().clone(), x.grad.clone(), arena.g.clone())
    (yc, dxc, gc), (yp, dxp, gp) = res[True], res[False]
    assert float(yp.float().abs().max()) > 0 and float(gp.abs().max()) > 0
    assert torch.equal(yc, yp) or float((yc.float() - yp.float()).abs().max()) <= 8e-3 * float(yp.float().abs().max())
    assert torch.equal(dxc, dxp) or float((dxc.float() - dxp.float()).abs().max()) <= 8e-3 * float(dxp.float().abs().max())
    assert float((gc - gp).abs().max()) <= 2e-3 * float(gp.abs().max())
    for name, o, z in zip(arena.names, arena.offsets, arena.sizes):          # per parameter too (the qkv bias and the table are small ones)
        a, b_ = gc[o:o + z].double(), gp[o:o + z].double()
        assert float((a - b_).norm()) <= 2e-3 * float(b_.norm()) + 1e-7, name


def test_swin_blocks_weight_gradients_loader_wave_group(monkeypatch):
    """Three Swin blocks of width 768 whose weight gradients are queued for ONE launch of the persistent loader-wave kernel
    (layers/swin_block.py::_defer_wgrads -> csrc/wgrad_lw.hip; 3 x 144 tiles = 0.84 of two rounds of the chip) against the same blocks
    with one split-M launch per group of <= 12 problems (csrc/wgrad256.hip): same outputs, same input gradient (the data path does not
    change), parameter gradients to fp32 summation order; the lazy zero-grad protocol must see every segment written in both."""
    from divergen_amd.modeling.backbone import swintransformer as S
    from divergen_amd.layers import swin_block as SB
    from divergen_amd.solver import FlatArena
    torch.manual_seed(15)
    dim, nH, ws, B, H, W = 768, 24, 12, 2, 24, 24          # 1 152 tokens: every contraction >= 1 024 rows
    blocks = torch.nn.ModuleList([S.SwinTransformerBlock(dim, nH, window_size=ws, shift_size=0 if i % 2 == 0 else 6, drop_path=0.1)
                                  for i in range(3)]).to(DEV).train()
    for p in blocks.parameters():
        torch.nn.init.normal_(p, std=0.03)
    for b in blocks:
        b.H, b.W = H, W
    arena = FlatArena(blocks)
    from divergen_amd.layers import shift_regions
    region = shift_regions(H, W, ws).to(DEV)
    x0 = torch.randn(B, H * W, dim, device=DEV).bfloat16()
    go = torch.randn(B, H * W, dim, device=DEV).bfloat16()
    launched = []
    orig = SB.wgrad_grouped
    monkeypatch.setattr(SB, "wgrad_grouped", lambda problems, beta=1.0: (launched.append(len(problems)), orig(problems, beta))[1])
    res = {}
    for lw in (True, False):
        monkeypatch.setattr(SB, "_LW_MIN_M", 1024 if lw else 0)
        monkeypatch.setattr(SB, "_LW_ITEMS", 400)
        del launched[:]
        arena.zero_grad(lazy=True)
        torch.manual_seed(21)           # same DropPath draws
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = x
            for i, b in enumerate(blocks):
                y = b(y, region if i % 2 else None)
        y.backward(go)
        arena.finish_grads()
        res[lw] = (y.detach().clone(), x.grad.clone(), arena.g.clone(), list(launched))
    assert res[True][3] == [12], res[True][3]              # the three blocks' 12 problems in one launch
    assert len(res[False][3]) >= 2 and max(res[False][3]) <= 12
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    ga, gb = res[True][2], res[False][2]
    assert float(gb.abs().max()) > 0
    assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-6


def test_fused_detic_losses_vs_reference_golden(golden):
    """dgx_detic_losses (one pass: sigmoid CE with the federated class weights + L1 box regression + gradients) against
    the outputs of the reference's own loss methods (golden 'roi_losses')."""
    import divergen_amd.modeling.roi_heads.detic_fast_rcnn as M
    g = golden("roi_losses")
    C = 40
    w = torch.zeros(C + 1)
    w[T(g["appeared"]).long()] = 1
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
        logits = T(g["logits"]).to(DEV).to(dt).requires_grad_(True)
        pd = T(g["pred_deltas"]).to(DEV).to(dt).requires_grad_(True)
        gtc = T(g["gt_classes"]).to(DEV)
        lc, lb, out = M._DeticLosses.apply(logits, pd, gtc, w[:C].to(DEV), T(g["prop_boxes"]).to(DEV), T(g["gt_boxes"]).to(DEV), None,
                                           tuple(g["weights"].tolist()))
        torch.testing.assert_close(lc.float().cpu(), T(g["loss_cls"]), atol=tol, rtol=tol)
        torch.testing.assert_close(lb.float().cpu(), T(g["loss_box"]), atol=tol, rtol=tol)
        (lc + lb).backward()
        torch.testing.assert_close(logits.grad.float().cpu(), T(g["d_logits"]), atol=1e-7 if dt == torch.float32 else 2e-4, rtol=1e-4 if dt == torch.float32 else 2e-2)
        # box gradient: sign(pred - target) / (4 * selected rows) on foreground rows
        fgm = (gtc >= 0) & (gtc < C)
        assert float(pd.grad.float()[~fgm].abs().max()) == 0.0
        assert torch.allclose(pd.grad.float()[fgm].abs(), torch.full_like(pd.grad.float()[fgm], 1.0 / (4 * int(fgm.sum()))), rtol=1e-2)
        # statistics = D2 fast_rcnn.py:88-114 on the same logits
        pred = logits.detach().float().argmax(1)
        assert abs(float(out[11]) - float((pred == gtc).float().mean())) < 1e-6


def test_fed_loss_class_mask_matches_reference_draw():
    """The sync-free class-set draw keeps the reference's set for the same generator state (multinomial without
    replacement == top-k of prob / Exponential(1)), and its invariants."""
    import divergen_amd.modeling.roi_heads.detic_fast_rcnn as M
    C, K = 1203, 50
    g = torch.Generator().manual_seed(3)
    weight = (torch.rand(C, generator=g) ** 2).to(DEV)
    weight[::7] = 0
    for n_app in (0, 5, 49, 50, 80):
        gt = torch.cat([torch.randperm(C, generator=g)[:n_app], torch.full((30,), C)]).to(DEV)      # + background rows
        torch.manual_seed(1234)
        ref = M.get_fed_loss_inds(gt, K, C, weight)
        torch.manual_seed(1234)
        m = M.fed_loss_class_mask(gt, K, C, weight)
        app = torch.unique(gt)
        assert bool(m[app].all())
        assert int(m.sum()) == max(len(app), K)
        extra = m.clone()
        extra[app] = False
        assert bool((weight[extra[:C].nonzero().squeeze(1)] > 0).all())
        ref_m = torch.zeros(C + 1, dtype=torch.bool, device=DEV)
        ref_m[ref] = True
        assert torch.equal(m, ref_m), n_app


def test_cascade_refine_equals_composed_ops():
    """dgx_cascade_refine = apply_deltas + clip + nonempty + iou_match + gathers of the composed path, bit for bit on
    the integer outputs (matched class / source, validity) and on the decoded boxes."""
    import ctypes
    from divergen_amd import _lib as L
    from divergen_amd.layers import iou_match
    from divergen_amd.modeling.box_regression import Box2BoxTransform
    g = torch.Generator().manual_seed(77)
    sizes, counts, gts, C = [(480, 640), (512, 333)], [300, 212], [7, 0], 1203
    tr = Box2BoxTransform((20.0, 20.0, 10.0, 10.0))
    props, deltas, gtb, gtc, gsrc = [], [], [], [], []
    for (H, W), n, m in zip(sizes, counts, gts):
        xy = torch.rand(n, 2, generator=g) * torch.tensor([W * 0.8, H * 0.8])
        wh = torch.rand(n, 2, generator=g) * 120 + 1
        props.append(torch.cat([xy, xy + wh], 1))
        d = torch.randn(n, 4, generator=g) * 1.5
        d[::17, 0] = 4000.0         # decoded far outside the image -> empty after clipping
        gxy = torch.rand(m, 2, generator=g) * torch.tensor([W * 0.5, H * 0.5])
        gtb.append(torch.cat([gxy, gxy + torch.rand(m, 2, generator=g) * 150 + 20], 1))
        if m:                       # a few proposals sit on ground-truth boxes with ~zero deltas -> foreground matches
            props[-1][1:1 + m] = gtb[-1]
            d[1:1 + m] = torch.randn(m, 4, generator=g) * 0.05
        deltas.append(d)
        gtc.append(torch.randint(0, C, (m,), generator=g))
        gsrc.append(torch.randint(0, 2, (m,), generator=g))
    P, D = torch.cat(props).to(DEV), torch.cat(deltas).to(DEV)
    R, B = P.shape[0], 2
    out = dict(nb=torch.empty(R, 4, device=DEV), valid=torch.empty(R, dtype=torch.uint8, device=DEV), cls=torch.empty(R, dtype=torch.int64, device=DEV),
               gtb=torch.empty(R, 4, device=DEV), src=torch.empty(R, dtype=torch.int64, device=DEV), nfg=torch.empty(1, dtype=torch.int32, device=DEV))
    row0 = (ctypes.c_int * 3)(0, counts[0], R)
    gt0 = (ctypes.c_int * 3)(0, gts[0], gts[0] + gts[1])
    ih, iw = (ctypes.c_float * 2)(*[float(s[0]) for s in sizes]), (ctypes.c_float * 2)(*[float(s[1]) for s in sizes])
    GB, GC, GS = torch.cat(gtb).to(DEV), torch.cat(gtc).to(DEV), torch.cat(gsrc).to(DEV)
    L.check(L.lib().dgx_cascade_refine(L.ptr(P), L.ptr(D), None, B, row0, gt0, ih, iw, L.ptr(GB), L.ptr(GC), L.ptr(GS), 0.7, C,
                                       20.0, 20.0, 10.0, 10.0, float(tr.scale_clamp), L.ptr(out["nb"]), L.ptr(out["valid"]), L.ptr(out["cls"]),
                                       L.ptr(out["gtb"]), L.ptr(out["src"]), L.ptr(out["nfg"]), 0, L.stream()), "refine")
    r0, nfg = 0, 0
    for i, ((H, W), n, m) in enumerate(zip(sizes, counts, gts)):
        b = tr.apply_deltas(D[r0:r0 + n], P[r0:r0 + n])
        b = torch.stack([b[:, 0].clamp(0, W), b[:, 1].clamp(0, H), b[:, 2].clamp(0, W), b[:, 3].clamp(0, H)], 1)
        ok = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
        torch.testing.assert_close(out["nb"][r0:r0 + n], b, atol=2e-3, rtol=1e-6)
        assert torch.equal(out["valid"][r0:r0 + n].bool(), ok)
        midx, mlab = iou_match(gtb[i].to(DEV), out["nb"][r0:r0 + n], 0.7)
        if m:
            cls = gtc[i].to(DEV)[midx]
            cls[mlab == 0] = C
            src = gsrc[i].to(DEV)[midx]
            src[mlab == 0] = 0
            assert torch.equal(out["gtb"][r0:r0 + n], gtb[i].to(DEV)[midx])
        else:
            cls, src = torch.full((n,), C, device=DEV), torch.zeros(n, dtype=torch.int64, device=DEV)
        cls = torch.where(ok, cls, torch.full_like(cls, -1))
        assert torch.equal(out["cls"][r0:r0 + n], cls)
        assert torch.equal(out["src"][r0:r0 + n], src)
        nfg += int(((mlab == 1) & ok).sum())
        r0 += n
    assert int(out["nfg"]) == nfg and int((~out["valid"].bool()).sum()) > 0


@pytest.mark.parametrize("not_norm_reg,alpha,ihf,cared_all", [(True, 0.25, 0.85, False), (False, -1.0, 0.0, True), (False, 0.25, 0.85, False)])
def test_fused_centernet_losses_vs_torch_formulation(not_norm_reg, alpha, ihf, cared_all, monkeypatch):
    """dgx_centernet_losses vs the elementwise torch formulation of centernet.py:237-314 (values and gradients), incl.
    rows without a regression target, duplicate positive indices and un-cared positives."""
    import divergen_amd.modeling.dense_heads.centernet as CM
    from divergen_amd.utils.events import EventStorage
    g = torch.Generator().manual_seed(5)
    M, P = 5000, 37
    net = CM.CenterNet(in_channels=16, num_classes=7, with_agn_hm=True, only_proposal=True, reg_weight=2.0,
                       not_norm_reg=not_norm_reg, pos_weight=0.5, neg_weight=0.75, ignore_high_fp=ihf, hm_focal_alpha=alpha,
                       centernet_head=torch.nn.Identity()).to(DEV).train()
    reg_t = torch.rand(M, 4, generator=g) * 40
    reg_t[torch.rand(M, generator=g) < 0.7] = -1e8                       # INF-style "no target" rows
    reg_t[3] = torch.tensor([5.0, 5.0, 5.0, 5.0])
    hm = torch.rand(M, 1, generator=g) ** 3
    rp0 = torch.rand(M, 4, generator=g) * 40
    rp0[3] = torch.tensor([5.0, 7.0, 5.0, 2.0])                          # ties with the target: split gradient
    al0 = torch.randn(M, generator=g) * 4                                # includes logits clamped by SIGMOID_CLAMP
    idx = torch.randint(0, M, (P,), generator=g)
    idx[1] = idx[0]                                                      # duplicate positive location
    cared = torch.ones(P, dtype=torch.bool) if cared_all else torch.rand(P, generator=g) < 0.8
    res = []
    for fused in (True, False):
        monkeypatch.setattr(CM, "_FUSED_CN_LOSSES", fused)
        rp, al = rp0.clone().to(DEV).requires_grad_(), al0.clone().to(DEV).requires_grad_()
        with EventStorage(0):
            losses = net.losses((idx.to(DEV), cared.to(DEV)), reg_t.to(DEV), hm.to(DEV), rp, al)
        (losses["loss_centernet_loc"] * 1.3 + losses["loss_centernet_agn_pos"] * 0.7 + losses["loss_centernet_agn_neg"] * 1.9).backward()
        res.append(({k: float(v) for k, v in losses.items()}, rp.grad.clone(), al.grad.clone()))
    (lf, grf, gaf), (lt, grt, gat) = res
    for k in lt:
        assert abs(lf[k] - lt[k]) <= 1e-5 * max(1.0, abs(lt[k])), (k, lf[k], lt[k])
    torch.testing.assert_close(grf, grt, atol=1e-7, rtol=2e-4)
    torch.testing.assert_close(gaf, gat, atol=1e-7, rtol=2e-4)


def test_predict_instances_candidates_vs_reference_golden(golden):
    """Product decode (all levels at once, per-level top-k where needed) vs the reference's predict_single_level output
    (tests/golden/centernet_predict.npz): one level with more locations than the pre-NMS top-k, NMS off so that the
    candidate list itself is returned (scores are sqrt(heat map), centernet.py:702, in the golden as well)."""
    from divergen_amd.modeling.dense_heads.centernet import CenterNet
    g = golden("centernet_predict")
    hm, reg, grids = T(g["hm"]).to(DEV), T(g["reg"]).to(DEV), T(g["grids"]).to(DEV)
    net = CenterNet(in_channels=16, num_classes=1, in_features=("p4",), strides=(int(g["stride"]),), with_agn_hm=True,
                    only_proposal=True, score_thresh=float(g["thresh"]), pre_nms_topk_test=int(g["topk"]), post_nms_topk_test=100,
                    not_nms=True, sizes_of_interest=((0, 1e8),), centernet_head=torch.nn.Identity()).to(DEV).eval()
    res = net.predict_instances([grids], [hm], [reg], [(256, 320), (256, 320)])
    for i, r in enumerate(res):
        want_b, want_s = T(g["boxes%d" % i]).to(DEV), T(g["scores%d" % i]).to(DEV)
        assert len(r) == want_s.numel(), (i, len(r), want_s.numel())
        o = torch.argsort(r.scores, descending=True, stable=True)
        torch.testing.assert_close(r.scores[o], want_s, atol=0, rtol=0)
        torch.testing.assert_close(r.pred_boxes.tensor[o], want_b, atol=0, rtol=0)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_fused_proposal_decode_equals_composed_decode(dt):
    """CenterNet._predict_instances_fused (dgx_centernet_scores / _decode / _finalize around torch's top-k, sort and the NMS
    kernel) against predict_instances (the composed torch form, itself pinned on the reference's predict_single_level golden
    above): the same fixed-length boxes / scores / validity rows, bit for bit -- five levels, two of them larger than the pre-NMS
    top-k, channel-sliced channels-last maps as the head produces them, invalid tail included."""
    from divergen_amd.modeling.dense_heads.centernet import CenterNet
    g = torch.Generator().manual_seed(23)
    net = CenterNet(in_channels=16, num_classes=1, with_agn_hm=True, only_proposal=True, score_thresh=0.3, pre_nms_topk_train=300,
                    post_nms_topk_train=100, nms_thresh_train=0.9, centernet_head=torch.nn.Identity()).to(DEV).train()
    B, shapes = 2, [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    both = [(torch.randn(B, h, w, 8, generator=g) * 2).to(DEV).to(dt) for h, w in shapes]          # NHWC storage, 8 channels
    hm_logits = [t.permute(0, 3, 1, 2)[:, :1] for t in both]
    regs = [torch.relu(t.permute(0, 3, 1, 2)[:, 1:5].float() * 1.5 + 1.0).to(dt) for t in both]     # elementwise result: dense NHWC, C = 4
    sizes = [(256, 320), (250, 300)]
    fused = net._predict_instances_fused(hm_logits, regs, sizes)
    assert fused is not None
    grids = net.compute_grids(regs)
    ref = net.predict_instances(grids, [x.float().sigmoid() for x in hm_logits], [r.float() for r in regs], sizes)
    for a, b in zip(fused, ref):
        assert torch.equal(a.proposal_valid, b.proposal_valid) and 0 < int(a.proposal_valid.sum()) < a.proposal_valid.numel()
        assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor)
        assert torch.equal(a.scores, b.scores)
    assert fused.batch[0].shape == (B, 164, 4)


def test_drop_path_factors_survive_activation_checkpointing():
    """MODEL.SWIN.USE_CHECKPOINT: the recomputation of a block in backward must use the DropPath factors of the forward
    pass (drawn once per step for all blocks by SwinTransformer.forward): gradients with and without checkpointing agree."""
    from divergen_amd.modeling.backbone.swintransformer import SwinTransformer
    res = {}
    for ckpt in (False, True):
        torch.manual_seed(3)
        net = SwinTransformer(embed_dim=32, depths=[2, 2], num_heads=[1, 2], window_size=7, drop_path_rate=0.5,
                              out_indices=(0, 1), use_checkpoint=ckpt).to(DEV).train()
        torch.manual_seed(17)                      # same DropPath draw in both runs
        x = torch.randn(4, 3, 64, 64, device=DEV)
        outs = net(x)
        sum(o.float().square().mean() for o in outs.values()).backward()
        res[ckpt] = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    assert res[False].keys() == res[True].keys() and len(res[True]) > 10
    for k in res[False]:
        a, b = res[False][k], res[True][k]
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-7, k


def test_lazy_zero_grad_first_writer_gradients_equal_zero_filled_ones(monkeypatch):
    """FlatArena.zero_grad(lazy=True): from the second step on the Swin block's Linear weight / bias gradient segments are left
    to their first writer (the grouped weight-gradient launch with beta = 0) instead of being zero-filled and accumulated into.
    Over four steps with different inputs the whole gradient arena must equal that of the zero-filled run bit for bit, the
    segments must really have been skipped, a pass that goes through the OTHER (composed) path finds its segments zeroed, and a
    step in which the layer does not run at all leaves zeros, not the previous step's gradient."""
    from divergen_amd import solver
    from divergen_amd.layers import shift_regions
    from divergen_amd.modeling.backbone import swintransformer as S
    dim, nH, ws, B, H, W = 192, 6, 12, 2, 30, 26

    def run(lazy, composed_at=None, skip_at=None):
        monkeypatch.setattr(solver, "_LAZY_ZERO", lazy)
        torch.manual_seed(5)
        blk = S.SwinTransformerBlock(dim, nH, window_size=ws, shift_size=6, drop_path=0.0).to(DEV).train()
        for p in blk.parameters():
            torch.nn.init.normal_(p, std=0.05)
        blk.H, blk.W = H, W
        arena = solver.FlatArena(blk)
        region = shift_regions(H, W, ws).to(DEV)
        gen = torch.Generator(device=DEV).manual_seed(9)
        grads, skipped = [], []
        for step in range(4):
            arena.zero_grad(lazy=True)
            skipped.append(len(arena._lazy_pending))
            monkeypatch.setattr(S, "_FUSED_BLOCK", step != composed_at)
            x = torch.randn(B, H * W, dim, device=DEV, generator=gen).to(torch.bfloat16).requires_grad_(True)
            go = torch.randn(B, H * W, dim, device=DEV, generator=gen).to(torch.bfloat16)
            if step != skip_at:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = blk(x, region)
                y.backward(go)
            arena.finish_grads()
            grads.append(arena.g.clone())
        monkeypatch.setattr(S, "_FUSED_BLOCK", True)
        lin = torch.zeros(arena.numel, dtype=torch.bool, device=DEV)      # the Linear weights / biases: deterministic kernels
        for n_, o, q in zip(arena.names, arena.offsets, arena.params):
            if any(k in n_ for k in ("qkv", "proj", "fc1", "fc2")):
                lin[o:o + q.numel()] = True
        return grads, skipped, lin

    def same(a, b, lin):
        # Linear segments bit for bit; the rest (LayerNorm parameters, the relative-position table: fp32 atomics) to summation order
        return torch.equal(a[lin], b[lin]) and float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())
    ref, sk0, lin = run(False)
    got, sk1, _ = run(True)
    assert sk0 == [0, 0, 0, 0] and sk1[0] == 0 and sk1[1] == sk1[2] == sk1[3] == 8          # qkv, proj, fc1, fc2: weight + bias
    for a, b in zip(ref, got):
        assert float(a[lin].abs().max()) > 0 and same(a, b, lin)
    ref_c, _, _ = run(False, composed_at=2)
    got_c, sk, _ = run(True, composed_at=2)
    assert sk[2] == 8 and sk[3] == 0          # step 2 went the other way: its segments are no longer taken to be written directly
    for a, b in zip(ref_c, got_c):
        assert same(a, b, lin)
    got_s, _, _ = run(True, skip_at=2)
    assert float(got_s[2].abs().max()) == 0.0 and same(got_s[1], ref[1], lin)
