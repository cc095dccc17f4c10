"""GPU tests of the assembled model: loss keys / finiteness, eager vs hipGraph-replayed head segments, the whole training forward
against the assembled oracle, and the fused optimizer step through the arenas."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(graph, swin="T"):
    from divergen_amd.config import get_cfg
    from divergen_amd.modeling import build_model
    from divergen_amd.modeling.backbone.swintransformer import DropPath
    from divergen_amd.solver import build_optimizer
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "DiverGen_swinL.yaml"))
    cfg.merge_from_list(["MODEL.SWIN.SIZE", swin, "MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH",
                         os.path.join(ROOT, "configs", "metadata", "ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json")])
    torch.manual_seed(42)
    model = build_model(cfg).train()
    for m in model.modules():
        if isinstance(m, DropPath):
            m.drop_prob = 0.0
    return cfg, model, build_optimizer(cfg, model)


def test_training_step_losses_and_update():
    from divergen_amd.data import synthetic_batch
    from divergen_amd.utils.events import EventStorage
    cfg, model, opt = _build(False)
    batch = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
    p_before = opt.arena.p.clone()
    with EventStorage(0) as st:
        opt.zero_grad()
        losses = model(batch)
        total = sum(losses.values())
        total.backward()
        opt.step()
        torch.cuda.synchronize()
        assert st.latest()["roi_head/num_fg_samples"][0] > 0
    assert set(losses) == {"loss_cls_stage0", "loss_box_reg_stage0", "loss_cls_stage1", "loss_box_reg_stage1",
                           "loss_cls_stage2", "loss_box_reg_stage2", "loss_mask", "loss_centernet_loc",
                           "loss_centernet_agn_pos", "loss_centernet_agn_neg"}
    assert all(bool(torch.isfinite(v)) for v in losses.values())
    assert bool(torch.isfinite(opt.arena.g).all()) and float(opt.arena.g.abs().sum()) > 0
    moved = (opt.arena.p - p_before).abs().max()
    assert 0 < float(moved) <= 1.01 * opt.param_groups[0]["lr"] * 1.1 + 1e-3   # AdamW step bounded by ~lr
    # bf16 shadow follows the fp32 weights; EMA moved by (1-decay) of the pre-step weights
    assert torch.equal(opt.arena.p16, opt.arena.p.to(torch.bfloat16))
    assert torch.allclose(opt.ema, p_before, rtol=1e-6, atol=1e-7)   # ema(p0, p0) = p0 (fp32 rounding) at the first step


def test_graphed_head_segments_match_eager(monkeypatch):
    """FPN top-down + CenterNet tower replayed as hipGraphs (utils/graphs.py) vs issued eagerly: same losses and
    same gradients (fp32 atomics / split-sum order only), on the first (capturing) AND on a later (replaying) step."""
    from divergen_amd.data import synthetic_batch
    from divergen_amd.utils import graphs
    from divergen_amd.utils.events import EventStorage
    res = {}
    for on in (False, True):
        monkeypatch.setattr(graphs, "ENABLED", on)
        cfg, model, opt = _build(False)
        batch = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
        outs = []
        with EventStorage(0):
            for it in range(3):
                torch.manual_seed(100 + it)      # same sampling RNG in both runs
                opt.zero_grad()
                losses = model(batch)
                sum(losses.values()).backward()
                torch.cuda.synchronize()
                outs.append(({k: float(v) for k, v in losses.items()}, opt.arena.g.clone()))
        res[on] = outs
    for it in range(3):
        l0, g0 = res[False][it]
        l1, g1 = res[True][it]
        for k in l0:
            assert abs(l0[k] - l1[k]) <= 2e-3 * abs(l0[k]) + 1e-5, (it, k, l0[k], l1[k])
        # two separate runs: fp32 atomics (ROIAlign / bias-table scatters) and bf16 rounding reorder sums; a borderline
        # discrete decision (NMS / matching) may flip for a single RoI -> bound the worst element at 10 % of the scale
        assert float((g0 - g1).abs().max()) <= 1e-1 * float(g0.abs().max()), it


def test_graphed_segments_with_several_batch_sizes(monkeypatch):
    """The real loader hands over batches of MANY padded sizes (EfficientDetResizeCrop).  Two sizes in alternation with graphs on:
    the first sight of a size runs eagerly, the second captures, later ones replay THAT size's graphs (rounds 1-5 captured at first
    sight and died on the second size: make_graphed_callables had replaced the module's forward); a third size beyond MAX_GRAPHS
    stays eager.  Losses equal the graph-free run's."""
    from divergen_amd.data import synthetic_batch
    from divergen_amd.utils import graphs
    from divergen_amd.utils.events import EventStorage
    monkeypatch.setattr(graphs, "MAX_GRAPHS", 2)
    sizes = [256, 320, 256, 320, 256, 320, 192, 192, 192, 256]
    res = {}
    for on in (False, True):
        monkeypatch.setattr(graphs, "ENABLED", on)
        cfg, model, opt = _build(False)
        batches = {s_: synthetic_batch(2, s_, cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=s_, device="cuda") for s_ in set(sizes)}
        outs = []
        with EventStorage(0):
            for it, s_ in enumerate(sizes):
                torch.manual_seed(100 + it)
                opt.zero_grad()
                losses = model(batches[s_])
                sum(losses.values()).backward()
                torch.cuda.synchronize()
                outs.append({k: float(v) for k, v in losses.items()})
        res[on] = outs
        if on:
            seg = model.backbone.__dict__["_segment"]
            assert len(seg._fns) == 2 and max(seg._seen.values()) >= 2, (len(seg._fns), seg._seen)      # 256 and 320 captured, 192 eager
    for it in range(len(sizes)):
        for k in res[False][it]:
            a, b = res[False][it][k], res[True][it][k]
            assert abs(a - b) <= 2e-3 * abs(a) + 1e-5, (it, sizes[it], k, a, b)


LIBRARY_COMPUTE = ("Cijk_", "rocblas", "hipblas", "miopen", "MIOpen", "naive_conv", "igemm", "layer_norm", "LayerNorm", "group_norm",
                   "GroupNorm", "batch_norm", "gelu", "Gelu", "GELU", "max_pool", "softmax", "Softmax")


def library_compute_kernels(prof):
    """Names of launched kernels that belong to a vendor / framework COMPUTE routine (GEMM, convolution, normalisation,
    activation, pooling, softmax): the product path must not launch any -- every such op is a libdgx kernel."""
    names = set()
    for ev in prof.events():
        for k in ev.kernels:
            if any(t in k.name for t in LIBRARY_COMPUTE):
                names.add(k.name[:120])
    own = ("gemm_nt_kernel", "gemm_lw_kernel", "gemm_splitk_fold_kernel", "wgrad256", "win_attn", "ln_fwd_kernel", "ln_bwd_kernel", "pm_ln_", "gn_",
           "gelu_fwd_kernel", "gelu_bwd_colsum_kernel", "gelu_colsum_final_kernel", "maxpool3x3s2")
    return sorted(n for n in names if not any(o in n for o in own))


# Gradient parity per LOSS GROUP.  The oracle runs with the product's bf16 storage points AND the product's ReLU
# on/off patterns (oracle/quant.py relu_masks: a ReLU decision is a discrete intermediate like a proposal box or a matched label).
# Round 5 asserted 8 % / 25 % and explained the distance by flipped ReLU masks; handing the patterns over shows that this IS the
# distance: the oracle with its own decisions differs from the oracle with the product's by 5.8 % (CenterNet group: four
# GroupNorm + ReLU tower layers), 1.3 % (box group) and 0.02 % (mask group) over the arena -- 0.3 % of the elements flip, because the
# two pipelines' activations differ at the bf16 noise floor (~1e-3) by the time they reach the heads, and a bias / weight gradient is
# a signed sum over the elements, so a fraction f of flips moves it by ~sqrt(2 f).  With the patterns shared what is left is
# rounding and summation order: measured (profiles/r06_grad_parity_by_loss_group.txt) arena 0.51 / 0.64 / 0.05 / 0.54 %, worst
# segment 1.25 / 0.97 / 0.55 / 1.24 %, worst parameter 1.4 % (relative-position tables, |g| ~ 1e-4 of the arena).
# Bounds = measured x 1.5, per segment <= 2 % (the verdict's target); a parameter or segment whose gradient is < 2e-3 of the arena's
# norm is bounded through its share of the whole instead.
GRAD_GROUPS = {
    "centernet": (lambda k: "centernet" in k, 8e-3, 2.2e-2, 2.0e-2),
    "box": (lambda k: "stage" in k, 1.0e-2, 2.0e-2, 2.0e-2),
    "mask": (lambda k: k == "loss_mask", 1.0e-3, 2.0e-2, 1.0e-2),
    "all": (None, 8.5e-3, 2.2e-2, 2.0e-2),
}


@pytest.mark.parametrize("group", list(GRAD_GROUPS))
def test_end_to_end_losses_and_gradients_vs_assembled_oracle(monkeypatch, group):
    """All ten losses within 1e-3 of both oracles (every group runs the same forward), the group's gradient -- d(sum of the
    group's losses) / d(every parameter) -- within the bounds above: whole arena, every segment (backbone stage / FPN conv / head),
    every parameter.  'mask' is the path mask head -> RoIAlign gather backward -> FPN -> whole Swin backbone; 'box' the three
    cascade stages (fused box stage, _ScaleGradient, pooler backward); 'centernet' the tower + losses + the early backward."""
    f, arena, per_param, per_segment = GRAD_GROUPS[group]
    run_e2e_vs_oracle(monkeypatch, "T", 256, grads=True, loss_filter=f, grad_bounds=(arena, per_param), segment_bounds=per_segment)


def _capture_relu_sites(monkeypatch, model):
    """The product's ReLU on/off patterns of one training forward, by site: FPN top block (p6), the CenterNet tower's GroupNorm +
    ReLU layers per level, the regression ReLU, the box heads' two FC ReLUs per cascade stage, the mask head's convolutions and
    deconvolution.  Read from the OUTPUTS the product stores (out > 0).  Returned dict is filled during the forward."""
    import divergen_amd.layers.box_stage as BS
    import divergen_amd.modeling.dense_heads.centernet_head as CH
    site = {"tower": [], "reg": None, "box": [], "mask": {}, "p6": None}
    orig_gn, orig_out, orig_act = CH.groupnorm_relu_multi, CH.centernet_head_outputs, BS.G.gemm_nt_act

    def gn(xs, *a, **k):
        out = orig_gn(xs, *a, **k)
        site["tower"].append([(o > 0).detach().cpu() for o in out])
        return out

    def head_out(boths, scales):
        reg, hm = orig_out(boths, scales)
        site["reg"] = ((reg > 0).detach().cpu(), [tuple(b.shape) for b in boths])
        return reg, hm

    def act(x, w, b, relu=False):
        out = orig_act(x, w, b, relu=relu)
        if relu and sys._getframe(1).f_code.co_filename.endswith("box_stage.py"):      # (the mask head's deconvolution is a ReLU GEMM too)
            site["box"].append((out > 0).detach().cpu())
        return out
    monkeypatch.setattr(CH, "groupnorm_relu_multi", gn)
    monkeypatch.setattr(CH, "centernet_head_outputs", head_out)
    monkeypatch.setattr(BS.G, "gemm_nt_act", act)
    mh = model.roi_heads.mask_head
    for name in [n for n, _ in mh.named_children() if n.startswith("mask_fcn") and "relu" not in n] + ["deconv"]:
        getattr(mh, name).register_forward_hook(lambda m, i, o, name=name: site["mask"].__setitem__(name, (o > 0).detach().cpu()))
    model.backbone.register_forward_hook(lambda m, i, o: site.__setitem__("p6", (o["p6"] > 0).detach().cpu()))
    # the head called level by level (graphs off): its regression outputs per level, (N, 4, H, W)
    model.proposal_generator.centernet_head.register_forward_hook(
        lambda m, i, o: site.__setitem__("reg_levels", [(r > 0).detach().cpu() for r in o[1]]))
    return site


def _oracle_relu_masks(site, box_rows, n_mask_rows):
    """site (product patterns) -> {oracle site tag: mask in the oracle's shape and row convention} (oracle/quant.py relu_masks)."""
    m = {"top_block.p6": site["p6"]}
    assert len(site["tower"]) == 4 and len(site["box"]) == 6, (len(site["tower"]), len(site["box"]))
    assert site["reg"] is not None or site.get("reg_levels") is not None
    for i, per_level in enumerate(site["tower"]):
        for l, k in enumerate(per_level):
            m["tower.%d.%d" % (l, i)] = k
    if site["reg"] is not None:          # the flattened tail (levels stacked in (level, image, y, x) order)
        reg, shapes = site["reg"]
        off = 0
        for l, (N, _, H, W) in enumerate(shapes):
            m["reg.%d" % l] = reg[off:off + N * H * W].reshape(N, H, W, 4).permute(0, 3, 1, 2)
            off += N * H * W
        assert off == reg.shape[0]
    else:
        for l, k in enumerate(site["reg_levels"]):
            m["reg.%d" % l] = k
    for k in range(3):
        for j, fc in enumerate(("fc1", "fc2")):
            m["roi_heads.box_head.%d.%s" % (k, fc)] = site["box"][2 * k + j][box_rows[k]]
    for name, k in site["mask"].items():
        m["roi_heads.mask_head." + name] = k[:n_mask_rows]
    return m


def run_e2e_vs_oracle(monkeypatch, swin, size, grads=False, loss_filter=None, grad_bounds=(2e-2, 5e-2), segment_bounds=None):
    """Whole training forward of the PRODUCT path -- the one bench.py times: bf16 operands, fp32 accumulation, every GEMM /
    convolution / normalisation / attention / loss on libdgx kernels -- against the assembled CPU oracle (oracle/model.py,
    fp32) on the same operands: same batch, the oracle's Linear / convolution weights rounded to bf16 (the values the product's
    GEMMs read from the arena's bf16 shadow), the two random draws of the step replaced by one deterministic rule on both
    sides, and the DISCRETE intermediate results of the product handed to the oracle -- the proposal boxes (with near-tied
    scores the top-k / NMS survivor set is not stable under reordering; decoding is pinned separately by
    test_gpu_parity_modules / test_gpu_kernels) and each cascade stage's matched labels (a refined box within rounding of an
    IoU threshold would otherwise flip a label).  Everything continuous is the oracle's own.
    Asserted: every one of the 10 losses within 1e-3 (north_star) of the oracle run with the product's bf16 storage points
    (oracle/quant.py) AND within 1e-3 (+ 1e-5 absolute) of the plain fp32 oracle (round 5; the deltas are printed as a table), and NO
    vendor / framework compute kernel in the launch list of the step.
    grads=True: the PARAMETER GRADIENTS of the same step as well -- the whole gradient arena of the product (every hand-written
    backward: attention, LayerNorm, the grouped weight gradients, GroupNorm, RoIAlign's gather, the fused losses) against the
    autograd of the fp32 oracle on the same batch: relative L2 over the arena and per parameter, printed as a table."""
    import divergen_amd.modeling.roi_heads.detic_fast_rcnn as FR
    import divergen_amd.modeling.roi_heads.detic_roi_heads as RH
    from divergen_amd.data import synthetic_batch
    from divergen_amd.utils import graphs
    from divergen_amd.utils.events import EventStorage
    from torch.profiler import ProfilerActivity, profile

    from tests._recipes import assembled_oracle_losses, det_fed_mask, det_sample
    # the product's batch-level sampler with "the first k in index order" in place of the random permutation = det_sample on the
    # oracle side (tests/_recipes.py)
    monkeypatch.setattr(RH, "draw_permutation", lambda n, k, device: torch.arange(k, device=device))
    monkeypatch.setattr(FR, "fed_loss_class_mask", det_fed_mask)
    monkeypatch.setattr(graphs, "ENABLED", False)      # graph replay vs eager is test_graphed_head_segments_match_eager's subject
    cfg, model, opt = _build(False, swin)
    batch = synthetic_batch(2, size, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
    captured, stages = {}, {}
    orig = model.roi_heads.forward

    def spy(images, features, proposals, targets=None, **kw):
        captured["props"] = [p.proposal_boxes.tensor[p.proposal_valid].detach().cpu().float() if p.has("proposal_valid")
                             else p.proposal_boxes.tensor.detach().cpu().float() for p in proposals]
        return orig(images, features, proposals, targets, **kw)
    monkeypatch.setattr(model.roi_heads, "forward", spy)
    model.roi_heads.stage_observer = lambda k, d: stages.__setitem__(
        k, {n: (v.detach().cpu() if torch.is_tensor(v) else v) for n, v in d.items()})
    relu_site = _capture_relu_sites(monkeypatch, model) if grads else None
    with EventStorage(0):
        opt.zero_grad()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            losses = model(batch)
            sum(v for k, v in losses.items() if loss_filter is None or loss_filter(k)).backward()
            torch.cuda.synchronize()
    got = {k: float(v) for k, v in losses.items()}
    opt.arena.finish_grads()        # segments no first writer reached are zeroed now (what optimizer.step() does first)
    torch.cuda.synchronize()
    lib = library_compute_kernels(prof)
    assert not lib, "vendor / framework compute kernels on the product path: %s" % lib
    launched = {k.name for ev in prof.events() for k in ev.kernels}
    assert any(("gemm_nt_kernel" in n) or ("gemm_lw_kernel" in n) for n in launched), "own GEMM not launched"
    for must in ("wgrad256_partial_kernel", "win_attn_fwd_kernel", "win_attn_bwd_kernel", "ln_fwd_kernel", "gn_"):
        assert any(must in n for n in launched), "expected libdgx kernel not launched: %s" % must

    # ---- oracle side
    gemm_params = set()
    for mn, m in model.named_modules():
        if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            gemm_params.update("%s.%s" % (mn, n) for n, _ in m.named_parameters(recurse=False))
    p = {k: (v.detach().to(torch.bfloat16).float().cpu() if k in gemm_params else v.detach().cpu().float())
         for k, v in model.state_dict().items()}
    images = model.preprocess_image(batch).tensor.cpu().float()      # the normalised batch as PatchEmbed's GEMM reads it (bf16 values)
    gts = [dict(boxes=b["instances"].gt_boxes.tensor.cpu().float(), classes=b["instances"].gt_classes.cpu(),
                masks=b["instances"].gt_masks.tensor.cpu()) for b in batch]
    C = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    fw = model.roi_heads.box_predictor[0].freq_weight.cpu().float()
    # each stage's discrete labels, in the oracle's row convention (rows of empty refined boxes are DROPPED there, kept as
    # "ignore" rows here): keep mask over the rows still alive, classes and matched ground-truth index of the kept rows
    stage_labels = {}
    counts = stages[0]["counts"]
    alive = [torch.ones(n, dtype=torch.bool) for n in counts]
    box_rows = {0: torch.arange(int(sum(counts)))}          # the product's rows the oracle keeps, per stage
    for k in (1, 2):
        d, per, off = stages[k], [], 0
        box_rows[k] = []
        for i, n in enumerate(counts):
            v = d["valid"][off:off + n].bool()
            cls, gtb = d["gt_classes"][off:off + n], d["gt_boxes"][off:off + n]
            keep = v[alive[i]]
            sel = alive[i] & v
            idx = (gtb[sel][:, None, :] - gts[i]["boxes"][None]).abs().sum(-1).argmin(1) if len(gts[i]["boxes"]) else torch.zeros(int(sel.sum()), dtype=torch.int64)
            per.append((keep, cls[sel], idx))
            alive[i] = sel
            box_rows[k].append(off + sel.nonzero().squeeze(1))
            off += n
        stage_labels[k] = per
        box_rows[k] = torch.cat(box_rows[k])
    from oracle.quant import bf16_storage

    def oracle_losses():
        with torch.no_grad():
            w = assembled_oracle_losses(p, images, gts, [tuple(b["instances"].image_size) for b in batch], swin, C, fw,
                                        cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE, cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION,
                                        cfg.MODEL.ROI_BOX_HEAD.FED_LOSS_NUM_CAT, model.roi_heads.mask_weight,
                                        proposals=captured["props"], stage_labels=stage_labels)
        return {k: float(v) for k, v in w.items()}
    # (1) the oracle with the product's bf16 STORAGE points (oracle/quant.py: every tensor that crosses a kernel boundary rounded
    # to bf16 where the product stores bf16, fp32 wherever it accumulates): what is left is summation order -> north_star's 1e-3
    with bf16_storage():
        want = oracle_losses()
    # (2) the plain fp32 oracle: the bf16 deltas, listed separately, bounded at 1 %
    want32 = oracle_losses()
    assert set(got) == set(want) == set(want32), (sorted(got), sorted(want))
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-6)
    report = {k: dict(product=got[k], oracle_bf16_storage=want[k], rel=rel(got[k], want[k]), oracle_fp32=want32[k],
                      bf16_delta=rel(got[k], want32[k])) for k in sorted(got)}
    print("e2e parity report, bf16 HIP product path vs the oracle (Swin-%s, %d px):" % (swin, size))
    for k, r in report.items():
        print("  %-24s product %.7f | oracle with bf16 storage %.7f rel %.2e | fp32 oracle %.7f bf16 delta %.2e"
              % (k, r["product"], r["oracle_bf16_storage"], r["rel"], r["oracle_fp32"], r["bf16_delta"]))
    for k, r in report.items():
        assert abs(r["product"] - r["oracle_bf16_storage"]) <= 1e-3 * abs(r["oracle_bf16_storage"]) + 1e-6, (k, report)
        assert abs(r["product"] - r["oracle_fp32"]) <= 1e-3 * abs(r["oracle_fp32"]) + 1e-5, (k, report)
    if grads:
        n_fg = sum(int(((c >= 0) & (c < C)).sum()) for c in stages[0]["gt_classes"].split(list(counts)))
        masks = _oracle_relu_masks(relu_site, box_rows, n_fg)
        report["_grad_rows"] = _compare_gradients(model, p, relu=masks, segment_bounds=segment_bounds, oracle_losses_fn=lambda pp: assembled_oracle_losses(
            pp, images, gts, [tuple(b["instances"].image_size) for b in batch], swin, C, fw, cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
            cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION, cfg.MODEL.ROI_BOX_HEAD.FED_LOSS_NUM_CAT, model.roi_heads.mask_weight,
            proposals=captured["props"], stage_labels=stage_labels), swin=swin, size=size, loss_filter=loss_filter, bounds=grad_bounds)
    return report


def _compare_gradients(model, p, oracle_losses_fn, swin, size, loss_filter=None, bounds=(2e-2, 5e-2), relu=None, segment_bounds=None):
    """d(sum of the 10 losses) / d(parameter) for EVERY parameter: the product's gradient arena (bf16 operands, fp32 accumulation,
    hand-written backward kernels) against torch autograd through the oracle at the same point (the oracle's GEMM weights are the
    bf16-rounded values the product reads), twice:
      * the oracle with the product's bf16 STORAGE points (oracle/quant.py; autograd rounds the gradients at the same points on the
        way back) -- ASSERTED: what is left is summation order, so a wrong ring wrap, a dropped tile or a mis-scaled term in any
        backward kernel shows up as O(1) on its parameter;
      * the plain fp32 oracle -- REPORTED as the bf16 delta of the gradients.  It is large by construction, not by defect: a ReLU
        whose bf16-stored pre-activation lies within rounding of zero flips its mask, and a fraction f of flipped elements is a
        relative L2 error of sqrt(f) in the gradient behind it (~3-5 % per GroupNorm + ReLU layer of the CenterNet tower, measured
        layer by layer with tools/grad_parity_probe.py); the reference's own fp16 autocast has the same property against fp32."""
    from oracle.quant import bf16_storage, relu_masks

    def oracle_grads(storage, masks=None):
        pg = {k: v.clone().requires_grad_(True) for k, v in p.items() if v.is_floating_point()}
        pall = dict(p)
        pall.update(pg)
        with bf16_storage(storage), relu_masks(masks) as flips:
            total = sum(v for k, v in oracle_losses_fn(pall).items() if loss_filter is None or loss_filter(k))
            total.backward()
            flips = {k: sum(v) / len(v) for k, v in flips.items()}
        return {k: v.grad for k, v in pg.items()}, flips
    # ASSERTED: bf16 storage points AND the product's ReLU patterns (discrete intermediates, like proposals and labels)
    (ref_q, flips), (ref_f, _) = oracle_grads(True, relu), oracle_grads(False)
    if relu is not None:
        own, _ = oracle_grads(True)              # the oracle's own ReLU decisions: REPORTED (what the flips cost)
        n2 = sum(float((own[k] - ref_q[k]).double().square().sum()) for k in own if own[k] is not None and ref_q[k] is not None)
        d2 = sum(float(ref_q[k].double().square().sum()) for k in ref_q if ref_q[k] is not None)
        print("ReLU sites where the oracle's own sign differs from the product's pattern (fraction of elements): "
              + ", ".join("%s %.1e" % (k, v) for k, v in sorted(flips.items()) if v > 0))
        print("gradient of the oracle with its OWN ReLU decisions vs with the product's: relative L2 %.3e over the arena" % ((n2 / max(d2, 1e-300)) ** 0.5))
    rows, num, den, numf = [], 0.0, 0.0, 0.0
    for name, q in model.named_parameters():
        if not q.requires_grad or q.grad is None:
            continue
        g = q.grad.detach()
        if hasattr(q, "_dgx_sd_perm"):            # stored in another column order than its state-dict form (box heads' first FC)
            g = q._dgx_sd_perm[0](g)
        g = g.double().cpu()
        r = torch.zeros_like(g) if ref_q[name] is None else ref_q[name].double()
        rf = torch.zeros_like(g) if ref_f[name] is None else ref_f[name].double()
        d2, r2, f2 = float((g - r).square().sum()), float(r.square().sum()), float((g - rf).square().sum())
        num += d2
        den += r2
        numf += f2
        rel = lambda e2: (e2 / r2) ** 0.5 if r2 > 0 else (0.0 if e2 == 0 else float("inf"))
        rows.append((name, tuple(g.shape), r2 ** 0.5, rel(d2), rel(f2)))
    whole, whole_f = (num / den) ** 0.5, (numf / den) ** 0.5
    print("e2e gradient parity, product gradient arena vs the oracle's autograd (Swin-%s, %d px), relative L2 over all %d parameters: "
          "%.3e vs the oracle with bf16 storage, %.3e vs the fp32 oracle (|g| = %.4e)" % (swin, size, len(rows), whole, whole_f, den ** 0.5))
    for name, shape, nr, rel_q, rel_f in sorted(rows, key=lambda t: -t[3])[:12]:
        print("  %-64s %-20s |g| %.3e  rel L2 %.3e  (fp32 oracle: %.3e)" % (name, shape, nr, rel_q, rel_f))
    by_group = {}
    for name, shape, nr, rel_q, rel_f in rows:
        key = ".".join(name.split(".")[:4]) if name.startswith("backbone.bottom_up.layers") else ".".join(name.split(".")[:2])
        a = by_group.setdefault(key, [0.0, 0.0, 0.0])
        a[0] += (rel_q * nr) ** 2
        a[1] += nr ** 2
        a[2] += (rel_f * nr) ** 2
    for key, (d2, r2, f2) in sorted(by_group.items()):
        print("  segment %-44s |g| %.3e  rel L2 %.3e  (fp32 oracle: %.3e)" % (key, r2 ** 0.5, (d2 / r2) ** 0.5 if r2 > 0 else 0.0,
                                                                              (f2 / r2) ** 0.5 if r2 > 0 else 0.0))
    if bounds is None:
        return rows
    if segment_bounds is not None:              # per segment of the arena (backbone stage / FPN conv / head): relative L2
        for key, (d2, r2, f2) in sorted(by_group.items()):
            seg = (d2 / r2) ** 0.5 if r2 > 0 else 0.0
            assert seg <= segment_bounds or (d2 ** 0.5) <= 2e-3 * den ** 0.5, (key, seg)
    assert whole <= bounds[0], whole
    tot = den ** 0.5
    for name, shape, nr, rel_q, rel_f in rows:
        # per parameter; the tiny ones (a bias whose gradient is 1e-6 of the total) are bounded through their share of the whole
        assert rel_q <= bounds[1] or rel_q * nr <= 2e-3 * tot, (name, shape, nr, rel_q)
    return rows


def test_overfits_a_fixed_batch():
    """End-to-end sanity of forward + hand-written backward + fused optimizer: 40 steps on ONE synthetic batch (Swin-T,
    256 px, bf16 product path) must drive the total loss well below its starting value, with finite losses throughout."""
    from divergen_amd.data import synthetic_batch
    from divergen_amd.utils.events import EventStorage
    cfg, model, opt = _build(False)
    batch = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
    for g in opt.param_groups:
        g["lr"] = 2e-4
    hist = []
    with EventStorage(0):
        for it in range(40):
            torch.manual_seed(1000 + it)          # proposal sampling / federated-class draws differ per step, as in training
            opt.zero_grad()
            losses = model(batch)
            total = sum(losses.values())
            total.backward()
            opt.step()
            hist.append(float(total))
            assert hist[-1] == hist[-1] and abs(hist[-1]) < 1e4, (it, hist[-1])
    first, last = sum(hist[:3]) / 3, sum(hist[-3:]) / 3
    assert last < 0.6 * first, (first, last, hist)


def test_inference_path_end_to_end():
    """Evaluation forward (CustomRCNN.inference, custom_rcnn.py:87-115 -> cascade box inference, fast_rcnn_inference,
    mask inference, detector_postprocess) on a random-init Swin-T model: structural invariants of the results and agreement
    of the GPU run-length route with the bitmask route of the results writer."""
    from divergen_amd.data import synthetic_batch
    from divergen_amd.evaluation import instances_to_coco_json
    from divergen_amd.modeling.meta_arch.custom_rcnn import detector_postprocess
    cfg, model, opt = _build(False)
    model.eval()
    batch = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
    for b in batch:
        b["height"], b["width"] = 300, 380                     # results are reported at the ORIGINAL image size
    with torch.no_grad():
        out = model(batch)
        raw = model.inference(batch, do_postprocess=False)
    assert len(out) == 2
    C, top = cfg.MODEL.ROI_HEADS.NUM_CLASSES, cfg.TEST.DETECTIONS_PER_IMAGE
    for o, r in zip(out, raw):
        inst = o["instances"]
        assert inst.image_size == (300, 380)
        n = len(inst)
        assert 0 < n <= top
        bx = inst.pred_boxes.tensor
        assert bx.shape == (n, 4) and bool((bx[:, 2] > bx[:, 0]).all()) and bool((bx[:, 3] > bx[:, 1]).all())
        assert float(bx.min()) >= 0 and float(bx[:, 2].max()) <= 380 and float(bx[:, 3].max()) <= 300
        sc = inst.scores
        assert bool((sc[:-1] >= sc[1:]).all()) and 0 < float(sc.min()) and float(sc.max()) <= 1.0
        assert inst.pred_classes.dtype == torch.int64 and 0 <= int(inst.pred_classes.min()) and int(inst.pred_classes.max()) < C
        assert inst.pred_masks.shape == (n, 300, 380) and inst.pred_masks.dtype == torch.bool
        # same detections through the fused paste + run-length route
        a = instances_to_coco_json(inst, 7)
        b = instances_to_coco_json(detector_postprocess(r, 300, 380, mask_format="rle"), 7)
        assert len(a) == len(b) == n
        for x, y in zip(a, b):
            assert x["segmentation"] == y["segmentation"] and x["category_id"] == y["category_id"]
            assert abs(x["score"] - y["score"]) < 1e-7 and max(abs(p - q) for p, q in zip(x["bbox"], y["bbox"])) < 1e-4
    # the evaluator loop: results file in LVIS format (1-indexed category ids), model restored to its mode
    import json
    import tempfile
    from divergen_amd.evaluation import LVISResultsWriter, inference_on_dataset
    for i, b in enumerate(batch):
        b["image_id"] = 100 + i
    with tempfile.TemporaryDirectory() as d:
        model.train()
        res = inference_on_dataset(model, [batch], LVISResultsWriter(d, distributed=False))
        assert res == {} and model.training
        rows = json.load(open(os.path.join(d, "lvis_instances_results.json")))
    assert len(rows) == sum(len(o["instances"]) for o in out)
    assert {r["image_id"] for r in rows} == {100, 101} and min(r["category_id"] for r in rows) >= 1
    assert all(isinstance(r["segmentation"]["counts"], str) and r["segmentation"]["size"] == [300, 380] for r in rows)


def test_eval_only_flow_reports_lvis_ap(tmp_path, monkeypatch):
    """DG/train_net.py --eval-only -> do_test: test loader over a (tiny, generated) LVIS-format split, inference with GPU
    post-processing + run-length encoding, results json, box and mask AP from the LVISEval restatement."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import train_net
    from tests.test_host_data import _tiny_lvis
    root = _tiny_lvis(tmp_path)
    os.replace(os.path.join(root, "lvis", "lvis_v1_train.json"), os.path.join(root, "lvis", "lvis_v1_val.json"))
    monkeypatch.setenv("DETECTRON2_DATASETS", root)
    cfg, model, opt = _build(False)
    cfg.merge_from_list(["DATASETS.TEST", ("lvis_v1_val",), "OUTPUT_DIR", str(tmp_path / "out"), "DATALOADER.NUM_WORKERS", 0,
                         "INPUT.MIN_SIZE_TEST", 128, "INPUT.MAX_SIZE_TEST", 192, "MODEL.DEVICE", "cuda"])
    res = train_net.do_test(cfg, model)
    assert set(res) == {"bbox", "segm"}
    for task in ("bbox", "segm"):
        assert set(res[task]) == {"AP", "AP50", "AP75", "APs", "APm", "APl", "APr", "APc", "APf"}
        assert -100.0 <= res[task]["AP"] <= 100.0
    rows = json.load(open(tmp_path / "out" / "inference_lvis_v1_val" / "lvis_instances_results.json"))
    assert len(rows) > 0 and {r["image_id"] for r in rows} <= {1, 2, 3, 4, 5, 6}


def test_fused_sampler_equals_composed_sampler(monkeypatch):
    """label_and_sample_proposals as two batch-level kernels around the step's one device->host read (dgx_roi_label /
    dgx_roi_gather) against the composed per-image path (IoU match, masks, stable sorts, index gathers): the same
    torch.randperm draws in the same order, hence the same sampled rows -- every loss of the step bit-identical."""
    import divergen_amd.modeling.roi_heads.detic_roi_heads as RH
    from divergen_amd.data import synthetic_batch
    from divergen_amd.utils import graphs
    from divergen_amd.utils.events import EventStorage
    monkeypatch.setattr(graphs, "ENABLED", False)
    cfg, model, opt = _build(False)
    batch = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
    res = []
    used = []
    orig = RH.DeticCascadeROIHeads._label_and_sample_fused
    for fused in (True, False):
        def spy(self, proposals, targets, _f=fused):
            out = orig(self, proposals, targets) if _f else None
            used.append(out is not None)
            return out
        monkeypatch.setattr(RH.DeticCascadeROIHeads, "_label_and_sample_fused", spy)
        with EventStorage(0) as st:
            torch.manual_seed(123)
            losses = model(batch)
            torch.cuda.synchronize()
            res.append(({k: float(v) for k, v in losses.items()}, float(st.latest()["roi_head/num_fg_samples"][0])))
    assert used == [True, False]
    assert res[0] == res[1], res


def test_first_writer_gradients_train_like_zero_filled_ones(monkeypatch):
    """Three optimizer steps of the assembled model (Swin-T CenterNet2, two different batches) with the training loop's
    zero_grad(lazy=True) -- backbone Linear and box-cascade gradient segments left to their first writers -- against the same
    steps with every segment zero-filled: the same weights up to the summation order of the kernels that use atomics, and the
    lazy run really skipped segments."""
    from divergen_amd import solver
    from divergen_amd.data import synthetic_batch
    from divergen_amd.modeling.backbone import swintransformer as S
    from divergen_amd.utils.events import EventStorage
    # (the protocol lives in the eagerly issued blocks -- what every batch size beyond graphs.MAX_GRAPHS runs as; a replayed block group
    # was captured accumulating, beta = 1, and its segments are zero-filled like everything else: a captured launch cannot ask whether it
    # is the first writer of the pass)
    monkeypatch.setattr(S, "GRAPH_BLOCKS", False)

    def run(lazy):
        monkeypatch.setattr(solver, "_LAZY_ZERO", lazy)
        cfg, model, opt = _build(False)
        batches = [synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=11 + k, device="cuda") for k in range(2)]
        skipped = []
        with EventStorage(0):
            for k in range(3):
                opt.zero_grad()
                skipped.append(len(opt.arena._lazy_pending))
                torch.manual_seed(100 + k)
                losses = model(batches[k % 2])
                sum(losses.values()).backward()
                opt.step()
        return opt.arena.p.clone(), skipped, opt.arena
    p0, s0, _ = run(False)
    p1, s1, arena = run(True)
    assert s0 == [0, 0, 0] and s1[0] == 0 and s1[1] == s1[2] > 40, (s0, s1)
    assert float((p0 - p1).abs().max()) <= 2e-3 * float(p0.abs().max())
    names = {arena.names[i] for i in arena.direct}
    assert any("box_head" in n for n in names) and any("mlp.fc1.weight" in n for n in names)


def test_early_proposal_backward_contract_is_enforced():
    """the early mode is correct only for ONE backward of the plain sum per forward.  The loss dict is marked
    (EarlyLosses), engine.total_loss() consumes it once, and a forward while an unconsumed dict is outstanding raises."""
    from divergen_amd.data import synthetic_batch
    from divergen_amd.engine import total_loss
    from divergen_amd.modeling.meta_arch.custom_rcnn import EarlyLosses
    from divergen_amd.utils.events import EventStorage
    cfg, model, opt = _build(False)
    model.early_proposal_backward = True
    batch = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
    with EventStorage(0):
        opt.zero_grad()
        losses = model(batch)
        assert isinstance(losses, EarlyLosses) and not losses.consumed
        with pytest.raises(RuntimeError, match="never passed to engine.total_loss"):
            model(batch)                               # a second forward before the first one's backward
        total_loss(losses).backward()
        with pytest.raises(RuntimeError, match="summed before"):
            total_loss(losses)
        opt.zero_grad()
        total_loss(model(batch)).backward()            # the ordinary sequence goes on working
        model.early_proposal_backward = False
        assert not isinstance(model(batch), EarlyLosses)
    torch.cuda.synchronize()


def test_early_proposal_backward_is_the_same_step():
    """model.early_proposal_backward (the training loops switch it on: the proposal generator's losses are back-propagated from inside the
    forward, ahead of the RoI heads' device->host read, the box cascade's right behind its forward): the forward is untouched -- every loss bit-identical, the same random draws --,
    the proposal losses come back detached, the parameters whose gradient does not pass through the FPN levels (CenterNet head, RoI
    heads) get bit-identical gradients, and backbone + FPN differ only by the association of the per-level sum of the consumers'
    gradients (bf16 maps: the poolers add onto the head's map instead of the head's map being added last)."""
    from divergen_amd.data import synthetic_batch
    from divergen_amd.engine import total_loss
    from divergen_amd.utils.events import EventStorage
    res = []
    for early, box in ((False, False), (True, False), (True, True)):
        cfg, model, opt = _build(True)
        model.early_proposal_backward, model.early_box_backward = early, box
        batch = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
        with EventStorage(0):
            for k in range(3):          # the second pass captures the segments (graphs.CAPTURE_AFTER), the third replays them
                torch.manual_seed(7)
                opt.zero_grad()
                losses = model(batch)
                total_loss(losses).backward()
                torch.cuda.synchronize()
        # early: the proposal generator's losses (and with early_box_backward the box cascade's) come back detached
        assert all(v.requires_grad != early for k, v in losses.items() if "centernet" in k)
        assert all(v.requires_grad != (early and box) for k, v in losses.items() if "stage" in k)
        assert losses["loss_mask"].requires_grad
        res.append(({k: float(v) for k, v in losses.items()}, opt.arena.g.clone(), opt.arena))
    (l0, g0, arena), (l1, g1, _), (l2, g2, _) = res
    assert l0 == l1 == l2, (l0, l1, l2)
    up = torch.zeros(g0.numel(), dtype=torch.bool, device="cuda")          # upstream of the FPN levels
    tables = torch.zeros_like(up)
    for n, o, z in zip(arena.names, arena.offsets, arena.sizes):
        if n.startswith("backbone."):
            up[o:o + z] = True
        if "relative_position_bias_table" in n:
            tables[o:o + z] = True
    assert bool(up.any()) and bool((~up).any())
    for g in (g1, g2):
        assert torch.equal(g0[~up], g[~up])
        a, b = g0[up & ~tables].double(), g[up & ~tables].double()
        # measured 8e-3: one-ulp differences of the bf16 level gradients, carried through the FPN and the backbone's backward in bf16 storage
        assert float((a - b).norm() / a.norm()) <= 2e-2
        assert float(a.norm()) > 0


def test_overlapped_transposes_are_the_same_training():
    """solver.OVERLAP_TRANSPOSES (train_net.py and bench.py switch it on): the transposed weight images -- B operands of the input-gradient
    GEMMs and of the mask head's deconvolution -- are refreshed on a side stream behind the optimizer kernel and joined behind the next
    backbone forward.  Losses and weights over several steps equal the run that refreshes them on the step's own stream bit for bit, with
    the early backward (which reads them from inside the forward) on."""
    from divergen_amd import solver
    from divergen_amd.data import synthetic_batch
    from divergen_amd.engine import total_loss
    from divergen_amd.utils.events import EventStorage
    res = []
    try:
        for overlap in (False, True):
            solver.OVERLAP_TRANSPOSES = overlap
            cfg, model, opt = _build(True)
            model.early_proposal_backward = model.early_box_backward = True
            batches = [synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda", seed=s) for s in (1, 2)]
            hist = []
            with EventStorage(0):
                for k in range(6):
                    torch.manual_seed(11 + k)
                    opt.zero_grad()
                    losses = model(batches[k % 2])
                    total_loss(losses).backward()
                    hist.append({n: float(v) for n, v in losses.items()})
                    opt.step()
            torch.cuda.synchronize()
            assert ("_tstream" in opt.arena.__dict__) == overlap and not solver._pending_transposes[1:]
            solver.join_transposes()
            res.append((hist, opt.arena.p.clone(), opt.arena.p16t.clone()))
    finally:
        solver.OVERLAP_TRANSPOSES = False
    (h0, p0, t0), (h1, p1, t1) = res
    assert h0 == h1
    assert torch.equal(p0, p1) and torch.equal(t0, t1)
