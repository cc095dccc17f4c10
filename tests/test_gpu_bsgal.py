"""GPU parity of the BSGAL gradient-bank kernels vs the reference's outputs (golden) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from divergen_amd import _lib  # noqa: E402
from divergen_amd.engine import bsgal as BG  # noqa: E402
from oracle import bsgal as B  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden", "bsgal_bank.npz")
DEV = "cuda"


class _Arena:
    def __init__(self, n):
        self.g = torch.zeros(n, device=DEV)
        self.p = torch.zeros(n, device=DEV)

    def zero_grad(self):
        self.g.zero_()

    def sync_shadow(self):
        self.synced = True


@pytest.mark.parametrize("mode", ["AVERAGE", "MOMENTUM0.9"])
def test_bank_update_vs_reference_golden(mode):
    z = np.load(G)
    grads = z["%s_grads" % mode]
    bank = BG.GradBank(_Arena(grads.shape[1]), update=mode)
    want_dev = np.zeros(grads.shape[1], np.float32)
    for it in range(grads.shape[0]):
        out = bank.update(torch.from_numpy(grads[it]).to(DEV), it + 1)
        assert out.data_ptr() == bank.bank.data_ptr()
        want_dev = B.update_grad_bank(want_dev, grads[it], it + 1, mode, reciprocal=True)
        got = out.cpu().numpy()
        assert np.array_equal(got, want_dev), (mode, it)                         # bit-exact vs the device semantics
        ref = z["%s_bank_%d" % (mode, it)]                                        # the reference's own (CPU) output
        if "MOMENTUM" in mode:
            assert np.array_equal(got, ref)
        else:
            assert np.abs(got - ref).max() <= 2.0 ** -22 * np.abs(grads).max()
    probe = torch.from_numpy(z["%s_probe" % mode]).to(DEV)
    assert abs(float(bank.similarity(probe)) - float(z["%s_sim_norm" % mode])) < 1e-6
    raw = float(z["%s_sim_raw" % mode])
    assert abs(float(bank.similarity(probe, norm=False)) - raw) < 1e-5 * max(1.0, abs(raw))


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 65536 + 1, 3_000_001])
def test_grad_sim_vs_oracle_sizes(n):
    g = torch.Generator().manual_seed(n)
    a, b = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3 + 0.1
    o3, o4 = BG.grad_sim(a.to(DEV), b.to(DEV))
    a64, b64 = a.double().numpy(), b.double().numpy()
    want = np.array([(a64 * b64).sum(), (a64 * a64).sum(), (b64 * b64).sum()])
    assert np.allclose(o3.cpu().numpy(), want, rtol=1e-12, atol=1e-12)
    assert abs(float(o4[3]) - B.compute_grad_sim(a.numpy(), b.numpy(), True)) < 1e-6
    # deterministic: a second launch gives the same bits
    o3b, _ = BG.grad_sim(a.to(DEV), b.to(DEV))
    assert torch.equal(o3, o3b)


def test_arena_scale_properties():
    """Swin-L-sized arena (197 M floats): linearity of the bank and Cauchy-Schwarz / self-similarity of the score."""
    n = 197_000_003
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn(n, device=DEV, generator=g)
    bank = BG.GradBank(_Arena(n), update="MOMENTUM0.5")
    bank.update(a, 1)
    assert torch.equal(bank.bank, a * 0.5)
    bank.update(a, 2)
    assert torch.equal(bank.bank, a * 0.5 * 0.5 + a * 0.5)
    assert abs(float(bank.similarity(a)) - 1.0) < 1e-6
    assert abs(float(bank.similarity(-a)) + 1.0) < 1e-6
    o3, o4 = BG.grad_sim(a, bank.bank)
    assert abs(float(o3[1]) / n - 1.0) < 1e-3 and float(o3[0]) ** 2 <= float(o3[1]) * float(o3[2]) * (1 + 1e-12)


def test_empty_and_bad_args():
    lib = _lib.lib()
    o3 = torch.zeros(3, dtype=torch.float64, device=DEV)
    o4 = torch.zeros(4, device=DEV)
    ws = torch.zeros(64, dtype=torch.uint8, device=DEV)
    assert lib.dgx_grad_sim(None, None, 0, o3.data_ptr(), o4.data_ptr(), ws.data_ptr(), _lib.stream()) == 0
    torch.cuda.synchronize()
    assert o3.tolist() == [0.0, 0.0, 0.0] and float(o4[3]) == 0.0
    assert lib.dgx_grad_sim(None, None, 8, o3.data_ptr(), o4.data_ptr(), ws.data_ptr(), _lib.stream()) != 0
    x = torch.zeros(9, device=DEV)
    assert lib.dgx_grad_bank_update(x.data_ptr() + 4, x.data_ptr(), 4, 0.5, 0.5, _lib.stream()) != 0    # misaligned
    with pytest.raises(NotImplementedError):
        BG.GradBank(_Arena(4), update="MEDIAN")


def test_loss_grad_decision_and_weight_snapshot_on_a_model():
    """get_loss_grad + paste_or_ori decision on a real FlatArena: gradients of the same batch agree (cosine 1), of an
    unrelated objective less so; WeightSnapshot undoes an optimizer step exactly."""
    from divergen_amd.solver import FlatArena
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4)).to(DEV)
    arena = FlatArena(net)
    bank = BG.GradBank(arena, update="AVERAGE")
    x, y = torch.randn(64, 16, device=DEV), torch.randn(64, 4, device=DEV)
    held = bank.loss_grad({"l": ((net(x) - y) ** 2).mean()})
    ref = torch.cat([p.grad.flatten() for p in net.parameters()])
    assert torch.equal(held[:ref.numel()][: net[0].weight.numel()], net[0].weight.grad.flatten())
    bank.update(held, 1)                                     # it=1: bank = 0*1/2 + held/2
    same = bank.loss_grad({"l": ((net(x) - y) ** 2).mean()})
    other = bank.loss_grad({"l": (net(torch.randn(64, 16, device=DEV)) ** 2).mean()})
    assert abs(float(bank.similarity(same)) - 1.0) < 1e-5
    assert float(bank.similarity(other)) < 0.999
    assert bool(bank.paste_is_better(same, other)) and not bool(bank.paste_is_better(other, same))
    snap = BG.WeightSnapshot(arena)
    before = [p.detach().clone() for p in net.parameters()]
    with torch.no_grad():
        arena.p.add_(arena.g, alpha=-0.1)                    # trial update (update_with_loss)
    assert not torch.equal(net[0].weight, before[0])
    snap.restore()
    for p, q in zip(net.parameters(), before):
        assert torch.equal(p, q)


class _TinyModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = torch.nn.Linear(4, 1, bias=False)
        with torch.no_grad():
            self.backbone.weight.zero_()

    def training_losses(self, batch):
        x = torch.stack([d["image"] for d in batch])
        y = torch.stack([d["instances"].target[0] for d in batch])
        return {"loss_cls_stage0": ((self.backbone(x).squeeze(1) - y) ** 2).mean()}


def test_gradient_comparison_rule_on_known_gradients():
    """ACTIVE_GRAD_COMPARE (custom_rcnn.py:345-355, :447-460, :592-603) with gradients known in closed form: at w = 0 the
    gradient of (w.x - y)^2 is -2 y x.  Held-out x = e0; a pasted batch along e0 has cosine 1 with it, an original batch along
    e1 cosine 0 -> paste; swapped -> original.  The bank (ACTIVE_GRAD_SAVE, AVERAGE) holds the running mean of the held-out
    gradients, weights untouched, gradients left zeroed."""
    from divergen_amd.solver import FlatArena
    from divergen_amd.structures import Instances
    model = _TinyModel().to(DEV).train()
    arena = FlatArena(model)
    sel = BG.ActiveSelector(model, arena, model.training_losses, mode="paste_or_ori", grad_compare=True, grad_save=True,
                            grad_update="AVERAGE", loss="cls")

    def sample(vec, y=1.0):
        inst = Instances((1, 1))
        inst.target = torch.tensor([y], device=DEV)
        return inst, torch.tensor(vec, device=DEV)

    def batch(p, o):
        pi, px = sample(p)
        oi, ox = sample(o)
        ti, tx = sample([1.0, 0.0, 0.0, 0.0])
        return [{"image": px, "instances": pi, "origin_image": ox, "origin_instances": oi, "test_image": tx, "test_instances": ti}]
    e0, e1 = [2.0, 0.0, 0.0, 0.0], [0.0, 3.0, 0.0, 0.0]
    chosen, paste = sel.select(batch(e0, e1))
    assert paste and abs(float(sel.last["sim_paste_init"]) - 1.0) < 1e-6 and abs(float(sel.last["sim_ori_init"])) < 1e-6
    assert torch.equal(sel.bank.bank[:4], torch.tensor([-2.0, 0.0, 0.0, 0.0], device=DEV))       # iter 0: bank = 0*bank + 1*grad, grad = -2 y x
    chosen, paste = sel.select(batch(e1, e0))
    assert not paste and torch.equal(chosen[0]["image"], torch.tensor(e0, device=DEV))
    assert float(arena.p.abs().sum()) == 0.0 and float(arena.g.abs().sum()) == 0.0
    assert (sel.paste_count, sel.not_paste_count) == (1, 1)


@pytest.mark.parametrize("grad_compare", [False, True])
def test_active_selection_inside_the_training_forward(grad_compare):
    """BSGAL's outer loop end to end on the registry-built model (Swin-T CenterNet2, 256 px, BS/configs/BSGAL/BSGAL_SwinL.yaml):
    the forward receives the pasted sample with its original and a held-out image, ActiveSelector runs its trial passes on
    the HIP path (backbone in eval mode; trial SGD step + restore, or three flattened gradients + bank + cosine), the weights
    come back bit-identical, the gradient arena is clean, and the step then trains on the chosen batch."""
    from divergen_amd.config import add_bsgal_config, get_cfg
    from divergen_amd.data import synthetic_batch
    from divergen_amd.modeling import build_model
    from divergen_amd.solver import build_optimizer
    from divergen_amd.utils.events import EventStorage
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = add_bsgal_config(get_cfg())
    cfg.merge_from_file(os.path.join(root, "configs", "BSGAL", "BSGAL_SwinL.yaml"))
    cfg.merge_from_list(["MODEL.SWIN.SIZE", "T", "MODEL.ACTIVE_GRAD_COMPARE", grad_compare, "MODEL.ACTIVE_GRAD_SAVE", grad_compare,
                         "MODEL.ACTIVE_SEED", 3, "MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH",
                         os.path.join(root, "configs", "metadata", "lvis_v1_train_cat_info.json"), "OUTPUT_DIR", ""])
    torch.manual_seed(42)
    model = build_model(cfg).train()
    opt = build_optimizer(cfg, model)
    model.active_selector = BG.ActiveSelector.from_config(cfg, model, opt.arena, model.training_losses)
    base = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device=DEV)
    other = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=7, device=DEV)
    held = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=99, device=DEV)
    batch = []
    for d, o, t in zip(base, other, held):
        e = dict(d)
        e["origin_image"], e["origin_instances"] = o["image"], o["instances"]
        e["test_image"], e["test_instances"], e["test_image_class"] = t["image"], t["instances"], 0
        e["paste_filename_list"] = []
        batch.append(e)
    w0 = opt.arena.p.clone()
    with EventStorage(0):
        opt.zero_grad()
        losses = model(batch)
        sel = model.active_selector
        assert torch.equal(opt.arena.p, w0), "trial updates were not undone exactly"
        assert sel.count == 1 and sel.paste_count + sel.not_paste_count == 1 and sel.iter == 1
        if grad_compare:
            sp, so = float(sel.last["sim_paste_init"]), float(sel.last["sim_ori_init"])
            assert -1.0001 <= sp <= 1.0001 and -1.0001 <= so <= 1.0001 and sel.last["paste"] == (not so > sp)
            assert float(sel.bank.bank.abs().sum()) > 0          # ACTIVE_GRAD_SAVE: the held-out gradient went into the bank
        else:
            old, new = sel.last["old_test_loss"], sel.last["paste_test_loss"]
            assert set(old) == set(new) and all(np.isfinite(float(v)) for v in list(old.values()) + list(new.values()))
            assert sel.last["paste"] == bool(BG.loss_sum(new, "cls") < BG.loss_sum(old, "cls"))
        total = sum(losses.values())
        total.backward()
        assert float(opt.arena.g.abs().sum()) > 0
        opt.step()
    assert np.isfinite(float(total)) and not torch.equal(opt.arena.p, w0)


def test_paste_split_and_no_fed_losses_follow_the_reference_formulas():
    """BS detic_fast_rcnn.py:431-470 (`sigmoid_cross_entropy_loss_with_fed`: weighted BCE summed over pasted / original rows, both
    over B) and :393-430 (no-fed form) restated with plain torch on the same logits and the same class weights."""
    from divergen_amd.modeling import ShapeSpec
    from divergen_amd.modeling.box_regression import Box2BoxTransform
    from divergen_amd.modeling.roi_heads.detic_fast_rcnn import DeticFastRCNNOutputLayers
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(2)
    C, R = 37, 64
    pred = DeticFastRCNNOutputLayers(ShapeSpec(channels=32), box2box_transform=Box2BoxTransform(weights=(10, 10, 5, 5)), num_classes=C,
                                     cls_agnostic_bbox_reg=True, use_sigmoid_ce=True, only_paste_sup=True).to(DEV)
    scores = torch.randn(R, C + 1, generator=g).to(DEV)
    gt = torch.randint(0, C + 1, (R,), generator=g).to(DEV)
    gt[-5:] = -1                                           # padding rows of a fixed-length proposal list
    src = torch.randint(0, 4, (R,), generator=g).to(DEV)
    w = torch.rand(C, generator=g).to(DEV)
    lp, lo = pred.paste_split(scores, gt, w, src)
    v = gt >= 0
    s, t, sr = scores[v], gt[v], src[v]
    target = torch.zeros(len(t), C + 1, device=DEV)
    target[torch.arange(len(t)), t] = 1
    ce = F.binary_cross_entropy_with_logits(s[:, :-1], target[:, :C], reduction="none") * w.view(1, C)
    B = len(t)
    assert torch.allclose(lp, ce[sr >= 1].sum() / B, rtol=1e-5) and torch.allclose(lo, ce[sr == 0].sum() / B, rtol=1e-5)
    assert torch.allclose(lp + lo, ce.sum() / B, rtol=1e-5)          # the two parts make up loss_cls
    nf = pred.sigmoid_cross_entropy_loss(s, t, "none")
    assert torch.allclose(nf, F.binary_cross_entropy_with_logits(s[:, :-1], target[:, :C], reduction="none").sum() / B, rtol=1e-5)


def test_bsgal_r50_forward_once_selection_end_to_end():
    """BS/configs/BSGAL/BSGAL_R50.yaml through the registry (R50-timm FPN, ONLY_PASTE_SUP box heads, ACTIVE_MODE paste_only +
    gradient comparison + forward once + 'only_paste_-0.05', ACTIVE_ONLY_GT_TEST, bank MOMENTUM0.1): the held-out pass runs over
    ground-truth proposals only, the per-paste terms of the pasted batch are differentiated on their own, the decision is the
    reference's `threshold > similarity -> original`, weights and gradient arena come back clean, the returned dict has no
    per-paste keys, and the step trains."""
    from divergen_amd.config import add_bsgal_config, get_cfg
    from divergen_amd.data import synthetic_batch
    from divergen_amd.modeling import build_model
    from divergen_amd.solver import build_optimizer
    from divergen_amd.utils.events import EventStorage
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = add_bsgal_config(get_cfg())
    cfg.merge_from_file(os.path.join(root, "configs", "BSGAL", "BSGAL_R50.yaml"))
    cfg.merge_from_list(["MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", os.path.join(root, "configs", "metadata", "lvis_v1_train_cat_info.json"),
                         "OUTPUT_DIR", ""])
    assert cfg.MODEL.ONLY_PASTE_SUP and cfg.MODEL.ACTIVE_ONLY_GT_TEST and cfg.MODEL.BACKBONE.NAME == "build_p67_timm_fpn_backbone"
    torch.manual_seed(42)
    model = build_model(cfg).train()
    opt = build_optimizer(cfg, model)
    sel = model.active_selector = BG.ActiveSelector.from_config(cfg, model, opt.arena, model.training_losses)
    assert sel.mode == "paste_only" and sel.only_gt_test and sel.bank.update_mode == "MOMENTUM0.1"
    base = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device=DEV)           # instance_source = [0]*8 + [1]*4
    other = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=7, device=DEV)
    held = synthetic_batch(2, 256, cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=99, device=DEV)
    batch = []
    for d, o, t in zip(base, other, held):
        assert int(d["instances"].instance_source.sum()) > 0
        e = dict(d)
        e["origin_image"], e["origin_instances"] = o["image"], o["instances"]
        ti = t["instances"]
        ti.remove("instance_source")
        e["test_image"], e["test_instances"], e["test_image_class"] = t["image"], ti, 0
        e["paste_filename_list"] = ["x.png"]
        batch.append(e)
    # the plain training forward of this configuration carries the per-paste terms ...
    with EventStorage(0):
        plain = model.training_losses([dict(d) for d in base])
        assert {"loss_paste_ins_stage0", "loss_nopaste_ins_stage2"} <= set(plain)
        tot = float(plain["loss_paste_ins_stage0"] + plain["loss_nopaste_ins_stage0"])
        assert abs(tot - float(plain["loss_cls_stage0"])) <= 2e-2 * abs(float(plain["loss_cls_stage0"])) + 1e-3
        # ... and the held-out pass over ground-truth proposals only has exactly one RoI per ground-truth box
        held_l = model.training_losses([{"image": e["test_image"], "instances": e["test_instances"]} for e in batch], only_gt_proposals=True)
        assert all(np.isfinite(float(v)) for v in held_l.values())
    opt.zero_grad()
    w0 = opt.arena.p.clone()
    with EventStorage(0):
        losses = model(batch)
        assert torch.equal(opt.arena.p, w0) and sel.count == 1
        sp, thr = float(sel.last["sim_paste_init"]), float(sel.last["sim_ori_init"])
        assert thr == -0.05 and (np.isnan(sp) or -1.0001 <= sp <= 1.0001) and sel.last["paste"] == (not thr > sp)
        assert float(sel.bank.bank.abs().sum()) > 0
        assert not any("paste" in k for k in losses)
        total = sum(losses.values())
        total.backward()
        opt.step()
    assert np.isfinite(float(total.detach())) and not torch.equal(opt.arena.p, w0)


@pytest.mark.parametrize("optim_mode,use_optimizer", [("sgd", True), ("adam", True), ("adamw", True), ("sgd", False), ("adam", False)])
def test_trial_update_matches_the_reference_optimizers(optim_mode, use_optimizer):
    """update_with_loss (custom_rcnn.py:941-971) for every ACTIVE_OPTIMIZER_MODE the reference constructs (:146-158): SGD(lr),
    Adam(lr, betas=(0, 0)), AdamW(lr) -- whose moments survive the weight restore between trials -- and the manual
    `p -= lr * g` of ACTIVE_OPTIMIZER false, against torch.optim on a CPU copy over three consecutive trial updates."""
    from divergen_amd.solver import FlatArena
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4)).to(DEV)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    arena = FlatArena(net)
    lr = 0.05
    sel = BG.ActiveSelector(net, arena, None, lr=lr, optim_mode=optim_mode, use_optimizer=use_optimizer, loss_update="all")
    mode = optim_mode if use_optimizer else "sgd"
    opt = {"sgd": lambda: torch.optim.SGD(ref.parameters(), lr=lr), "adam": lambda: torch.optim.Adam(ref.parameters(), lr=lr, betas=(0.0, 0.0)),
           "adamw": lambda: torch.optim.AdamW(ref.parameters(), lr=lr)}[mode]()
    for it in range(3):
        x, y = torch.randn(32, 8), torch.randn(32, 4)
        opt.zero_grad()
        ((ref(x) - y) ** 2).mean().backward()
        opt.step()
        sel._update_with_loss({"loss_a": ((net(x.to(DEV)) - y.to(DEV)) ** 2).mean()})
        for p, q in zip(net.parameters(), ref.parameters()):
            torch.testing.assert_close(p.detach().cpu(), q.detach(), atol=2e-6, rtol=2e-5)


def test_active_compare_all_trains_on_both_batches():
    """ACTIVE_COMPARE 'all' (custom_rcnn.py:339, :556-557, :772-774, :1099-1100): no trial passes; the step's losses are the pasted
    batch's plus the original batch's, term by term; counted as 'paste'."""
    from divergen_amd.solver import FlatArena
    from divergen_amd.structures import Instances
    model = _TinyModel().to(DEV).train()
    with torch.no_grad():
        model.backbone.weight.copy_(torch.tensor([[0.5, -1.0, 0.25, 2.0]]))
    arena = FlatArena(model)
    sel = BG.ActiveSelector(model, arena, model.training_losses, mode="paste_or_ori", compare="all", loss="cls")

    def sample(vec, y):
        inst = Instances((1, 1))
        inst.target = torch.tensor([y], device=DEV)
        return inst, torch.tensor(vec, device=DEV)
    pi, px = sample([1.0, 2.0, 0.0, 0.0], 1.0)
    oi, ox = sample([0.0, 0.0, 3.0, 1.0], -2.0)
    ti, tx = sample([1.0, 0.0, 0.0, 0.0], 0.0)
    batch = [{"image": px, "instances": pi, "origin_image": ox, "origin_instances": oi, "test_image": tx, "test_instances": ti}]
    before = arena.p.clone()
    chosen, paste = sel.select(batch)
    assert paste and torch.equal(chosen[0]["image"], px) and torch.equal(arena.p, before)
    extra = sel.extra_losses()
    lp = model.training_losses(chosen)
    total = {k: v + extra[k] for k, v in lp.items()}
    w = torch.tensor([0.5, -1.0, 0.25, 2.0])
    want = float((w @ torch.tensor([1.0, 2.0, 0.0, 0.0]) - 1.0) ** 2 + (w @ torch.tensor([0.0, 0.0, 3.0, 1.0]) + 2.0) ** 2)
    assert abs(float(total["loss_cls_stage0"]) - want) < 1e-5
    assert sel.extra_losses() is None                       # consumed: one original-batch pass per step
    assert (sel.paste_count, sel.not_paste_count, sel.iter) == (1, 0, 1)
