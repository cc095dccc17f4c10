"""CPU tests of BSGAL's outer loop (divergen_amd/engine/bsgal.py: the decision part of BS/bsgal/modeling/meta_arch/
custom_rcnn.py:278-780): helper semantics restated from the reference (:29-48 DynamicThreshold, :1088-1209 fetchloss /
compare_loss / pop_loss_paste, :317-327 reset_instance_source) and the loss-comparison selection on a small model whose
answer is known in closed form.  The gradient-comparison branch runs libdgx kernels: tests/test_gpu_bsgal.py."""
import numpy as np
import pytest
import torch

from divergen_amd.engine import bsgal as BG
from divergen_amd.solver import FlatArena
from divergen_amd.structures import Instances


def test_loss_helpers_follow_the_reference_rules():
    losses = {"loss_cls_stage0": torch.tensor(1.0), "loss_box_reg_stage0": torch.tensor(2.0), "loss_cls_stage1": torch.tensor(4.0),
              "loss_mask": torch.tensor(8.0), "loss_cls_paste_stage0": torch.tensor(16.0), "loss_paste_ins": torch.tensor(32.0)}
    assert set(BG.fetchloss(losses, ["cls_stage0"])) == {"loss_cls_stage0"}
    assert set(BG.fetchloss(losses, ["_paste_"])) == {"loss_cls_paste_stage0", "loss_paste_ins"}
    rest, paste = BG.pop_loss_paste(losses)
    assert set(paste) == {"loss_cls_paste_stage0", "loss_paste_ins"} and set(rest) == set(losses) - set(paste)
    assert float(BG.loss_sum(rest, "cls")) == 5.0 and float(BG.loss_sum(rest, "all")) == 15.0
    assert float(BG.loss_sum(rest, "stage0")) == 3.0 and float(BG.loss_sum(rest, "mask")) == 8.0
    with pytest.raises(NotImplementedError):
        BG.loss_sum(rest, "nonsense")
    lo, hi = {"loss_cls": torch.tensor(1.0)}, {"loss_cls": torch.tensor(2.0)}
    # '<' keeps the old (original) batch, '>' takes the new (pasted) one; 'default' takes the new one when ITS loss is lower
    assert BG.compare_loss(hi, lo, "default") == ">" and BG.compare_loss(lo, hi, "default") == "<"
    assert BG.compare_loss(hi, lo, "contra") == "<" and BG.compare_loss(lo, hi, "contra") == ">"
    assert BG.compare_loss(lo, hi, "all") == ">"
    assert BG.compare_loss(hi, lo, "random_0.8", rand=lambda: 0.9) == "<" and BG.compare_loss(hi, lo, "random_0.8", rand=lambda: 0.1) == ">"
    assert BG.compare_loss(hi, lo, "random", rand=lambda: 0.6) == "<"
    assert BG.compare_loss(hi, lo, "prob", rand=lambda: 0.5) == ">" and BG.compare_loss(hi, lo, "prob", rand=lambda: 0.9) == "<"
    assert BG.compare_loss(lo, hi, "schedule", it=45000, rand=lambda: 0.9) == "<"      # 0.9 > 0.5: compare -> old is lower
    assert BG.compare_loss(lo, hi, "schedule", it=45000, rand=lambda: 0.1) == ">"      # past the schedule: always paste
    with pytest.raises(NotImplementedError):
        BG.compare_loss(lo, hi, "nonsense")


def test_dynamic_threshold_is_a_running_percentile():
    q = BG.DynamicThreshold(buffer_size=5, percentile=0.8)
    assert q.get_threshold() == 0
    xs = [0.3, -0.1, 0.7, 0.2, 0.9, 0.05, 0.4]
    for x in xs:
        q.add_score(x)
    assert q.get_threshold() == pytest.approx(np.percentile(np.array(xs[-5:]), 80.0))
    q.set_percentile(0.25)
    assert q.get_threshold() == pytest.approx(np.percentile(np.array(xs[-5:]), 25.0))


def test_reset_instance_source_numbers_pastes_over_the_batch():
    a, b, c = Instances((4, 4)), Instances((4, 4)), Instances((4, 4))
    a.instance_source = torch.tensor([0, 0, 1, 1])
    b.instance_source = torch.tensor([0, 0, 0])
    c.instance_source = torch.tensor([0, 1, 1, 1])
    out = BG.reset_instance_source([a, b, c])
    assert [o.instance_source.tolist() for o in out] == [[0, 0, 1, 2], [0, 0, 0], [0, 3, 4, 5]]
    assert a.instance_source.tolist() == [0, 0, 1, 1]          # copies: the training batch keeps its 0/1 flags


class _Tiny(torch.nn.Module):
    """y = w . x with a `backbone` submodule (the selector switches it to eval mode for the trial passes)."""

    def __init__(self):
        super().__init__()
        self.backbone = torch.nn.Linear(2, 1, bias=False)
        with torch.no_grad():
            self.backbone.weight.copy_(torch.tensor([[0.0, 0.0]]))
        self.modes = []

    def training_losses(self, batch):
        self.modes.append(self.backbone.training)
        x = torch.stack([d["image"] for d in batch])
        y = torch.stack([d["instances"].target[0] for d in batch])
        return {"loss_cls_stage0": ((self.backbone(x).squeeze(1) - y) ** 2).mean(), "loss_cls_paste_stage0": torch.zeros(())}


def _sample(x, y):
    inst = Instances((1, 1))
    inst.target = torch.tensor([y])
    return inst, torch.tensor(x)


@pytest.mark.parametrize("mode", ["paste_or_ori", "paste_or_zero"])
def test_loss_comparison_picks_the_batch_whose_trial_step_helps_the_held_out_batch(mode, tmp_path):
    """Held-out batch: x = (1, 0), y = 1 -> its loss falls only when w[0] grows.  The 'pasted' batch teaches w[0] (x = (1, 0),
    y = 1), the original one teaches w[1] (x = (0, 1), y = 1): one SGD step on the pasted batch lowers the held-out loss, one
    on the original leaves it -> paste.  With the roles swapped -> original.  Weights come back bit-identical, gradients
    zeroed, trial passes ran with the backbone in eval mode, and the log line is written."""
    model = _Tiny().train()
    arena = FlatArena(model)
    w0 = arena.p.clone()
    sel = BG.ActiveSelector(model, arena, model.training_losses, mode=mode, compare="default", loss="cls", lr=0.1,
                            output_dir=str(tmp_path))
    good, bad, held = ([1.0, 0.0], 1.0), ([0.0, 1.0], 1.0), ([1.0, 0.0], 1.0)

    def batch(pasted, original):
        pi, px = _sample(*pasted)
        oi, ox = _sample(*original)
        ti, tx = _sample(*held)
        pi.instance_source = torch.tensor([1])
        return [{"image": px, "instances": pi, "origin_image": ox, "origin_instances": oi, "test_image": tx, "test_instances": ti,
                 "test_image_class": 7, "paste_filename_list": ["a.png", "b.png"]}]

    chosen, paste = sel.select(batch(good, bad))
    assert paste and torch.equal(chosen[0]["image"], torch.tensor(good[0]))
    assert torch.equal(arena.p, w0) and float(arena.g.abs().sum()) == 0.0
    assert model.modes and not any(model.modes) and model.backbone.training      # eval during the trials, train mode restored
    chosen, paste = sel.select(batch(bad, good))
    if mode == "paste_or_ori":
        assert not paste and torch.equal(chosen[0]["image"], torch.tensor(good[0]))
    else:       # paste_or_zero compares with the untouched weights: the useless pasted batch does not LOWER the loss -> not pasted
        assert not paste
    assert torch.equal(arena.p, w0) and float(arena.g.abs().sum()) == 0.0
    assert (sel.count, sel.paste_count, sel.not_paste_count, sel.iter) == (2, 1, 1, 2)
    lines = open(tmp_path / "paste_source" / "rank_0" / "10000.txt").read().splitlines()
    assert len(lines) == 4 and lines[0].startswith("a.png select_class: 7 paste: 1 iter: 0 loss_dif: ")
    assert " paste: 0 iter: 1 " in lines[2]


def test_unbuilt_modes_say_so():
    model = _Tiny()
    arena = FlatArena(model)
    with pytest.raises(NotImplementedError):
        BG.ActiveSelector(model, arena, model.training_losses, mode="paste_only")
    with pytest.raises(NotImplementedError):
        BG.ActiveSelector(model, arena, model.training_losses, optim_mode="rmsprop")      # :150-158 knows sgd / adam / adamw
    assert BG.ActiveSelector(model, arena, model.training_losses, optim_mode="adam").optim_mode == "adam"
    assert BG.ActiveSelector(model, arena, model.training_losses, optim_mode="adam", use_optimizer=False).optim_mode == "sgd"
    assert BG.ActiveSelector(model, arena, model.training_losses, compare="all").compare == "all"


def test_bsgal_configs_load_and_mapper_adds_the_selection_inputs(tmp_path, monkeypatch):
    """BS/configs/BSGAL/BSGAL_SwinL.yaml through add_bsgal_config; CopyPasteMapper with INPUT.ACTIVE_SELECT keeps the un-pasted
    sample and draws a held-out image that shows one of the pasted classes (mapper.py:260-284, :958-963, :1038-1057)."""
    import os
    from divergen_amd.config import add_bsgal_config, get_cfg
    from divergen_amd.data import build as B
    from test_host_data import _tiny_lvis
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = add_bsgal_config(get_cfg())
    cfg.merge_from_file(os.path.join(root, "configs", "BSGAL", "BSGAL_SwinL.yaml"))
    assert cfg.MODEL.ACTIVE_MODE == "paste_or_ori" and cfg.INPUT.ACTIVE_SELECT and not cfg.MODEL.ACTIVE_GRAD_COMPARE
    cfg.merge_from_file(os.path.join(root, "configs", "BSGAL", "BSGAL_R50.yaml"))
    assert cfg.MODEL.ACTIVE_GRAD_UPDATE == "MOMENTUM0.1" and cfg.MODEL.ACTIVE_ONCE_MODE == "only_paste_-0.05"
    monkeypatch.setenv("DETECTRON2_DATASETS", _tiny_lvis(tmp_path))
    cfg = add_bsgal_config(get_cfg())
    cfg.merge_from_file(os.path.join(root, "configs", "DiverGen_swinL.yaml"))
    cfg.merge_from_list(["INPUT.ACTIVE_SELECT", True, "INPUT.INST_POOL", False, "INPUT.USE_COPY_METHOD", "none", "INPUT.TRAIN_SIZE", 64])
    dicts = B.get_detection_dataset_dicts(["lvis_v1_train"])
    mapper = B.CopyPasteMapper(B.DatasetMapper(cfg, True), cfg)
    mapper.set_dataset(dicts)
    assert set(mapper.per_cat_pool_real) == {0, 1, 2} and all(mapper.per_cat_pool_real[c] for c in (0, 1, 2))

    class _Pool:                                     # stands in for InstPool (its compositor needs the GPU): "pastes" class 2
        def prepare(self, data):
            out = dict(data)
            out["paste_labels"], out["paste_filename_list"] = [2], ["p.png"]
            return out
    mapper.inst_pool = _Pool()
    np.random.seed(0)
    out = mapper.finish(mapper(dicts[0]), "cpu")          # worker half, then the training process's half
    assert out["test_image_class"] == 2 and (out["test_instances"].gt_classes == 2).any()
    assert torch.equal(out["origin_image"], out["image"]) and len(out["origin_instances"]) == len(out["instances"])
    assert out["origin_instances"].instance_source.tolist() == [0] * len(out["instances"])
