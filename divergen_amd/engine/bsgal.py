"""BSGAL gain scoring on the flat arenas (BS/bsgal/modeling/meta_arch/custom_rcnn.py).

The reference scores "does this augmented batch help" by comparing flattened gradients: `get_loss_grad` (:973-1001)
backward + torch.cat of every parameter gradient, `update_grad_bank` (:1046-1072) keeps a running mean / momentum of the
held-out batch's gradient in an nn.Embedding(grad_size, 1), `compute_grad_sim` (:1074-1086) is a dot product or cosine, and
weights are saved / restored around trial updates with state_dict copies (:395-400 `load_state_dict(old_weights)`).
With every gradient already a view into one arena (solver.FlatArena) nothing needs flattening: a gradient snapshot is one
device copy, bank update and similarity are one streaming kernel each (dgx_grad_bank_update, dgx_grad_sim), and weights are
saved / restored with one copy of the parameter arena.  Nothing here synchronises with the host; `paste_is_better` returns a
device boolean."""
import torch

from .. import _lib as L


def bank_coefficients(it, mode):
    """(a, b) of bank = bank*a + grad*b as fp32 scalars: AVERAGE -> it/(it+1), 1/(it+1);  MOMENTUMm -> m, 1-m."""
    f = torch.tensor
    if mode == "AVERAGE":
        return float(f(it / (it + 1), dtype=torch.float32)), float(f(1.0, dtype=torch.float32) / f(it + 1, dtype=torch.float32))
    if "MOMENTUM" in mode:
        m = float(mode.split("TUM")[1])
        return float(f(m, dtype=torch.float32)), float(f(1 - m, dtype=torch.float32))
    raise NotImplementedError(mode)


def grad_sim(g1, g2):
    """One pass over both vectors -> (f64[3] = dot, |g1|^2, |g2|^2 ; f32[4] = dot, |g1|, |g2|, cosine), device tensors."""
    assert g1.numel() == g2.numel() and g1.dtype == g2.dtype == torch.float32
    n = g1.numel()
    lib = L.lib()
    ws = torch.empty(max(int(lib.dgx_grad_sim_workspace_bytes(n)), 8), dtype=torch.uint8, device=g1.device)
    o3 = torch.empty(3, dtype=torch.float64, device=g1.device)
    o4 = torch.empty(4, dtype=torch.float32, device=g1.device)
    L.check(lib.dgx_grad_sim(L.ptr(g1), L.ptr(g2), n, L.ptr(o3), L.ptr(o4), L.ptr(ws), L.stream()), "dgx_grad_sim")
    return o3, o4


class GradBank:
    """init_grad_bank / update_grad_bank / compute_grad_sim (:1031-1086) over a FlatArena."""

    def __init__(self, arena, update="AVERAGE", norm=True):
        self.arena, self.update_mode, self.norm = arena, update, norm
        self.bank = torch.zeros_like(arena.g)          # init_grad_bank: zeros of grad_size
        bank_coefficients(1, update)                   # validates the mode string

    def loss_grad(self, losses, retain=False):
        """get_loss_grad: zero the gradients, backward the summed losses, return the flattened gradient (a snapshot;
        parameters without a gradient contribute zeros, as in the reference)."""
        self.arena.zero_grad()
        total = sum(losses.values()) if isinstance(losses, dict) else losses
        total.backward(retain_graph=retain)
        return self.arena.g.clone()

    def update(self, grad, it):
        """update_grad_bank with the reference's self.iter = it; returns the bank (not a copy, like `.weight.data`)."""
        a, b = bank_coefficients(it, self.update_mode)
        L.check(L.lib().dgx_grad_bank_update(L.ptr(self.bank), L.ptr(grad), self.bank.numel(), a, b, L.stream()),
                "dgx_grad_bank_update")
        return self.bank

    def similarity(self, g1, g2=None, norm=None):
        """compute_grad_sim: cosine (norm) or plain dot product; 0-dim fp32 device tensor."""
        _, o4 = grad_sim(g1, self.bank if g2 is None else g2)
        return o4[3] if (self.norm if norm is None else norm) else o4[0]

    def paste_is_better(self, paste_grad, ori_grad, ref_grad=None):
        """The gradient-compare decision of `paste_or_ori` (:447-454 + the `>` comparison that follows): the augmented
        batch is used when its gradient agrees with the held-out gradient at least as well as the original batch's."""
        return self.similarity(paste_grad, ref_grad) >= self.similarity(ori_grad, ref_grad)


class WeightSnapshot:
    """`old_weights = copy.deepcopy(self.state_dict())` ... `self.load_state_dict(old_weights)` (:330-400) as one arena copy."""

    def __init__(self, arena):
        self.arena = arena
        self.saved = arena.p.clone()

    def restore(self):
        self.arena.p.copy_(self.saved)
        self.arena.sync_shadow()


# ------------------------------------------------------------------------------------------------------------------
# The outer loop: "is the pasted batch better than the original one?" decided inside the training forward
# (BS/bsgal/modeling/meta_arch/custom_rcnn.py:278-780), for the modes the shipped Swin-L configuration and the gradient
# variant of the paper use.  Everything below is host logic over the model's own forward; the heavy parts are the model's
# HIP path, the arena copies / axpy of the trial update and the two kernels above.
class DynamicThreshold:
    """custom_rcnn.py:29-48: percentile of the last `buffer_size` scores."""

    def __init__(self, buffer_size=100, percentile=0.85):
        from collections import deque
        self.queue = deque(maxlen=buffer_size)
        self.percentile = percentile * 100

    def add_score(self, score):
        self.queue.append(score)

    def set_percentile(self, percentile):
        self.percentile = percentile * 100

    def get_threshold(self):
        import numpy as np
        return 0 if len(self.queue) == 0 else np.percentile(np.array(self.queue), self.percentile)


def fetchloss(losses, str_list):
    """:1088-1095 -- the entries whose key contains one of the strings."""
    return {k: v for k, v in losses.items() if any(s in k for s in str_list)}


def pop_loss_paste(losses):
    """:1202-1209 -- split off the entries with 'paste' in their key (per-paste supervision terms)."""
    paste = {k: v for k, v in losses.items() if "paste" in k}
    return {k: v for k, v in losses.items() if k not in paste}, paste


_LOSS_MODES = {"cls": "cls", "box": "box", "mask": "mask", "cls_stage0": "cls_stage0", "stage0": "stage0"}


def loss_sum(losses, mode):
    """The sums compare_loss / compute_diff_loss take (:1113-1131, :1173-1189)."""
    if mode == "all":
        return sum(losses.values())
    if mode not in _LOSS_MODES:
        raise NotImplementedError(mode)
    return sum(v for k, v in losses.items() if _LOSS_MODES[mode] in k)


def compare_loss(old_loss, new_loss, active_compare="default", active_loss="cls", it=0, rand=None):
    """compare_loss (:1097-1169): '<' = keep the old (original) batch, '>' = take the new (pasted) one.  `rand` = a
    callable returning a uniform [0,1) draw (random.random in the reference)."""
    import random
    rand = rand or random.random
    if active_compare == "all":
        return ">"
    if "random" in active_compare:
        thres = 0.5 if active_compare == "random" else float(active_compare.split("_")[1])
        return "<" if rand() > thres else ">"
    old_s, new_s = loss_sum(old_loss, active_loss), loss_sum(new_loss, active_loss)
    lower = bool(new_s < old_s)
    if active_compare == "contra":
        return "<" if lower else ">"
    if active_compare == "prob":
        if rand() < 0.8:
            return ">" if lower else "<"
        return "<" if lower else ">"
    if active_compare == "default":
        return ">" if lower else "<"
    if active_compare == "schedule":
        if rand() > it / 90000:
            return ">" if lower else "<"
        return ">"
    raise NotImplementedError(active_compare)


def reset_instance_source(gt_instances):
    """:317-327 -- [0,0,0,1,1,1] per image -> [0,0,0,1,2,3] with ids running over the whole batch (copies)."""
    import copy
    total = 1
    out = copy.deepcopy(gt_instances)
    for inst in out:
        n = int(inst.instance_source.sum())
        if n > 0:
            assert int(inst.instance_source[-n:].sum()) == n
            inst.instance_source[-n:] = torch.arange(total, total + n, device=inst.instance_source.device)
        total += n
    return out


class ActiveSelector:
    """The decision part of BSGAL's CustomRCNN.forward for ACTIVE_MODE 'paste_or_ori' / 'paste_or_zero'.

    Loss comparison (ACTIVE_GRAD_COMPARE false; BS/configs/BSGAL/BSGAL_SwinL.yaml, :330-470, :560-600): from a snapshot of the
    weights, take one trial SGD step (lr ACTIVE_LR) on the pasted batch, measure the loss of a held-out 'test' batch, restore;
    the same for the original batch; train on the batch whose trial step left the lower test loss.
    Gradient comparison (ACTIVE_GRAD_COMPARE true, :345-355, :447-460): gradient of the held-out batch's ACTIVE_LOSS terms
    (optionally averaged into the bank), gradients of the pasted and of the original batch's training losses, cosine (or dot)
    of each with the held-out gradient; the batch that agrees better wins.
    `loss_fn(batched_inputs) -> loss dict` is the model's plain training forward.  Trial passes run with the backbone in
    eval mode (no stochastic depth), as `no_grad_loss` (:780-939) does, and never signal the data-parallel reducer (each rank
    decides for its own batch, like the reference, which calls the module below its DDP wrapper)."""

    def __init__(self, model, arena, loss_fn, *, mode="paste_or_ori", compare="default", loss="cls", loss_update="all", lr=1e-4,
                 use_optimizer=True, optim_mode="sgd", grad_compare=False, grad_norm=True, grad_save=False, grad_update="AVERAGE",
                 seed=0, test_batchsize=4, output_dir=None, rank=0, forward_once=False, once_mode="only_gt", only_gt_test=False,
                 max_iter=90000):
        once = mode == "paste_only" and forward_once and grad_compare and once_mode.startswith("only_paste")
        if mode not in ("paste_or_ori", "paste_or_zero") and not once:
            raise NotImplementedError("ACTIVE_MODE '%s' (forward_once %s, once mode '%s'): built are 'paste_or_ori', 'paste_or_zero' "
                                      "and 'paste_only' with ACTIVE_GRAD_COMPARE + ACTIVE_FORWARD_ONCE + ACTIVE_ONCE_MODE "
                                      "'only_paste_*' (BSGAL_R50.yaml)" % (mode, forward_once, once_mode))
        self.once_mode, self.only_gt_test, self.max_iter = once_mode, only_gt_test, max_iter
        self.dynamic_queue = None
        if once and "dynamic" in once_mode:                    # :127-136
            if "linear" not in once_mode:
                self.dynamic_queue = DynamicThreshold(buffer_size=1000, percentile=1 - float(once_mode.split("_")[-1]))
            else:                                              # "only_paste_dynamic_linear_0.3_0.5"
                self.start_rate, self.end_rate = float(once_mode.split("_")[-2]), float(once_mode.split("_")[-1])
                self.dynamic_queue = DynamicThreshold(buffer_size=1000, percentile=1 - self.start_rate)
        # the trial update (:146-158, :941-971): ACTIVE_OPTIMIZER false = the manual `p -= lr * g`, which is what SGD(lr) does too;
        # 'adam' = torch.optim.Adam(lr, betas=(0, 0)) -- stateless, p -= lr * g / (|g| + 1e-8); 'adamw' = torch.optim.AdamW(lr) whose
        # moments persist across trials (the reference never restores them with the weights)
        self.optim_mode = optim_mode if use_optimizer else "sgd"
        if self.optim_mode not in ("sgd", "adam", "adamw"):
            raise NotImplementedError("ACTIVE_OPTIMIZER_MODE '%s' (:150-158 knows 'sgd', 'adam', 'adamw')" % optim_mode)
        self._m = self._v = None
        self._trial_steps = 0
        if grad_compare and mode == "paste_or_zero":
            raise NotImplementedError("gradient comparison is defined for 'paste_or_ori' / 'paste_only'")
        self.model, self.arena, self.loss_fn = model, arena, loss_fn
        self.mode, self.compare, self.loss, self.loss_update, self.lr = mode, compare, loss, loss_update, lr
        self.grad_compare, self.grad_save, self.seed, self.test_batchsize = grad_compare, grad_save, seed, test_batchsize
        self.bank = GradBank(arena, update=grad_update, norm=grad_norm) if grad_compare else None
        self.iter = self.count = self.paste_count = self.not_paste_count = 0
        self.output_dir, self.rank = output_dir, str(rank)
        self.last = {}

    @classmethod
    def from_config(cls, cfg, model, arena, loss_fn, rank=0):
        m = cfg.MODEL
        return cls(model, arena, loss_fn, mode=m.ACTIVE_MODE, compare=m.ACTIVE_COMPARE, loss=m.ACTIVE_LOSS,
                   loss_update=m.ACTIVE_LOSS_UPDATE, lr=m.ACTIVE_LR, use_optimizer=m.ACTIVE_OPTIMIZER,
                   optim_mode=m.ACTIVE_OPTIMIZER_MODE, grad_compare=m.ACTIVE_GRAD_COMPARE, grad_norm=m.ACTIVE_GRAD_NORM,
                   grad_save=m.ACTIVE_GRAD_SAVE, grad_update=m.ACTIVE_GRAD_UPDATE, seed=m.ACTIVE_SEED,
                   test_batchsize=m.ACTIVE_TEST_BATCHSIZE, output_dir=cfg.OUTPUT_DIR, rank=rank,
                   forward_once=m.ACTIVE_FORWARD_ONCE, once_mode=m.ACTIVE_ONCE_MODE, only_gt_test=m.ACTIVE_ONLY_GT_TEST,
                   max_iter=cfg.SOLVER.MAX_ITER)

    # ---- pieces
    def _trial_losses(self, inputs, no_grad, for_test=False):
        """no_grad_loss (:780-939): backbone in eval mode for the pass, the rest of the model as in training; the held-out pass
        (for_test) runs over ground-truth proposals only when ACTIVE_ONLY_GT_TEST is set (:898-905)."""
        bb = self.model.backbone
        was = bb.training
        bb.eval()
        # :816-829 every no-grad pass, :892-905 the held-out pass with gradients unless an image has no ground truth at all
        only_gt = self.only_gt_test and (no_grad or (for_test and all(len(d["instances"]) > 0 for d in inputs)))
        kw = {"only_gt_proposals": True} if only_gt else {}
        try:
            if no_grad:
                with torch.no_grad():
                    return self.loss_fn(inputs, **kw)
            return self.loss_fn(inputs, **kw)
        finally:
            bb.train(was)

    def _once_threshold(self, sim_paste):
        """:526-541 -- the stand-in for the original batch's similarity in the forward-once modes: a constant taken from the
        mode string ('only_paste_-0.05'), or a running percentile of the similarities seen so far."""
        if self.dynamic_queue is None:
            return float(self.once_mode.split("_")[-1])
        if "linear" in self.once_mode:
            self.dynamic_queue.set_percentile(1 - (self.start_rate + (self.end_rate - self.start_rate) * self.iter / self.max_iter))
        thr = self.dynamic_queue.get_threshold()
        self.dynamic_queue.add_score(float(sim_paste))
        return thr

    def _reseed(self):
        if self.seed != 0:
            torch.manual_seed(self.seed + self.iter)

    def _update_with_loss(self, losses):
        """update_with_loss (:941-961) with torch.optim.SGD(lr): p -= lr * grad over the whole arena, shadow refreshed."""
        total = loss_sum(losses, self.loss_update) if self.loss_update in ("all", "cls") else None
        if total is None:
            raise NotImplementedError(self.loss_update)
        self.arena.zero_grad()
        total.backward()
        if self.optim_mode == "sgd":
            self.arena.p.add_(self.arena.g, alpha=-self.lr)
            self.arena.sync_shadow()
            return
        from ..layers.optim_ops import adamw_ema_step
        if self._m is None:
            self._m, self._v = torch.zeros_like(self.arena.p), torch.zeros_like(self.arena.p)
        p16 = self.arena.p16 if self.arena.p16.is_cuda else None
        if self.optim_mode == "adam":      # betas (0, 0): exp_avg = g, exp_avg_sq = g^2, both bias corrections 1 at every step
            adamw_ema_step(self.arena.p, self.arena.g, self._m, self._v, None, 1, self.lr, (0.0, 0.0), 1e-8, 0.0, 0.0, 1.0, 0.0, p_bf16=p16)
        else:                              # torch.optim.AdamW defaults: betas (0.9, 0.999), eps 1e-8, weight decay 0.01
            self._trial_steps += 1
            adamw_ema_step(self.arena.p, self.arena.g, self._m, self._v, None, self._trial_steps, self.lr, (0.9, 0.999), 1e-8, 0.01, 0.0,
                           1.0, 0.0, p_bf16=p16)
        if p16 is None:
            self.arena.sync_shadow()
        else:
            self.arena.refresh_transposes()

    def _split(self, batched_inputs):
        keep = ("height", "width", "file_name", "image_id")
        paste = [dict(x) for x in batched_inputs]
        gt = reset_instance_source([x["instances"] for x in batched_inputs]) \
            if all(x["instances"].has("instance_source") for x in batched_inputs) else [x["instances"] for x in batched_inputs]
        for d, g in zip(paste, gt):
            d["instances"] = g
        ori = [{**{k: x[k] for k in keep if k in x}, "image": x["origin_image"], "instances": x["origin_instances"]}
               for x in batched_inputs]
        test = [{"image": x["test_image"], "instances": x["test_instances"]} for x in batched_inputs]
        if self.test_batchsize > len(test) and all("test_image2" in x for x in batched_inputs):
            test += [{"image": x["test_image2"], "instances": x["test_instances2"]} for x in batched_inputs]
        return paste, ori, test[:self.test_batchsize] if self.test_batchsize < len(test) else test

    # ---- the decision
    def select(self, batched_inputs):
        """-> (the batch to train on, paste?).  Leaves weights, shadow and the gradient arena exactly as it found them
        (weights bit-identical; gradients zeroed, as `self.zero_grad()` does at :399,:468)."""
        from ..layers.linear_ops import suspend_ready
        paste_in, ori_in, test_in = self._split(batched_inputs)
        info = {}
        if self.compare == "all":
            # :339, :556-557, :772-774 -- no trial passes: the step trains on the pasted batch AND on the original one (whose losses
            # custom_rcnn.forward adds term by term: `extra_losses`); recorded as "paste"
            self.count += 1
            self.paste_count += 1
            self.last = {"paste": True, "extra": ori_in}
            self.iter += 1
            return paste_in, True
        with suspend_ready():
            if self.mode == "paste_only":
                # forward once (:345-355, :471-541, :592-603): held-out gradient (into the bank), then ONE pass over the pasted
                # batch whose per-paste classification terms (`loss_paste_ins_stage*`) are differentiated on their own
                tl = self._trial_losses(test_in, False, for_test=True)
                ref = self.bank.loss_grad(fetchloss(tl, [self.loss]) if self.loss != "all" else tl)
                if self.grad_save:
                    ref = self.bank.update(ref, self.iter)
                self._reseed()
                only_paste = fetchloss(self._trial_losses(paste_in, False), ["_paste_"])
                if not only_paste:
                    raise RuntimeError("ACTIVE_ONCE_MODE '%s' needs the per-paste loss terms (MODEL.ONLY_PASTE_SUP) and "
                                       "instance_source on the pasted batch" % self.once_mode)
                sim_paste = self.bank.similarity(ref, self.bank.loss_grad(only_paste))
                sim_ori = self._once_threshold(sim_paste)
                info = {"sim_paste_init": sim_paste, "sim_ori_init": sim_ori, "loss_dif": sim_paste - sim_ori}
                decision = "<" if bool(sim_ori > sim_paste) else ">"
            elif self.grad_compare:
                tl = self._trial_losses(test_in, False, for_test=True)
                ref = self.bank.loss_grad(fetchloss(tl, [self.loss]) if self.loss != "all" else tl)
                self._reseed()
                g_paste = self.bank.loss_grad(pop_loss_paste(self._trial_losses(paste_in, False))[0])
                g_ori = self.bank.loss_grad(self._trial_losses(ori_in, False))
                if self.grad_save:
                    ref = self.bank.update(ref, self.iter)
                sim_paste, sim_ori = self.bank.similarity(g_paste, ref), self.bank.similarity(g_ori, ref)
                info = {"sim_paste_init": sim_paste, "sim_ori_init": sim_ori, "loss_dif": sim_paste - sim_ori}
                decision = "<" if bool(sim_ori > sim_paste) else ">"          # :592-603
            else:
                snap = WeightSnapshot(self.arena)
                init_test = self._trial_losses(test_in, True, for_test=True) if self.mode == "paste_or_zero" else None
                self._reseed()
                paste_train = pop_loss_paste(self._trial_losses(paste_in, False))[0]
                self._update_with_loss(paste_train)
                self._reseed()
                paste_test = self._trial_losses(test_in, True)
                snap.restore()
                if self.mode == "paste_or_zero":
                    old = init_test
                else:
                    ori_train = self._trial_losses(ori_in, False)
                    self._update_with_loss(ori_train)
                    self._reseed()
                    old = self._trial_losses(test_in, True)
                    snap.restore()
                decision = compare_loss(old, paste_test, self.compare, self.loss, self.iter)
                info = {"old_test_loss": old, "paste_test_loss": paste_test,
                        "loss_dif": loss_sum(old, self.loss) - loss_sum(paste_test, self.loss)}
            self.arena.zero_grad()
        paste = decision != "<"
        self.count += 1
        self.paste_count += int(paste)
        self.not_paste_count += int(not paste)
        info["paste"] = paste
        self.last = info
        self._log(batched_inputs, paste, info)
        self.iter += 1
        return (paste_in if paste else ori_in), paste

    def extra_losses(self):
        """ACTIVE_COMPARE 'all': the training losses of the original batch (:556-557 `no_grad_loss(ori..., no_grad=False)`: backbone in
        eval mode, gradients on), which the caller adds to the pasted batch's losses (:772-774); None otherwise."""
        extra = self.last.pop("extra", None) if self.compare == "all" else None
        return None if extra is None else self._trial_losses(extra, False)

    def _log(self, batched_inputs, paste, info):
        """The per-iteration record under OUTPUT_DIR/paste_source/rank_R/ (:606-641), one line per pasted file."""
        if not self.output_dir:
            return
        import os
        path = os.path.join(self.output_dir, "paste_source", "rank_" + self.rank, str(self.iter // 10000 + 1) + "0000.txt")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        lists = [x.get("paste_filename_list", []) for x in batched_inputs]
        n = sum(len(v) for v in lists)
        dif = round(float(info["loss_dif"]), 4)
        with open(path, "a") as f:
            for i, names in enumerate(lists):
                for name in names:
                    f.write("%s select_class: %s paste: %d iter: %d loss_dif: %s paste_num: %d\n" % (
                        name, batched_inputs[i].get("test_image_class"), int(paste), self.iter, dif, n))
