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
