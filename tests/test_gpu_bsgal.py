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
