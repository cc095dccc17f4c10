"""oracle/bsgal.py vs the reference's own update_grad_bank / compute_grad_sim outputs (tests/golden/bsgal_bank.npz)."""
import os

import numpy as np
import pytest

from oracle import bsgal as B

G = os.path.join(os.path.dirname(__file__), "golden", "bsgal_bank.npz")


@pytest.mark.parametrize("mode", ["AVERAGE", "MOMENTUM0.9"])
def test_bank_updates_bit_exact(mode):
    z = np.load(G)
    grads = z["%s_grads" % mode]
    bank = np.zeros(grads.shape[1], np.float32)
    for it in range(grads.shape[0]):
        bank = B.update_grad_bank(bank, grads[it], it + 1, mode)             # CPU semantics = how the golden was made
        assert np.array_equal(bank, z["%s_bank_%d" % (mode, it)]), (mode, it)
    # the device semantics (reciprocal multiply) stay within an ulp of the term
    bank_r = np.zeros_like(bank)
    for it in range(grads.shape[0]):
        bank_r = B.update_grad_bank(bank_r, grads[it], it + 1, mode, reciprocal=True)
    assert np.abs(bank_r - bank).max() <= 2.0 ** -22 * np.abs(grads).max()          # one ulp of the largest term
    if "MOMENTUM" in mode:
        assert np.array_equal(bank_r, bank)


@pytest.mark.parametrize("mode", ["AVERAGE", "MOMENTUM0.9"])
def test_similarity(mode):
    z = np.load(G)
    last = z["%s_bank_3" % mode]
    assert abs(B.compute_grad_sim(z["%s_probe" % mode], last, True) - float(z["%s_sim_norm" % mode])) < 1e-6
    raw = float(z["%s_sim_raw" % mode])
    assert abs(B.compute_grad_sim(z["%s_probe" % mode], last, False) - raw) < 1e-5 * max(1.0, abs(raw))
