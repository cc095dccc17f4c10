"""CPU restatement of the BSGAL gradient-bank arithmetic.  TEST INFRASTRUCTURE ONLY.

* update_grad_bank  <- BS/bsgal/modeling/meta_arch/custom_rcnn.py:1046-1062
* compute_grad_sim  <- :1074-1086
Pinned: tests/golden/bsgal_bank.npz holds the outputs of the reference's own methods (called unbound on a stand-in `self`
by tests/golden/make_golden.py) -- tests/test_oracle_bsgal.py.

`grad / (iter + 1)` is a true division on the CPU but `grad * (1 / (iter + 1))` in ATen's GPU kernel when the divisor is a
host scalar (div_true_kernel_cuda's reciprocal path), which is what the reference executes on its training device; both
are restated (`reciprocal=`).  They differ by at most one ulp of that term."""
import numpy as np

f32 = np.float32


def update_grad_bank(bank, grad, it, mode="AVERAGE", reciprocal=False):
    """One update; `it` is the reference's self.iter at the call.  Returns the new bank (fp32)."""
    bank = np.asarray(bank, f32)
    grad = np.asarray(grad, f32)
    if mode == "AVERAGE":
        out = bank * f32(it / (it + 1))
        term = grad * (f32(1.0) / f32(it + 1)) if reciprocal else grad / f32(it + 1)
        return (out + term).astype(f32)
    if "MOMENTUM" in mode:
        m = float(mode.split("TUM")[1])
        out = bank * f32(m)
        return (out + grad * f32(1 - m)).astype(f32)
    raise NotImplementedError(mode)


def bank_coefficients(it, mode):
    """(a, b) of bank*a + grad*b as the fp32 scalars the device kernel is handed."""
    if mode == "AVERAGE":
        return f32(it / (it + 1)), f32(1.0) / f32(it + 1)
    m = float(mode.split("TUM")[1])
    return f32(m), f32(1 - m)


def compute_grad_sim(g1, g2, norm=True):
    """fp64 accumulation (the reference reduces in fp32 with an unspecified order; compare with a tolerance)."""
    a, b = np.asarray(g1, np.float64), np.asarray(g2, np.float64)
    dot = float((a * b).sum())
    if not norm:
        return dot
    return dot / (float(f32(np.sqrt((a * a).sum()))) * float(f32(np.sqrt((b * b).sum()))) + 1e-8)
