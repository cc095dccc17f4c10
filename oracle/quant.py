"""Oracle: optional emulation of the product's bf16 STORAGE points.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference runs its backbone under fp16 autocast and its heads in fp32; the
product stores every activation that crosses a kernel boundary as bf16 (BASELINE's precision) and accumulates in fp32.  north_star
asks for losses within 1e-3 of the reference's on identical inputs: an fp32 oracle differs from a bf16-storage pipeline by the
rounding of those stored tensors (measured 2e-5 .. 4e-3 per loss at random init), which is a property of the precision contract,
not of the kernels.  Inside `with bf16_storage():` the oracle functions round to bf16 (round-to-nearest-even, what
v_cvt_pk_bf16_f32 does) exactly where the product stores bf16 -- GEMM / convolution outputs (bias added in fp32 first), LayerNorm
/ GroupNorm outputs, the attention probabilities that feed P V and the attention output, GELU outputs, the residual stream of
Swin stages 1-3 (stage 0 stays fp32), RoIAlign outputs -- and keep fp32 wherever the product accumulates (softmax, reductions,
losses).  What is left between the two is summation ORDER only, and the end-to-end test asserts 1e-3.
Outside the context manager `rb` is the identity: every golden-pinned oracle test runs the plain fp32 restatement.
"""
import contextlib

import torch

_ON = [False]
_RELU = [None]


def relu(x, tag):
    """ReLU of the oracle's heads.  Inside `with relu_masks(m):` the 0/1 pattern of site `tag` is m[tag] -- the pattern the
    implementation under test used at that site, handed over like the proposals and the cascade labels are (a ReLU's on/off
    decision is a DISCRETE intermediate result: two bf16 pipelines whose activations differ at the bf16 noise floor, ~1e-3
    relative after the backbone, disagree on it for ~1e-3 of the elements, and a weight / bias gradient is a signed sum over
    those elements: a fraction f of flipped terms moves it by ~sqrt(2 f) in relative L2, i.e. 3-5 % per ReLU layer).  The value
    is x * mask (it differs from relu(x) only where the two sides disagree on the sign of an |x| ~ 0 element), the gradient is
    gated by the same mask.  Every site must be provided: a missing tag raises."""
    m = _RELU[0]
    if m is None:
        return torch.relu(x)
    if tag not in m:
        raise KeyError("relu_masks: no mask for site '%s' (have: %s)" % (tag, sorted(m)[:8]))
    k = m[tag]
    if tuple(k.shape) != tuple(x.shape):
        raise ValueError("relu_masks: site '%s' has shape %s, the mask %s" % (tag, tuple(x.shape), tuple(k.shape)))
    _RELU_STATS.setdefault(tag, []).append(float(((x > 0) != k.bool()).float().mean()))
    return x * k.to(x.dtype)


_RELU_STATS = {}


@contextlib.contextmanager
def relu_masks(masks):
    """masks: dict site tag -> bool / 0-1 tensor of the site's shape, or None (the oracle's own ReLU)."""
    prev, _RELU[0] = _RELU[0], masks
    _RELU_STATS.clear()
    try:
        yield _RELU_STATS          # per site: the fraction of elements on which the oracle's own sign disagrees with the mask
    finally:
        _RELU[0] = prev


def rb(x):
    """x as it reads back after being stored as bf16 (identity unless bf16_storage() is active)."""
    return x.to(torch.bfloat16).to(x.dtype) if _ON[0] else x


def active():
    return _ON[0]


@contextlib.contextmanager
def bf16_storage(on=True):
    prev, _ON[0] = _ON[0], bool(on)
    try:
        yield
    finally:
        _ON[0] = prev
