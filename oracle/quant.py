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
