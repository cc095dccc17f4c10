"""Fused clip + AdamW + EMA step over flat fp32 arenas.
Reference: DG/divergen/custom_solver.py:19-77, D2/solver/build.py:24-75, DG/divergen/ema.py:49-58."""
from .. import _lib as L


def adamw_ema_step(p, g, m, v, ema, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, clip_value=1.0,
                   grad_scale=1.0, ema_decay=0.999, p_bf16=None, lr_scale=None, seg_end=None, found_inf=None):
    n = p.numel()
    assert n % 4 == 0, "arena length must be a multiple of 4"
    n_seg = 0 if lr_scale is None else lr_scale.numel()
    L.check(L.lib().dgx_adamw_ema_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(ema), L.ptr(p_bf16), n, lr,
                                       betas[0], betas[1], eps, weight_decay, clip_value, grad_scale, int(step),
                                       ema_decay, L.ptr(lr_scale), L.ptr(seg_end), n_seg, L.ptr(found_inf),
                                       L.stream()), "dgx_adamw_ema_step")
