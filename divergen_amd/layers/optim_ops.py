"""Fused clip + AdamW + EMA step over flat fp32 arenas.
Reference: DG/divergen/custom_solver.py:19-77, D2/solver/build.py:24-75, DG/divergen/ema.py:49-58."""
from .. import _lib as L


def adamw_ema_step(p, g, m, v, ema, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, clip_value=1.0,
                   grad_scale=1.0, ema_decay=0.999, p_bf16=None, lr_scale=None, seg_end=None, found_inf=None, grad_scale_dev=None):
    """grad_scale_dev: optional device scalar multiplied into the gradient (the full-model norm-clip coefficient of `clip_coef`)."""
    n = p.numel()
    assert n % 4 == 0, "arena length must be a multiple of 4"
    n_seg = 0 if lr_scale is None else lr_scale.numel()
    L.check(L.lib().dgx_adamw_ema_step_scaled(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(ema), L.ptr(p_bf16), n, lr,
                                              betas[0], betas[1], eps, weight_decay, clip_value, grad_scale, L.ptr(grad_scale_dev),
                                              int(step), ema_decay, L.ptr(lr_scale), L.ptr(seg_end), n_seg, L.ptr(found_inf),
                                              L.stream()), "dgx_adamw_ema_step_scaled")


def sgd_ema_step(p, g, buf, ema, step, lr, momentum=0.0, nesterov=False, weight_decay=0.0, clip_value=0.0, grad_scale=1.0,
                 grad_scale_dev=None, ema_decay=0.0, p_bf16=None, lr_scale=None, seg_end=None, found_inf=None):
    """torch.optim.SGD (custom_solver.py:64-68) + EMA over flat fp32 arenas: dgx_sgd_ema_step."""
    n = p.numel()
    assert n % 4 == 0, "arena length must be a multiple of 4"
    n_seg = 0 if lr_scale is None else lr_scale.numel()
    L.check(L.lib().dgx_sgd_ema_step(L.ptr(p), L.ptr(g), L.ptr(buf), L.ptr(ema), L.ptr(p_bf16), n, lr, momentum, int(bool(nesterov)),
                                     weight_decay, clip_value, grad_scale, L.ptr(grad_scale_dev), int(step), ema_decay,
                                     L.ptr(lr_scale), L.ptr(seg_end), n_seg, L.ptr(found_inf), L.stream()), "dgx_sgd_ema_step")


def clip_coef(g, max_norm, grad_scale=1.0):
    """-> float32 (2,) device tensor [min(1, max_norm / (||g * grad_scale|| + 1e-6)), norm]: clip_grad_norm_ over the whole arena
    (custom_solver.py:46-60) without a host read."""
    import torch
    ws = torch.empty(int(L.lib().dgx_clip_coef_workspace_floats()), dtype=torch.float32, device=g.device)
    out = torch.empty(2, dtype=torch.float32, device=g.device)
    L.check(L.lib().dgx_clip_coef_f32(L.ptr(g), g.numel(), grad_scale, float(max_norm), L.ptr(ws), L.ptr(out), L.stream()), "dgx_clip_coef_f32")
    return out
