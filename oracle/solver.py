"""Oracle: parameter update of the training step on torch-CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
  build_custom_optimizer   DG/divergen/custom_solver.py:19-77 (AdamW, one group per tensor, no
                           per-group weight_decay => optimizer default), with
                           maybe_add_gradient_clipping D2/solver/build.py:24-75: per-parameter
                           clip_grad_value_(p, CLIP_VALUE=1.0) before the step;
  torch.optim.AdamW        (decoupled decay, bias correction, eps outside the sqrt scaling);
  ModelEma.update          DG/divergen/ema.py:49-58  (ema = ema*decay + (1-decay)*model, all
                           state-dict entries, called BEFORE the optimizer step: train_net.py:262-264);
  WarmupCosineLR           D2/solver/lr_scheduler.py:171-238.
"""
import math

import torch


def warmup_cosine_lr(base_lr, it, max_iters, warmup_iters, warmup_factor, method="linear"):
    if it >= warmup_iters:
        wf = 1.0
    elif method == "constant":
        wf = warmup_factor
    else:
        a = it / warmup_iters
        wf = warmup_factor * (1 - a) + a
    return base_lr * wf * 0.5 * (1.0 + math.cos(math.pi * it / max_iters))


def adamw_clip_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=1e-4, clip=1.0):
    """One in-place AdamW update with value clipping.  step is 1-based (after increment)."""
    g = g.clamp(-clip, clip)
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def ema_update(ema, model, decay):
    ema.copy_(ema * decay + (1.0 - decay) * model)
