"""Where do the CenterNet tower's gradients leave the oracle's?  k x [3x3 conv -> GroupNorm(32) -> ReLU] on the product kernels against
torch fp32 autograd of the same layers (and against the same with the activations rounded to bf16 where the product stores bf16), layer
by layer: relative L2 of the input gradient and of every weight gradient.
    python tools/tower_grad_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from divergen_amd.layers import conv_ops  # noqa: E402
from divergen_amd.layers.norm_ops import groupnorm_relu  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
torch.manual_seed(0)
N, C, H, W, K = 2, 256, 32, 32, 4


def rb(x, on):
    return x.to(BF).float() if on else x


def ref_tower(x, ws, bs, gs, betas, storage):
    for w, b, g, be in zip(ws, bs, gs, betas):
        x = rb(torch.nn.functional.conv2d(x, w, b, padding=1), storage)
        x = rb(torch.relu(torch.nn.functional.group_norm(x, 32, g, be, 1e-5)), storage)
    return x


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


x0 = torch.randn(N, C, H, W, device=DEV).to(BF).float()
ws = [(torch.randn(C, C, 3, 3, device=DEV) * 0.01).to(BF).float() for _ in range(K)]
bs = [torch.zeros(C, device=DEV) for _ in range(K)]
gs = [torch.ones(C, device=DEV) for _ in range(K)]
betas = [torch.zeros(C, device=DEV) for _ in range(K)]
for sparse in (False, True):
    go = torch.randn(N, C, H, W, device=DEV)
    if sparse:      # a heat-map-like upstream gradient: a few locations only
        m = torch.zeros(N, 1, H, W, device=DEV)
        m[:, :, ::9, ::7] = 1
        go = go * m
    go = go.to(BF).float()
    res = {}
    for storage in (False, True):
        xr = x0.clone().requires_grad_(True)
        pr = [[t.clone().requires_grad_(True) for t in grp] for grp in (ws, bs, gs, betas)]
        out = ref_tower(xr, *pr, storage)
        out.backward(go)
        res[storage] = (out.detach(), xr.grad, [[t.grad for t in grp] for grp in pr])
    xd = x0.to(BF).to(memory_format=torch.channels_last).requires_grad_(True)
    pd = [[t.clone().requires_grad_(True) for t in grp] for grp in (ws, bs, gs, betas)]
    y = xd
    for w, b, g, be in zip(*pd):
        y = conv_ops.conv3x3(y, w, b, 1)
        y = groupnorm_relu(y, g, be, 32, 1e-5, relu=True)
    y.backward(go.to(BF))
    print("upstream gradient:", "sparse" if sparse else "dense")
    for storage in (False, True):
        out, dx, grads = res[storage]
        tag = "bf16-storage torch" if storage else "fp32 torch        "
        print("  vs %s: out %.3e  dx %.3e" % (tag, rel(y.float(), out), rel(xd.grad.float(), dx)))
        for k in range(K):
            print("      layer %d: dW %.3e  db %.3e  dgamma %.3e  dbeta %.3e" % (k, rel(pd[0][k].grad, grads[0][k]), rel(pd[1][k].grad, grads[1][k]),
                                                                                 rel(pd[2][k].grad, grads[2][k]), rel(pd[3][k].grad, grads[3][k])))
    print("  fp32 torch vs bf16-storage torch (what the storage alone does): dx %.3e  dW0 %.3e" % (rel(res[True][1], res[False][1]), rel(res[True][2][0][0], res[False][2][0][0])))
