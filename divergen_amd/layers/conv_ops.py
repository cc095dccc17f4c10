"""Convolutions of the dense / RoI heads as libdgx GEMMs over channels-last bf16 tensors (no library route, no fp32 mode).

  conv3x3      stride 1: implicit GEMM over a zero-bordered NHWC image (no column matrix), forward, input and weight gradient;
               stride 2 (P6 / P7, R50 down-sampling): im2col (HIP) + the same MFMA GEMM; backward = GEMM + col2im (HIP)
  conv1x1      a Linear over NHWC pixels
  patch_embed  4x4 stride-4 conv = Linear over unfolded 4x4x3 patches (a pure view)
  deconv2x2    ConvTranspose2d(k=2,s=2) = Linear C -> 4*Cout per pixel + pixel shuffle (a view)

Weights keep nn.Conv2d / nn.ConvTranspose2d layouts so reference checkpoints load unchanged.
Inputs and outputs are logical (N,C,H,W) tensors in channels_last memory format (NHWC storage)."""
import math
import os

import torch

from .. import _lib as L
from .gemm_ops import gemm_nt
from .linear_ops import accumulate_grad, linear, notify_ready, shadow, wgrad_into


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _im2col(x_nhwc, stride):
    N, H, W, C = x_nhwc.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    col = torch.empty(N * Ho * Wo, 9 * C, dtype=x_nhwc.dtype, device=x_nhwc.device)
    L.check(L.lib().dgx_im2col3x3(L.ptr(x_nhwc), L.ptr(col), N, H, W, C, stride, L.dtype_code(x_nhwc), L.stream()),
            "dgx_im2col3x3")
    return col, Ho, Wo


def _ohwi_matrix(t):
    """(Cout, 9*Cin) row-major view of a weight-shaped arena tensor stored (Cout, kh, kw, Cin), else None."""
    if t is None or t.dim() != 4:
        return None
    q = t.permute(0, 2, 3, 1)
    return q.reshape(q.shape[0], -1) if q.is_contiguous() else None


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        # x: NHWC contiguous (N,H,W,C); weight (Cout,Cin,3,3); compute dtype = x.dtype
        N, H, W, C = x.shape
        bf = x.dtype == torch.bfloat16
        w = shadow(weight) if bf else weight.to(x.dtype)
        wk = _ohwi_matrix(w) if bf else None          # arena weights stored (Cout,kh,kw,Cin): the GEMM operand as is
        wm = wk.t() if wk is not None else w.permute(2, 3, 1, 0).reshape(9 * C, -1)     # (ky,kx,ci) x co
        col, Ho, Wo = _im2col(x, stride)
        if not (bf and x.is_cuda and w.shape[0] % 8 == 0 and C % 8 == 0):
            raise L.DgxError("conv3x3: bf16 GPU input with Cin, Cout multiples of 8 required (got %s %s, Cin %d, Cout %d)"
                             % (x.dtype, x.device, C, w.shape[0]))
        own = True
        if wk is None:
            wk = wm.t().contiguous()
        y = gemm_nt(col, wk, shadow(bias) if bias is not None else None)
        wt = getattr(weight, "_dgx16t", None)      # (9 Cin, Cout): the arena's transposed twin, else a copy
        plain = wt is not None and not getattr(weight, "_dgx16t_flipped", False) and _ohwi_matrix(shadow(weight)) is not None
        wm = wt if plain else wm.contiguous()
        ctx.own = own
        ctx.save_for_backward(col, wm)
        ctx.weight, ctx.bias = weight, bias
        ctx.cfg = (N, H, W, C, Ho, Wo, stride, weight.dtype, bias is not None, weight.shape)
        return y.view(N, Ho, Wo, -1)

    @staticmethod
    def backward(ctx, gy):
        col, wm = ctx.saved_tensors
        N, H, W, C, Ho, Wo, stride, wdt, has_bias, wshape = ctx.cfg
        g2 = gy.reshape(N * Ho * Wo, -1).to(col.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            dcol = gemm_nt(g2.contiguous(), wm)
            gx = torch.empty(N, H, W, C, dtype=col.dtype, device=col.device)
            L.check(L.lib().dgx_col2im3x3(L.ptr(dcol), L.ptr(gx), N, H, W, C, stride, L.dtype_code(gx), L.stream()),
                    "dgx_col2im3x3")
        gphys = _ohwi_matrix(ctx.weight.grad) if (ctx.weight.is_leaf and ctx.weight.grad is not None) else None
        if (ctx.needs_input_grad[1] and gphys is not None and col.dtype == torch.bfloat16 and gphys.dtype == torch.float32
                and getattr(ctx.weight, "_dgx16", None) is not None):
            # weight gradient accumulated straight into the arena, in its stored (Cout, kh*kw*Cin) order
            wgrad_into(gphys, g2.contiguous(), col, beta=1.0)
            notify_ready(ctx.weight)
        elif ctx.needs_input_grad[1]:
            def wgrad():
                if col.dtype != torch.bfloat16:
                    raise L.DgxError("conv3x3 weight gradient: bf16 operands required (got %s): the only GEMM on the path is "
                                     "libdgx's" % col.dtype)
                g = torch.empty(9 * C, g2.shape[1], dtype=torch.float32, device=col.device)
                wgrad_into(g, col, g2.contiguous(), beta=0.0)      # M-split MFMA kernel for long M
                return g.view(3, 3, C, -1).permute(3, 2, 0, 1)
            gw = accumulate_grad(ctx.weight, wgrad)
        if has_bias and ctx.needs_input_grad[2]:
            b = ctx.bias
            if (b.is_leaf and b.grad is not None and b.grad.dtype == torch.float32 and getattr(b, "_dgx16", None) is not None
                    and g2.dtype == torch.bfloat16 and g2.shape[1] % 8 == 0):
                from .swin_block import colsum_into
                colsum_into(b.grad, g2.contiguous())       # bias gradient summed straight into the arena
                notify_ready(b)
            else:
                gb = accumulate_grad(b, lambda: torch.sum(g2, 0, dtype=torch.float32))
        return gx, gw, gb, None


def _pad_image(x_nhwc):
    """Zero-bordered copy for the implicit convolution (dgx_conv3x3_pad): ((W+3) + N (H+2) (W+2) + (W+3), C)."""
    N, H, W, C = x_nhwc.shape
    rows = int(L.lib().dgx_conv3x3_pad_rows(N, H, W))
    xp = torch.empty(rows, C, dtype=x_nhwc.dtype, device=x_nhwc.device)
    L.check(L.lib().dgx_conv3x3_pad(L.ptr(x_nhwc), L.ptr(xp), N, H, W, C, L.stream()), "dgx_conv3x3_pad")
    return xp


def _pad_image_relu_grad(g_nhwc, y_nhwc):
    """Zero-bordered copy of g masked by ReLU'(y) (y = the layer's ReLU-ed output): dgx_conv3x3_pad_relu_grad."""
    N, H, W, C = g_nhwc.shape
    rows = int(L.lib().dgx_conv3x3_pad_rows(N, H, W))
    gp = torch.empty(rows, C, dtype=g_nhwc.dtype, device=g_nhwc.device)
    L.check(L.lib().dgx_conv3x3_pad_relu_grad(L.ptr(g_nhwc), L.ptr(y_nhwc), L.ptr(gp), N, H, W, C, L.stream()), "dgx_conv3x3_pad_relu_grad")
    return gp


def _flipped_twin(weight, w16):
    """(Cin, 3, 3, Cout) bf16 with [ci][ey][ex][co] = w[co][ci][2-ey][2-ex]: the arena's twin when it has one, else built here."""
    if getattr(weight, "_dgx16t_flipped", False):
        return weight._dgx16t
    return w16.flip(2, 3).permute(1, 2, 3, 0).reshape(w16.shape[1], -1).contiguous()


_SPLITK = True      # split-K workspace handed to the GEMM (small-M convolutions of the coarse levels)


class _NoWorkspace:
    @staticmethod
    def data_ptr():
        return None

    @staticmethod
    def numel():
        return 0


def _splitk_workspace(dev):
    """The GEMM module's per-device split-K scratch (small FPN levels leave most CUs idle without it)."""
    from . import gemm_ops
    return gemm_ops._workspace(dev) if _SPLITK else _NoWorkspace


class _Conv3x3Implicit(torch.autograd.Function):
    """3x3 / pad 1 / stride 1 convolution on libdgx's implicit GEMM (no column matrix): forward, input gradient and weight
    gradient all read zero-bordered copies of the NHWC tensors; bf16, Cin % 64 == 0, Cout % 8 == 0."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        N, H, W, C = x.shape
        w16 = shadow(weight)
        wk = _ohwi_matrix(w16)
        if wk is None:
            wk = w16.permute(0, 2, 3, 1).reshape(w16.shape[0], -1).contiguous()
        Co = wk.shape[0]
        xp = _pad_image(x)
        y = torch.empty(N, H, W, Co, dtype=torch.bfloat16, device=x.device)
        b16 = shadow(bias) if bias is not None else None
        ws = _splitk_workspace(x.device)
        L.check(L.lib().dgx_conv3x3_gemm(L.ptr(xp), wk.data_ptr(), L.ptr(b16), L.ptr(y), N, H, W, C, Co, int(relu),
                                         ws.data_ptr(), ws.numel(), L.stream()), "dgx_conv3x3_gemm")
        ctx.save_for_backward(xp, y if relu else None)
        ctx.weight, ctx.bias, ctx.w16 = weight, bias, w16
        ctx.cfg = (N, H, W, C, Co, relu)
        return y

    @staticmethod
    def backward(ctx, gy):
        xp, yact = ctx.saved_tensors
        N, H, W, C, Co, relu = ctx.cfg
        weight, bias = ctx.weight, ctx.bias
        lib = L.lib()
        g2 = gy.to(torch.bfloat16).contiguous()
        gx = gw = gb = None
        masked_in_pad = relu and Co % 64 == 0 and yact.is_contiguous()
        if relu and not masked_in_pad:
            g2 = g2 * (yact > 0)
        # ReLU' rides in the zero-bordered copy (one launch instead of compare + multiply + copy); everything behind reads gp
        gp = _pad_image_relu_grad(g2, yact) if masked_in_pad else _pad_image(g2)
        if ctx.needs_input_grad[0]:
            wf = _flipped_twin(weight, ctx.w16)                       # (Cin, 9 Cout)
            gsrc = gp
            if Co % 64:                                                # K-tiles of the input-gradient GEMM hold 64 channels of one tap
                cp = -(-Co // 64) * 64
                gsrc = _pad_image(torch.nn.functional.pad(g2, (0, cp - Co)))
                wf = torch.nn.functional.pad(wf.view(C, 9, Co), (0, cp - Co)).reshape(C, 9 * cp).contiguous()
                Ck = cp
            else:
                Ck = Co
            gx = torch.empty(N, H, W, C, dtype=torch.bfloat16, device=g2.device)
            ws = _splitk_workspace(g2.device)
            L.check(lib.dgx_conv3x3_gemm(L.ptr(gsrc), wf.data_ptr(), None, L.ptr(gx), N, H, W, Ck, C, 0, ws.data_ptr(), ws.numel(),
                                         L.stream()), "dgx_conv3x3_gemm")
        bias_arena = (bias is not None and ctx.needs_input_grad[2] and bias.is_leaf and bias.grad is not None
                      and bias.grad.dtype == torch.float32 and getattr(bias, "_dgx16", None) is not None)
        bias_done = False
        if ctx.needs_input_grad[1]:
            gphys = _ohwi_matrix(weight.grad) if (weight.is_leaf and weight.grad is not None) else None
            if gphys is not None and gphys.dtype == torch.float32 and getattr(weight, "_dgx16", None) is not None:
                # weight gradient (nine tap problems over the two padded images) AND the bias gradient (dy^T 1) in one launch
                gbp = bias.grad if bias_arena else None
                ws = torch.empty(max(int(lib.dgx_conv3x3_wgrad_bias_workspace_bytes(N, H, W, C, Co)), 16), dtype=torch.uint8, device=g2.device)
                L.check(lib.dgx_conv3x3_wgrad_bias(L.ptr(gp), L.ptr(xp), gphys.data_ptr(), L.ptr(gbp), N, H, W, C, Co, 1.0, L.ptr(ws),
                                                   L.stream()), "dgx_conv3x3_wgrad_bias")
                notify_ready(weight)
                if bias_arena:
                    bias_done = True
                    notify_ready(bias)
            else:
                ws = torch.empty(max(int(lib.dgx_conv3x3_wgrad_workspace_bytes(N, H, W, C, Co)), 16), dtype=torch.uint8, device=g2.device)

                def wgrad():
                    g = torch.empty(Co, 3, 3, C, dtype=torch.float32, device=g2.device)
                    L.check(lib.dgx_conv3x3_wgrad(L.ptr(gp), L.ptr(xp), L.ptr(g), N, H, W, C, Co, 0.0, L.ptr(ws), L.stream()),
                            "dgx_conv3x3_wgrad")
                    return g.permute(0, 3, 1, 2)
                gw = accumulate_grad(weight, wgrad)
        if bias is not None and ctx.needs_input_grad[2] and not bias_done:
            g2f = (g2 * (yact > 0) if masked_in_pad else g2).view(-1, Co)      # (outside the arena: the masked gradient is needed unpadded)
            if bias_arena:
                from .swin_block import colsum_into
                colsum_into(bias.grad, g2f)
                notify_ready(bias)
            else:
                gb = accumulate_grad(bias, lambda: torch.sum(g2f, 0, dtype=torch.float32))
        return gx, gw, gb, None


_IMPLICIT = True      # stride-1 3x3 convolutions with >= 64 channels run as implicit GEMMs; the column-matrix form serves stride 2 and narrow inputs


class _Conv3x3Group(torch.autograd.Function):
    """3x3 / pad 1 / stride 1 convolution whose output channels are the rows of an ARENA PARAMETER GROUP (solver.FlatArena:
    the group's weights sit back to back, rounded up to `pad_to` zero rows; CenterNetHead's agn_hm (1) | bbox_pred (4) | zeros):
    forward, input gradient and weight / bias gradient read and write the group's views (`_dgx16g`, the tap-flipped
    `_dgx16tg`, `_dgxgg`) -- ONE implicit GEMM each, no torch.cat of weights, no padded copies, no per-call flipped twin,
    gradients accumulated in place (round 2 rebuilt all of those per FPN level: ~20 small launches per level and direction)."""

    @staticmethod
    def forward(ctx, x, w_handle, b_handle):
        N, H, W, C = x.shape
        wk = w_handle._dgx16g                        # (rows_pad, 9 Cin) bf16, K-order (kh, kw, ci)
        Co = wk.shape[0]
        xp = _pad_image(x)
        y = torch.empty(N, H, W, Co, dtype=torch.bfloat16, device=x.device)
        ws = _splitk_workspace(x.device)
        L.check(L.lib().dgx_conv3x3_gemm(L.ptr(xp), wk.data_ptr(), b_handle._dgx16g.data_ptr(), L.ptr(y), N, H, W, C, Co, 0,
                                         ws.data_ptr(), ws.numel(), L.stream()), "dgx_conv3x3_gemm")
        ctx.save_for_backward(xp)
        ctx.w_handle, ctx.b_handle = w_handle, b_handle
        ctx.cfg = (N, H, W, C, Co)
        return y

    @staticmethod
    def backward(ctx, gy):
        (xp,) = ctx.saved_tensors
        N, H, W, C, Co = ctx.cfg
        w_handle, b_handle = ctx.w_handle, ctx.b_handle
        lib = L.lib()
        g2 = gy.to(torch.bfloat16).contiguous()
        gp = _pad_image(g2)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(N, H, W, C, dtype=torch.bfloat16, device=g2.device)
            ws = _splitk_workspace(g2.device)
            L.check(lib.dgx_conv3x3_gemm(L.ptr(gp), w_handle._dgx16tg.data_ptr(), None, L.ptr(gx), N, H, W, Co, C, 0, ws.data_ptr(),
                                         ws.numel(), L.stream()), "dgx_conv3x3_gemm")
        ws = torch.empty(max(int(lib.dgx_conv3x3_wgrad_bias_workspace_bytes(N, H, W, C, Co)), 16), dtype=torch.uint8, device=g2.device)
        L.check(lib.dgx_conv3x3_wgrad_bias(L.ptr(gp), L.ptr(xp), w_handle._dgxgg.data_ptr(), b_handle._dgxgg.data_ptr(), N, H, W, C, Co, 1.0,
                                           L.ptr(ws), L.stream()), "dgx_conv3x3_wgrad_bias")
        for q in tuple(w_handle._dgx_group_members) + tuple(b_handle._dgx_group_members):
            notify_ready(q)
        return gx, None, None


def conv3x3_group_usable(x, w_handle, b_handle):
    """The grouped predictor convolution needs the arena (group views with a tap-flipped twin, fp32 gradient rows), a GPU
    bf16-able input with Cin % 64 == 0 and a group padded to 64 rows (whole K-tiles for the input-gradient convolution)."""
    wg, wt, gg = (getattr(w_handle, n, None) for n in ("_dgx16g", "_dgx16tg", "_dgxgg"))
    bg, bgg = getattr(b_handle, "_dgx16g", None), getattr(b_handle, "_dgxgg", None)
    return (_IMPLICIT and x.is_cuda and torch.is_grad_enabled() and wg is not None and wt is not None and gg is not None and bg is not None
            and bgg is not None and getattr(w_handle, "_dgx16tg_flipped", False) and wg.shape[0] % 64 == 0 and x.shape[1] % 64 == 0
            and gg.dtype == torch.float32)


def conv3x3_group(x, w_handle, b_handle):
    """x logical (N, C, H, W) -> logical (N, rows_pad, H, W) (NHWC storage): rows of the group in member order, then zeros."""
    xh = _nhwc(x).to(torch.bfloat16)
    with torch.autocast("cuda", enabled=False):
        y = _Conv3x3Group.apply(xh, w_handle, b_handle)
    return y.permute(0, 3, 1, 2)


_CONV_MULTI = True      # a tower layer's convolution over all levels with shared padded-copy launches
_WGRAD_MULTI = True     # that layer's weight gradient over all levels in one partial + one reduce launch


def _pad_images(xs):
    """Zero-bordered copies of several NHWC images with the same channel count in ONE launch (dgx_conv3x3_pad_multi)."""
    n = len(xs)
    items = (L.PadItem * n)()
    outs = []
    lib = L.lib()
    for i, x in enumerate(xs):
        N, H, W, C = x.shape
        xp = torch.empty(int(lib.dgx_conv3x3_pad_rows(N, H, W)), C, dtype=x.dtype, device=x.device)
        items[i].x, items[i].xpad, items[i].N, items[i].H, items[i].W = L.ptr(x), L.ptr(xp), N, H, W
        outs.append(xp)
    L.check(lib.dgx_conv3x3_pad_multi(items, n, xs[0].shape[-1], L.stream()), "dgx_conv3x3_pad_multi")
    return outs


class _ConvOperands:
    """The arena views one 3x3 convolution needs in all three passes: wk (Cout, 9 Cin) bf16 in K-order (kh, kw, ci), wf its tap-flipped
    twin (Cin, 9 Cout), gw the fp32 gradient rows (Cout, 9 Cin), bias / bias gradient, and the parameters to signal."""

    def __init__(self, wk, wf, gw, b16, gb, params):
        self.wk, self.wf, self.gw, self.b16, self.gb, self.params = wk, wf, gw, b16, gb, params


def conv_operands(weight, bias):
    """_ConvOperands of an arena-resident stride-1 3x3 convolution (a Conv2d weight or the handle of a parameter group), else None."""
    if getattr(weight, "_dgx16g", None) is not None:          # parameter group (CenterNet predictors)
        if not conv3x3_group_usable_w(weight, bias):
            return None
        return _ConvOperands(weight._dgx16g, weight._dgx16tg, weight._dgxgg, bias._dgx16g, bias._dgxgg,
                             tuple(weight._dgx_group_members) + tuple(bias._dgx_group_members))
    if bias is None or not (weight.is_leaf and bias.is_leaf and weight.grad is not None and bias.grad is not None
                            and getattr(weight, "_dgx16", None) is not None and getattr(bias, "_dgx16", None) is not None
                            and getattr(weight, "_dgx16t_flipped", False) and weight.grad.dtype == torch.float32
                            and bias.grad.dtype == torch.float32):
        return None
    wk, gw = _ohwi_matrix(shadow(weight)), _ohwi_matrix(weight.grad)
    if wk is None or gw is None or wk.shape[0] % 64 or weight.shape[1] % 64:
        return None
    return _ConvOperands(wk, weight._dgx16t, gw, shadow(bias), bias.grad, (weight, bias))


def conv3x3_group_usable_w(w_handle, b_handle):
    wg, wt, gg = (getattr(w_handle, n, None) for n in ("_dgx16g", "_dgx16tg", "_dgxgg"))
    bg, bgg = getattr(b_handle, "_dgx16g", None), getattr(b_handle, "_dgxgg", None)
    return (wg is not None and wt is not None and gg is not None and bg is not None and bgg is not None
            and getattr(w_handle, "_dgx16tg_flipped", False) and wg.shape[0] % 64 == 0 and gg.dtype == torch.float32)


_CONV_GEMM_MULTI = True      # one grouped implicit-GEMM launch for the levels


def _conv_gemms(xps, ys, nhw, w, b16, Cin, Cout):
    """y_i = conv3x3(zero-bordered x_i, w) (+ bias) for the images of a tower layer: ONE grouped launch (dgx_conv3x3_gemm_multi: <= 6
    images, Cout <= 256; the small levels' tiles run beside the large level's instead of in split-K launches of their own), else one
    implicit GEMM per image."""
    lib = L.lib()
    n = len(xps)
    if _CONV_GEMM_MULTI and 1 < n <= 6 and Cout <= 256:
        items = (L.ConvItem * n)()
        for i, (xp, y, (N, H, W)) in enumerate(zip(xps, ys, nhw)):
            items[i].xpad, items[i].y, items[i].N, items[i].H, items[i].W = L.ptr(xp), L.ptr(y), N, H, W
        L.check(lib.dgx_conv3x3_gemm_multi(items, n, w.data_ptr(), b16.data_ptr() if b16 is not None else None, Cin, Cout, 0, L.stream()),
                "dgx_conv3x3_gemm_multi")
        return
    for xp, y, (N, H, W) in zip(xps, ys, nhw):
        ws = _splitk_workspace(xp.device)
        L.check(lib.dgx_conv3x3_gemm(L.ptr(xp), w.data_ptr(), b16.data_ptr() if b16 is not None else None, L.ptr(y), N, H, W, Cin, Cout, 0,
                                     ws.data_ptr(), ws.numel(), L.stream()), "dgx_conv3x3_gemm")


class _Conv3x3Multi(torch.autograd.Function):
    """One 3x3 / pad 1 / stride 1 convolution (shared weights) over SEVERAL NHWC bf16 images -- a CenterNet tower layer over the FPN
    levels: the zero-bordered copies of all inputs (forward) / output gradients (backward) are one launch each; the implicit GEMMs
    and the weight-gradient launches stay per image.  Gradients accumulate in place in the arena (beta = 1), in level order."""

    @staticmethod
    def forward(ctx, ops, *xs):
        lib = L.lib()
        xps = _pad_images(xs)
        Co = ops.wk.shape[0]
        ys = [torch.empty(x.shape[0], x.shape[1], x.shape[2], Co, dtype=torch.bfloat16, device=x.device) for x in xs]
        _conv_gemms(xps, ys, [tuple(x.shape[:3]) for x in xs], ops.wk, ops.b16, xs[0].shape[3], Co)
        ctx.save_for_backward(*xps)
        ctx.ops, ctx.shapes = ops, [tuple(x.shape) for x in xs]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        xps, ops = ctx.saved_tensors, ctx.ops
        lib = L.lib()
        Co = ops.wk.shape[0]
        g2s = []
        for gy, (N, H, W, C) in zip(gys, ctx.shapes):
            g2s.append(gy.to(torch.bfloat16).contiguous() if gy is not None else torch.zeros(N, H, W, Co, dtype=torch.bfloat16, device=xps[0].device))
        gps = _pad_images(g2s)
        need = [i for i in range(len(gps)) if ctx.needs_input_grad[1 + i]]
        gxs = [torch.empty(N, H, W, C, dtype=torch.bfloat16, device=gps[0].device) if i in need else None
               for i, (N, H, W, C) in enumerate(ctx.shapes)]
        if need:
            _conv_gemms([gps[i] for i in need], [gxs[i] for i in need], [ctx.shapes[i][:3] for i in need], ops.wf, None, Co, ctx.shapes[0][3])
        C = ctx.shapes[0][3]
        if _WGRAD_MULTI and len(gps) <= 6:
            items = (L.ConvWgradItem * len(gps))()
            for i, (gp, xp, (N, H, W, _)) in enumerate(zip(gps, xps, ctx.shapes)):
                items[i].dypad, items[i].xpad, items[i].N, items[i].H, items[i].W = L.ptr(gp), L.ptr(xp), N, H, W
            ws = torch.empty(max(int(lib.dgx_conv3x3_wgrad_bias_multi_workspace_bytes(items, len(gps), C, Co)), 16), dtype=torch.uint8,
                             device=gps[0].device)
            L.check(lib.dgx_conv3x3_wgrad_bias_multi(items, len(gps), ops.gw.data_ptr(), ops.gb.data_ptr(), C, Co, 1.0, L.ptr(ws), L.stream()),
                    "dgx_conv3x3_wgrad_bias_multi")
        else:
            for i, (gp, xp, (N, H, W, _)) in enumerate(zip(gps, xps, ctx.shapes)):
                ws = torch.empty(max(int(lib.dgx_conv3x3_wgrad_bias_workspace_bytes(N, H, W, C, Co)), 16), dtype=torch.uint8, device=gp.device)
                L.check(lib.dgx_conv3x3_wgrad_bias(L.ptr(gp), L.ptr(xp), ops.gw.data_ptr(), ops.gb.data_ptr(), N, H, W, C, Co, 1.0, L.ptr(ws),
                                                   L.stream()), "dgx_conv3x3_wgrad_bias")
        for q in ops.params:
            notify_ready(q)
        return (None,) + tuple(gxs)


def conv3x3_multi(xs, weight, bias):
    """list of logical (N, C, H, W) tensors (channels-last storage) through the SAME stride-1 3x3 convolution -> list of logical
    (N, Cout, H, W); None when the layer is not arena-resident / the inputs do not qualify (callers then go level by level)."""
    if not (_IMPLICIT and _CONV_MULTI and torch.is_grad_enabled() and 1 < len(xs) <= 8 and all(x.is_cuda and x.shape[1] % 64 == 0 for x in xs)):
        return None
    ops = conv_operands(weight, bias)
    if ops is None:
        return None
    xhs = [_nhwc(x).to(torch.bfloat16) for x in xs]
    with torch.autocast("cuda", enabled=False):
        ys = _Conv3x3Multi.apply(ops, *xhs)
    return [y.permute(0, 3, 1, 2) for y in ys]


MIN_COUT = 8  # the MFMA GEMM writes 8-column chunks: 1- and 4-channel predictors run with zero-padded output channels


def _pad_cout(weight, bias):
    co = weight.shape[0]
    if co >= MIN_COUT:
        return weight, bias, co
    pad = MIN_COUT - co
    weight = torch.cat([weight, weight.new_zeros((pad,) + tuple(weight.shape[1:]))], 0)
    if bias is not None:
        bias = torch.cat([bias, bias.new_zeros(pad)], 0)
    return weight, bias, co


def conv3x3(x, weight, bias=None, stride=1, relu=False):
    """x logical (N,C,H,W) -> logical (N,Cout,Ho,Wo), NHWC storage both sides.  relu: fused into the implicit-GEMM epilogue
    where that path applies, applied afterwards otherwise."""
    xh = _nhwc(x).to(torch.bfloat16)
    weight, bias, co = _pad_cout(weight, bias)
    implicit = _IMPLICIT and stride == 1 and xh.is_cuda and xh.shape[-1] % 64 == 0 and weight.shape[0] % 8 == 0
    with torch.autocast("cuda", enabled=False):
        if implicit:
            y = _Conv3x3Implicit.apply(xh, weight, bias, relu)
        else:
            y = _Conv3x3.apply(xh, weight, bias, stride)
            if relu:
                y = torch.relu(y)
    return y[..., :co].permute(0, 3, 1, 2)


def conv1x1(x, weight, bias=None):
    return linear(_nhwc(x), weight, bias).permute(0, 3, 1, 2)     # arena path: no casts, fp32 wgrad in place


def preprocess_patch_rows(images, mean, std, size_divisibility=0, patch=4):
    """list of uint8 (3,h,w) CUDA images -> (PatchRows of the zero-padded, normalised batch, image_sizes): one
    dgx_preprocess_patches launch per image, no fp32 batch tensor (rcnn.py:220-227 + image_list.py:59-110 + the unfold of
    the stride-4 PatchEmbed convolution)."""
    from ..structures import PatchRows
    sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
    H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
    st = max(int(size_divisibility), 1)
    st = st * patch // math.gcd(st, patch)
    H, W = -(-H // st) * st, -(-W // st) * st
    Hp, Wp = H // patch, W // patch
    rows = torch.empty(len(images), Hp * Wp, 3 * patch * patch, dtype=torch.bfloat16, device=images[0].device)
    m, s_ = mean.reshape(-1).float().contiguous(), std.reshape(-1).float().contiguous()
    for b, im in enumerate(images):
        im = im.contiguous()
        L.check(L.lib().dgx_preprocess_patches(L.ptr(im), sizes[b][0], sizes[b][1], L.ptr(m), L.ptr(s_), rows[b].data_ptr(), Hp, Wp,
                                               patch, L.stream()), "dgx_preprocess_patches")
    return PatchRows(rows, Hp, Wp), sizes


def patch_embed_rows(pr, weight, bias):
    """PatchRows -> tokens (B, Hp*Wp, embed): the PatchEmbed projection as a Linear over the prepared rows."""
    return linear(pr.rows, weight, bias), pr.Hp, pr.Wp


def patch_embed4x4(x, weight, bias, patch=4):
    """x (B,3,H,W) any layout, H,W multiples of `patch` -> tokens (B, H/4*W/4, embed)."""
    B, Cin, H, W = x.shape
    Hp, Wp = H // patch, W // patch
    u = x.reshape(B, Cin, Hp, patch, Wp, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, Hp * Wp, Cin * patch * patch)
    return linear(u, weight, bias), Hp, Wp          # arena path: (embed, 3,4,4) is a (embed, 48) Linear weight


class _Deconv2x2(torch.autograd.Function):
    """ConvTranspose2d(k 2, s 2) as ONE MFMA GEMM per direction over the weight as stored, (Cin, Cout*2*2):
    y[pix][(co,dy,dx)] = x[pix] . W[:, (co,dy,dx)] + b[co]  (+ ReLU, which commutes with the pixel shuffle that follows);
    forward reads the arena's transposed twin (4 Cout, Cin), the input gradient the stored matrix itself, the weight gradient
    is x^T dy in the stored layout (accumulated in place in the arena).  The pixel shuffle is one kernel each way
    (dgx_deconv2x2_shuffle / _unshuffle_relu_grad: the backward one carries the ReLU'), the bias gradient a column sum of the
    un-shuffled gradient (dgx_colsum_bf16) folded over the four sub-pixels."""

    @staticmethod
    def forward(ctx, x2, weight, bias, relu, geom):
        N, H, W = geom
        w16 = shadow(weight)
        Cin = w16.shape[0]
        Cout = weight.shape[1]
        wt = getattr(weight, "_dgx16t", None)
        if wt is None or getattr(weight, "_dgx16t_flipped", False):
            wt = w16.reshape(Cin, -1).t().contiguous()
        b4 = shadow(bias).repeat_interleave(4) if bias is not None else None
        from .gemm_ops import gemm_nt_act
        y2 = gemm_nt_act(x2, wt, b4, relu)
        out = torch.empty(N, 2 * H, 2 * W, Cout, dtype=torch.bfloat16, device=x2.device)
        L.check(L.lib().dgx_deconv2x2_shuffle(L.ptr(y2), L.ptr(out), N, H, W, Cout, L.stream()), "dgx_deconv2x2_shuffle")
        ctx.save_for_backward(x2, out if relu else None)
        ctx.weight, ctx.bias, ctx.w16, ctx.relu, ctx.geom = weight, bias, w16, relu, (N, H, W, Cout)
        return out

    @staticmethod
    def backward(ctx, gy):
        x2, yout = ctx.saved_tensors
        weight, bias, w16 = ctx.weight, ctx.bias, ctx.w16
        N, H, W, Cout = ctx.geom
        gy = gy.to(torch.bfloat16).contiguous()
        g2 = torch.empty(N * H * W, 4 * Cout, dtype=torch.bfloat16, device=gy.device)
        L.check(L.lib().dgx_deconv2x2_unshuffle_relu_grad(L.ptr(gy), L.ptr(yout) if ctx.relu else None, L.ptr(g2), N, H, W, Cout, L.stream()),
                "dgx_deconv2x2_unshuffle_relu_grad")
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm_nt(g2, w16.reshape(w16.shape[0], -1))           # (M, 4 Cout) x (Cin, 4 Cout)^T
        if ctx.needs_input_grad[1]:
            def into(g):
                wgrad_into(g.view(g.shape[0], -1), x2, g2)             # (Cin, 4 Cout) = x^T dy
            def outside_arena():           # a deconvolution weight that does not live in a parameter arena: own GEMM into a temporary
                g = torch.empty(weight.shape[0], weight.numel() // weight.shape[0], dtype=torch.float32, device=g2.device)
                wgrad_into(g, x2, g2, beta=0.0)
                return g.view(weight.shape)
            gw = accumulate_grad(weight, outside_arena, gemm_into=into)
        if bias is not None and ctx.needs_input_grad[2]:
            def colsums():
                from .swin_block import colsum_into
                s4 = torch.empty(4 * Cout, dtype=torch.float32, device=g2.device)
                colsum_into(s4, g2, beta=0.0)
                return s4.view(Cout, 4).sum(1)
            gb = accumulate_grad(bias, colsums)
        return gx, gw, gb, None, None


def deconv2x2(x, weight, bias, relu=False):
    """ConvTranspose2d(kernel 2, stride 2): weight (Cin,Cout,2,2)."""
    xh = _nhwc(x)
    N, H, W, Cin = xh.shape
    Cout = weight.shape[1]
    if not (xh.is_cuda and Cin % 8 == 0 and Cout % 8 == 0):
        raise L.DgxError("deconv2x2: GPU input with Cin %% 8 == 0 and Cout %% 8 == 0 required (Cin %d, Cout %d, %s)" % (Cin, Cout, xh.device))
    with torch.autocast("cuda", enabled=False):
        y = _Deconv2x2.apply(xh.reshape(-1, Cin).to(torch.bfloat16), weight, bias, relu, (N, H, W))
    return y.permute(0, 3, 1, 2)


class Conv2d(torch.nn.Conv2d):
    """nn.Conv2d parameters, GEMM-path forward (1x1, and 3x3 pad 1 stride 1|2)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if self.kernel_size == (3, 3) and self.groups == 1:
            self.weight._dgx_ohwi = True       # FlatArena stores it (Cout, kh, kw, Cin): see solver.FlatArena.view
            self.weight._dgx_flip = self.stride == (1, 1)      # its bf16 twin: tap-flipped (Cin, kh, kw, Cout), the input-gradient operand
        if self.kernel_size == (1, 1) and self.groups == 1 and self.out_channels % 8:
            # a 1x1 convolution is a Linear over pixels: widths that are not multiples of 8 (the 1-channel mask predictor) get
            # zero rows up to 8 inside the arena (layers.linear_ops.shadow_padded)
            from .linear_ops import pad8
            self.weight._dgx_pad_rows = pad8(self.out_channels)
            if self.bias is not None:
                self.bias._dgx_pad_rows = pad8(self.out_channels)

    def forward(self, x, relu=False):
        """relu=True: the activation that follows the layer, fused into the GEMM epilogue where the path has one."""
        k, s, p = self.kernel_size, self.stride, self.padding
        if k == (1, 1) and s == (1, 1) and p == (0, 0):
            y = conv1x1(x, self.weight, self.bias)
            return torch.relu(y) if relu else y
        if k == (3, 3) and p == (1, 1) and s[0] == s[1] and s[0] in (1, 2) and self.groups == 1:
            return conv3x3(x, self.weight, self.bias, s[0], relu=relu)
        raise L.DgxError("Conv2d %s/%s/%s is not on the hot path and has no HIP implementation" % (k, s, p))


class ConvTranspose2d(torch.nn.ConvTranspose2d):
    def forward(self, x, relu=False):
        if self.kernel_size == (2, 2) and self.stride == (2, 2) and self.padding == (0, 0):
            return deconv2x2(x, self.weight, self.bias, relu=relu)
        raise L.DgxError("only ConvTranspose2d(k=2, s=2) is built")
