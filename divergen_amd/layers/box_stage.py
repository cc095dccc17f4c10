"""One cascade stage of the box head as ONE autograd node with a hand-written backward.

Reference: DeticCascadeROIHeads._run_stage (DG/divergen/modeling/roi_heads/detic_roi_heads.py:396-414: _ScaleGradient ->
FastRCNNConvFCHead (D2/modeling/roi_heads/box_head.py:26-98: flatten, fc1, ReLU, fc2, ReLU) -> DeticFastRCNNOutputLayers
(detic_fast_rcnn.py:437-466: cls_score, bbox_pred) -> losses (:160-304)).

The composed form is ~12 autograd nodes forward (two slices, two ReLUs, a gradient scale, four Linears, the loss) and ~60 small
launches backward (slice / ReLU / AccumulateGrad bookkeeping around four GEMM pairs).  Here:
  forward   fc1 and fc2 = MFMA GEMM with bias + ReLU in the read-out; cls_score | bbox_pred = ONE GEMM over the arena group
            (1454 + 4 rows -> 1464); dgx_detic_losses_strided reads logits and deltas out of that joint output and writes the
            joint, K-padded gradient operand;
  backward  dgx_detic_grad_scale (the two loss gradients' scales, in place) -> three input-gradient GEMMs, the two through a
            ReLU carrying ReLU' in their read-out (DGX_EPI_RELU_GRAD) -> ONE grouped weight-gradient launch (fc1, fc2,
            predictor pair) that also produces the three bias gradients, accumulated in fp32 straight into the gradient arena.
Used when every parameter of the stage lives in a FlatArena (training); anything else takes the composed path."""
import torch

from .. import _lib as L
from . import gemm_ops as G
from .linear_ops import BF16, notify_ready, shadow, shadow_t, wgrad_grouped
from .swin_block import arena_resident


class _BoxStageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gt_classes, class_w, prop, gtb, src, cfg, fc1w, fc1b, fc2w, fc2b, clsw, clsb, boxw, boxb):
        R, C, wts, grad_scale = cfg
        ctx.set_materialize_grads(False)            # the statistics' and the deltas' gradients do not exist: no zero maps for them
        lib, st, dev = L.lib(), L.stream(), x.device
        Rp = x.shape[0]
        x2 = x.permute(0, 2, 3, 1).reshape(Rp, -1)         # (h, w, c) columns, the order fc1's weight is stored in: a view of the pooler's output
        if x2.dtype != BF16:
            x2 = x2.to(BF16)
        x2 = x2.contiguous()
        h1 = G.gemm_nt_act(x2, shadow(fc1w), shadow(fc1b), relu=True)
        h2 = G.gemm_nt_act(h1, shadow(fc2w), shadow(fc2b), relu=True)
        y = G.gemm_nt(h2, clsw._dgx16g, clsb._dgx16g)                       # (Rp, ld): logits | deltas | zero columns
        ld = y.shape[1]
        dy = torch.empty_like(y)
        if Rp > R:
            dy[R:].zero_()                                                  # shape-padding RoIs: no loss, no gradient
        dsign = torch.empty(max(R, 1), 4, dtype=torch.float32, device=dev)
        out = torch.empty(16, dtype=torch.float32, device=dev)
        part = torch.empty(max(R, 1) * 8, dtype=torch.float32, device=dev)
        L.check(lib.dgx_detic_losses_strided(y.data_ptr(), ld, y.data_ptr() + 2 * (C + 1), ld, L.ptr(gt_classes), L.ptr(class_w),
                                             L.ptr(prop), L.ptr(gtb), L.ptr(src), R, C, float(wts[0]), float(wts[1]), float(wts[2]),
                                             float(wts[3]), dy.data_ptr(), ld, ld, dsign.data_ptr(), out.data_ptr(), part.data_ptr(),
                                             L.DGX_BF16, st), "dgx_detic_losses_strided")
        ctx.save_for_backward(x2, h1, h2, dy, out)
        ctx.params = (fc1w, fc1b, fc2w, fc2b, clsw, clsb, boxw, boxb)
        ctx.cfg = (R, C, grad_scale, x.shape, x.dtype)
        deltas = y[:R, C + 1:C + 5]
        ctx.mark_non_differentiable(out, deltas)
        return out[8], out[9], out, deltas

    @staticmethod
    def backward(ctx, g_cls, g_box, _o, _d):
        x2, h1, h2, dy, out = ctx.saved_tensors
        fc1w, fc1b, fc2w, fc2b, clsw, clsb, boxw, boxb = ctx.params
        R, C, grad_scale, xshape, xdtype = ctx.cfg
        lib, st = L.lib(), L.stream()
        if ctx.__dict__.get("_used"):
            raise RuntimeError("box_stage: backward called twice on the same forward (its gradient buffer is scaled in place)")
        ctx._used = True
        g_cls = g_cls.float().contiguous() if g_cls is not None else torch.zeros((), device=dy.device)
        g_box = g_box.float().contiguous() if g_box is not None else torch.zeros((), device=dy.device)
        L.check(lib.dgx_detic_grad_scale(dy.data_ptr(), dy.shape[1], R, C, out.data_ptr(), g_cls.data_ptr(), g_box.data_ptr(),
                                         L.DGX_BF16, st), "dgx_detic_grad_scale")
        dz2 = G.gemm_relu_grad(dy, clsw._dgx16tg, h2)               # through the ReLU behind fc2
        dz1 = G.gemm_relu_grad(dz2, shadow_t(fc2w), h1)             # through the ReLU behind fc1
        dx = None
        if ctx.needs_input_grad[0]:
            dx = G.gemm_nt(dz1, shadow_t(fc1w))
            if grad_scale != 1.0:
                dx = dx * grad_scale                                # _ScaleGradient (cascade_rcnn.py:20-28): features only
            dx = dx.view(xshape[0], xshape[2], xshape[3], xshape[1]).permute(0, 3, 1, 2)
            if xdtype != BF16:
                dx = dx.to(xdtype)
        # the stage's one grouped launch is the first (and only) writer of all eight gradient segments in a pass: it overwrites
        # (solver.FlatArena.claim_first_write; 46 M parameters over the three stages that zero_grad then leaves alone)
        slot = getattr(fc1w, "_dgx_arena_slot", None)
        first = slot is not None and all(getattr(q, "_dgx_arena_slot", (None,))[0] is slot[0] for q in ctx.params) \
            and slot[0].claim_first_write(list(ctx.params))
        wgrad_grouped([(fc1w.grad.view(fc1w.shape[0], -1), dz1, x2, fc1b.grad), (fc2w.grad, dz2, h1, fc2b.grad),
                       (clsw._dgxgg, dy, h2, clsb._dgxgg)], beta=0.0 if first else 1.0)
        for p in ctx.params:
            notify_ready(p)
        return (dx,) + (None,) * 14


def box_stage_supported(box_head, predictor):
    """The fused stage covers the shipped recipe: two FCs, sigmoid CE + class-agnostic L1 box regression, every parameter in the
    arena with the predictor pair laid out as one group."""
    fcs = getattr(box_head, "fcs", None)
    if fcs is None or len(fcs) != 2 or not predictor.fused_supported or predictor.only_paste_sup:
        return False
    params = (fcs[0].weight, fcs[0].bias, fcs[1].weight, fcs[1].bias, predictor.cls_score.weight, predictor.cls_score.bias,
              predictor.bbox_pred.weight, predictor.bbox_pred.bias)
    return (arena_resident(params) and getattr(predictor.cls_score.weight, "_dgx16tg", None) is not None
            and getattr(predictor.cls_score.bias, "_dgx16g", None) is not None and torch.is_grad_enabled())


def box_stage(x, box_head, predictor, gt_classes, class_w, prop, gtb, src, R, grad_scale):
    """x (Rp, C, S, S) pooled features (Rp >= R: shape-padding rows) -> (loss_cls, loss_box_reg, out16 statistics, deltas (R, 4))."""
    fcs = box_head.fcs
    C = predictor.num_classes
    cfg = (int(R), int(C), tuple(float(w) for w in predictor.box2box_transform.weights), float(grad_scale))
    with torch.autocast("cuda", enabled=False):
        return _BoxStageFn.apply(x, gt_classes.contiguous(), class_w.float().contiguous() if class_w is not None else None,
                                 prop.float().contiguous(), gtb.float().contiguous(), src.contiguous() if src is not None else None, cfg,
                                 fcs[0].weight, fcs[0].bias, fcs[1].weight, fcs[1].bias, predictor.cls_score.weight,
                                 predictor.cls_score.bias, predictor.bbox_pred.weight, predictor.bbox_pred.bias)
