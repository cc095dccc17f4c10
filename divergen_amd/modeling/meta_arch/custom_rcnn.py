"""CustomRCNN meta-architecture.  Mirrors DG/divergen/modeling/meta_arch/custom_rcnn.py:24-207 over
D2/modeling/meta_arch/rcnn.py:24-243 for the box-supervised path the shipped configs train.

Precision: the reference runs the backbone under fp16 autocast + GradScaler and the heads in fp32
(custom_rcnn.py:141-146).  This build has ONE precision: bf16 operands, fp32 accumulation in every GEMM / convolution /
attention kernel of backbone and heads (no loss scaling needed); losses, box decoding, targets and the optimizer state are
fp32.  cfg.FP16 False (no shipped configuration) is refused: there is no fp32-activation path and no eager fallback."""
import os

import torch
from torch import nn

from .. import META_ARCH_REGISTRY, build_backbone, build_proposal_generator, build_roi_heads
from ...config import configurable
from ...layers.conv_ops import preprocess_patch_rows
from ...layers.roi_ops import FeatureGradients
from ...structures import ImageList

# dgx_preprocess_patches; the composed normalise + pad + unfold form is the reference of its parity test
_FUSED_PREPROCESS = True


class _CaptureGradient(torch.autograd.Function):
    """Identity; its backward hands the incoming gradient to slot i of a FeatureGradients (no copy) and ends the pass there."""

    @staticmethod
    def forward(ctx, fg, i, x):
        ctx.fg, ctx.i = fg, i
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.fg.add(ctx.i, g)
        return None, None, None


class _JoinGradients(torch.autograd.Function):
    """Identity on the feature maps; its backward -- which autograd runs after every consumer of its outputs -- returns what the
    consumers left in the FeatureGradients (the RoI poolers add in place, layers/roi_ops.py; an EARLIER backward pass of the
    proposal generator, `early_proposal_backward`, through _CaptureGradient) plus whatever arrived the ordinary way."""

    @staticmethod
    def forward(ctx, fg, *feats):
        ctx.fg = fg
        ctx.set_materialize_grads(False)
        return tuple(f.view_as(f) for f in feats)

    @staticmethod
    def backward(ctx, *grads):
        out = []
        for i, g in enumerate(grads):
            e = ctx.fg.take(i)
            out.append(g if e is None else e if g is None else e.add_(g))
        return (None,) + tuple(out)


class _PassThrough(torch.autograd.Function):
    """Identity whose backward hands its gradients on unchanged: a place for `torch.autograd.backward(..., inputs=...)` to stop.
    (backward() EXECUTES the node that produced a non-leaf input; stopping at _JoinGradients' own outputs would run its backward --
    and empty the shared maps -- before the last consumer has written.)"""

    @staticmethod
    def forward(ctx, *feats):
        ctx.set_materialize_grads(False)
        return tuple(f.view_as(f) for f in feats)

    @staticmethod
    def backward(ctx, *grads):
        return grads


def _shared_gradient_maps(features):
    """features (dict of maps that require grad) -> (FeatureGradients, the same dict behind _JoinGradients with the slots announced)"""
    names = list(features.keys())
    fg = FeatureGradients(len(names))
    joined = _JoinGradients.apply(fg, *[features[k] for k in names])
    for i, (k, j) in enumerate(zip(names, joined)):
        j._dgx_grad_sink = (fg, i)
        _same_buffer(features[k], j)
    return fg, dict(zip(names, joined))


def _same_buffer(src, view):
    """`view` is `src` behind an identity node: a hipGraph segment that takes `src` as its static input as it stands
    (utils/graphs.py ALIAS_STATIC) may do so with `view`."""
    if getattr(src, "_dgx_static_output", False):
        view._dgx_static_output = True
    return view


class EarlyLosses(dict):
    """The loss dict of a forward that ALREADY back-propagated the proposal generator's part (early_proposal_backward): those
    entries are detached, their gradients sit in the arena / on the feature maps.  Correct only for exactly one backward of the
    plain, unweighted sum of the dict -- which is what engine.total_loss() builds; it marks the dict consumed, and the next
    early forward refuses to run while an unconsumed one is outstanding (a second forward before backward, a loop that sums
    the dict itself, per-loss weights or gradient accumulation would otherwise get wrong gradients without a word)."""
    consumed = False


@META_ARCH_REGISTRY.register()
class CustomRCNN(nn.Module):
    @configurable
    def __init__(self, *, backbone, proposal_generator, roi_heads, pixel_mean, pixel_std, input_format=None,
                 vis_period=0, fp16=False, with_image_labels=False, roi_head_name="", **unused):
        super().__init__()
        assert proposal_generator is not None
        if with_image_labels:
            raise NotImplementedError("WITH_IMAGE_LABELS co-training is outside the shipped configs")
        self.backbone, self.proposal_generator, self.roi_heads = backbone, proposal_generator, roi_heads
        self.input_format, self.vis_period, self.fp16, self.roi_head_name = input_format, vis_period, fp16, roi_head_name
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std).view(-1, 1, 1), False)
        if not fp16:
            raise NotImplementedError("cfg.FP16 False: this build computes in bf16 with fp32 accumulation on its own HIP kernels; "
                                      "an fp32-activation mode does not exist (every shipped configuration sets FP16: True)")
        # hipGraph capture of the static-shape backbone fwd+bwd (launch-bound otherwise: ~3.5k launches)
        self.return_proposal = False
        # Trainer opt-in (bench.py, train_net.py's plain loop): the proposal generator's losses are back-propagated from INSIDE the
        # forward, queued on the GPU before the RoI heads' one device->host read (the proposal sampler), so the device has that
        # backward to run while the host -- which loses its whole lead at that read -- issues the RoI heads.  The loss dict then carries
        # those losses DETACHED (their gradients are already in the arena / on the feature maps): only valid for a loop that calls
        # backward once on the plain sum of the dict, after a zero_grad.
        self.early_proposal_backward = False
        # The box cascade's losses likewise, right behind the cascade's forward and up to the feature maps: measured 0.3 ms/step SLOWER
        # (24.8-25.1 against 24.4-24.7 ms in one box: one more engine start costs more than the stretch it covers), so it stays off.
        self.early_box_backward = False

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        shape = backbone.output_shape()
        return dict(backbone=backbone, proposal_generator=build_proposal_generator(cfg, shape),
                    roi_heads=build_roi_heads(cfg, shape), input_format=cfg.INPUT.FORMAT, vis_period=cfg.VIS_PERIOD,
                    pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, fp16=cfg.FP16,
                    with_image_labels=cfg.WITH_IMAGE_LABELS, roi_head_name=cfg.MODEL.ROI_HEADS.NAME)

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        """rcnn.py:220-227: H2D, (x - mean) / std, zero-pad to the backbone's divisibility."""
        images = [x["image"].to(self.device, non_blocking=True) for x in batched_inputs]
        if self._patch_rows_ok(images):
            # uint8 images -> PatchEmbed's GEMM operand in one pass (normalise + zero-pad + 4x4 unfold, bf16); the fp32 batch
            # tensor is only materialised if somebody asks for `.tensor`
            pr, sizes = preprocess_patch_rows(images, self.pixel_mean, self.pixel_std, self.backbone.size_divisibility)
            return ImageList(None, sizes, patch_rows=pr)
        images = [(x.float() - self.pixel_mean) / self.pixel_std for x in images]
        return ImageList.from_tensors(images, self.backbone.size_divisibility)

    def _patch_rows_ok(self, images):
        bu = getattr(self.backbone, "bottom_up", None)
        pe = getattr(bu, "patch_embed", None)
        return (_FUSED_PREPROCESS and pe is not None and pe.patch_size == (4, 4)
                and all(im.is_cuda and im.dtype == torch.uint8 and im.dim() == 3 and im.shape[0] == 3 for im in images))

    def _features(self, images):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if images.patch_rows is not None:
                return self.backbone(images.patch_rows)
            return self.backbone(images.tensor.to(memory_format=torch.channels_last))

    def forward(self, batched_inputs):
        if not self.training:
            return self.inference(batched_inputs)
        # BSGAL (BS/bsgal/modeling/meta_arch/custom_rcnn.py:278-780): with INPUT.ACTIVE_SELECT the loader hands the pasted
        # sample together with its un-pasted original and a held-out image; engine.bsgal.ActiveSelector (attached by the
        # trainer once the parameter arena exists) decides which of the two batches this step trains on
        sel = self.__dict__.get("active_selector")
        if sel is not None and len(batched_inputs) and "origin_image" in batched_inputs[0]:
            batched_inputs, paste = sel.select(batched_inputs)
            extra = sel.extra_losses()                                              # ACTIVE_COMPARE 'all': the original batch as well
            losses = self.training_losses(batched_inputs)
            losses = {k: v for k, v in losses.items() if "paste" not in k}          # :767 pop_loss_paste
            if extra is not None:                                                   # :772-774
                losses = {k: v + extra[k] for k, v in losses.items()}
            if sel.mode == "paste_or_zero" and not paste:                           # :769-771: the step trains on nothing
                losses = {k: v * 0.0 for k, v in losses.items()}
            return losses
        return self.training_losses(batched_inputs, early=self.early_proposal_backward and not self.return_proposal)

    def training_losses(self, batched_inputs, only_gt_proposals=False, early=False):
        """custom_rcnn.py:118-207 for the box-supervised path: the loss dict of one batch."""
        images = self.preprocess_image(batched_inputs)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        features = self._features(images)
        from ...solver import join_transposes
        join_transposes()        # transposed weight images refreshed beside the backbone forward: first read by what follows
        grads_on = torch.is_grad_enabled() and all(f.requires_grad for f in features.values())
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if grads_on:
                # the consumers of the FPN levels write ONE gradient map per level between them
                fg, joined = _shared_gradient_maps(features)
            if hasattr(self.roi_heads, "prepare_targets") and not only_gt_proposals:
                self.roi_heads.prepare_targets(gt_instances)
            if early and grads_on:
                prev = self.__dict__.get("_early_outstanding")
                if prev is not None and not prev.consumed:
                    raise RuntimeError("early_proposal_backward: the previous forward's loss dict was never passed to engine.total_loss() "
                                       "(one backward of the plain sum per forward is the contract; switch model.early_proposal_backward "
                                       "off for loops that weight losses, accumulate gradients or run two forwards per backward)")
                stubs = {k: _same_buffer(f, _CaptureGradient.apply(fg, i, f.detach().requires_grad_(True)))
                         for i, (k, f) in enumerate(features.items())}
                proposals, proposal_losses = self.proposal_generator(images, stubs, gt_instances)
                pl_total = torch.stack([v.float().reshape(()) for v in proposal_losses.values()]).sum()
                self.roi_heads.__dict__["_before_host_read"] = pl_total.backward
                if self.early_box_backward and not only_gt_proposals and all(getattr(j, "_dgx_grad_sink", None) is not None for j in joined.values()):
                    # ... and the box cascade's losses right behind its forward, up to the feature maps the RoI heads see (roi_heads.forward)
                    names = list(joined.keys())
                    stops = _PassThrough.apply(*[joined[k] for k in names])
                    for k, t in zip(names, stops):
                        t._dgx_grad_sink = joined[k]._dgx_grad_sink
                        _same_buffer(joined[k], t)
                    joined = dict(zip(names, stops))
                    self.roi_heads.__dict__["_early_box_backward"] = list(stops)
                try:
                    proposals, detector_losses = self.roi_heads(images, joined, proposals, gt_instances, ann_type="box",
                                                                only_gt_proposals=only_gt_proposals)
                finally:
                    pending = self.roi_heads.__dict__.pop("_before_host_read", None)
                    self.roi_heads.__dict__.pop("_early_box_backward", None)
                if pending is not None:         # the RoI heads took a path without the fused sampler
                    pending()
                proposal_losses = {k: v.detach() for k, v in proposal_losses.items()}
            else:
                if grads_on:
                    features = joined
                proposals, proposal_losses = self.proposal_generator(images, features, gt_instances)
                proposals, detector_losses = self.roi_heads(images, features, proposals, gt_instances, ann_type="box",
                                                            only_gt_proposals=only_gt_proposals)
        losses = EarlyLosses() if (early and grads_on) else {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        if isinstance(losses, EarlyLosses):
            self.__dict__["_early_outstanding"] = losses
        return (proposals, losses) if self.return_proposal else losses

    @torch.no_grad()
    def inference(self, batched_inputs, do_postprocess=True):
        assert not self.training
        images = self.preprocess_image(batched_inputs)
        features = self._features(images)
        from ...solver import join_transposes
        join_transposes()        # (the mask head's deconvolution reads a transposed weight image)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            proposals, _ = self.proposal_generator(images, features, None)
            results, _ = self.roi_heads(images, features, proposals)
        if not do_postprocess:
            return results
        out = []
        for r, inp, size in zip(results, batched_inputs, images.image_sizes):
            out.append({"instances": detector_postprocess(r, inp.get("height", size[0]), inp.get("width", size[1]))})
        return out


def detector_postprocess(results, output_height, output_width, mask_threshold=0.5, mask_format="bitmask"):
    """D2/modeling/postprocessing.py: rescale boxes to the output resolution, paste masks.
    mask_format="rle" (not in the reference): leave the masks as `pred_masks_rle` COCO run-length dicts, encoded on the GPU
    straight from the SxS probabilities -- what the evaluator's results writer turns the bitmasks into anyway."""
    from ...structures import Boxes, Instances
    sx, sy = output_width / results.image_size[1], output_height / results.image_size[0]
    out = Instances((output_height, output_width), **results.get_fields())
    b = out.pred_boxes.tensor.clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    bx = Boxes(b)
    bx.clip(out.image_size)
    out.pred_boxes = bx
    keep = bx.nonempty()
    out = out[keep]
    if out.has("pred_masks") and mask_format == "rle":
        from ...layers.mask_ops import paste_masks_rle
        rles = paste_masks_rle(out.pred_masks[:, 0], out.pred_boxes.tensor, (output_height, output_width), mask_threshold)
        out.remove("pred_masks")
        out.pred_masks_rle = rles
    elif out.has("pred_masks"):
        out.pred_masks = paste_masks_in_image(out.pred_masks[:, 0], out.pred_boxes.tensor, (output_height, output_width),
                                              mask_threshold)
    return out


def paste_masks_in_image(masks, boxes, image_shape, threshold=0.5):
    """D2/layers/mask_ops.py:73 -- aligned bilinear paste of the SxS probability maps, on the GPU (dgx_paste_masks)."""
    from ...layers.mask_ops import paste_masks_in_image as _paste
    return _paste(masks, boxes, image_shape, threshold)
