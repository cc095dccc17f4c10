"""DeticCascadeROIHeads: 3-stage cascade box head + class-agnostic mask head with instance_source
plumbing.  Mirrors DG/divergen/modeling/roi_heads/detic_roi_heads.py:29-414 over
D2/modeling/roi_heads/{cascade_rcnn.py:20-299, roi_heads.py:46-76,181-300}.

MI355X-first: IoU + Matcher is one kernel per image with no (M x N) matrix; all RoI pooling is the
multi-level NHWC HIP ROIAlign; statistics stay on the device."""
import math
import os

import torch
from torch import nn

from .. import ROI_HEADS_REGISTRY, ShapeSpec
from ...config import configurable
from ...layers import iou_match
from ...layers.box_stage import box_stage, box_stage_supported
from ...structures import Boxes, Instances, ProposalBatch
from ...utils.events import DeferredScalar, get_event_storage
from ..box_regression import Box2BoxTransform
from .box_head import build_box_head
from .detic_fast_rcnn import DeticFastRCNNOutputLayers, fast_rcnn_inference
from .mask_head import build_mask_head
from .poolers import ROIPooler


class _ScaleGradient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


def _subsample_labels(labels, num_samples, positive_fraction, bg_label, pre=None):
    """D2/modeling/sampling.py:9-54 (two torch.randperm draws on the labels' device).  pre = (positive, negative) index lists
    computed by the caller for the whole batch behind ONE device->host read (label_and_sample_proposals); the draws and their
    order are the reference's either way."""
    if pre is not None:
        positive, negative = pre
    else:
        positive = ((labels != -1) & (labels != bg_label)).nonzero().squeeze(1)
        negative = (labels == bg_label).nonzero().squeeze(1)
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    p1 = torch.randperm(positive.numel(), device=positive.device)[:num_pos]
    p2 = torch.randperm(negative.numel(), device=negative.device)[:num_neg]
    return positive[p1], negative[p2]


subsample_labels = _subsample_labels     # tests substitute a deterministic rule here


def draw_permutation(n, k, device):
    """The first k entries of a random permutation of n (sampling.py:42-47: `torch.randperm(n)[:k]`, torch's generator).  The
    batch-level sampler below calls it in the reference's order (image by image, positives then negatives); tests substitute
    `arange(k)` here (= "the first k in index order", the rule the CPU oracle applies)."""
    return torch.randperm(n, device=device)[:k]


def add_ground_truth_to_proposals(gt, proposals):
    """D2/modeling/proposal_generator/proposal_utils.py:126-196."""
    out = []
    logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
    for g, p in zip(gt, proposals):
        gp = Instances(p.image_size, **g.get_fields())
        gp.proposal_boxes = g.gt_boxes
        gp.objectness_logits = logit * torch.ones(len(g), device=p.objectness_logits.device)
        if p.has("proposal_valid"):     # fixed-length proposal lists (training): ground-truth rows are always valid
            gp.proposal_valid = torch.ones(len(g), dtype=torch.bool, device=p.objectness_logits.device)
        for key in p.get_fields().keys():
            assert gp.has(key), "The attribute '{}' in `proposals` does not exist in `gt`".format(key)
        sel = Instances(p.image_size, proposal_boxes=gp.proposal_boxes, objectness_logits=gp.objectness_logits)
        if p.has("proposal_valid"):
            sel.proposal_valid = gp.proposal_valid
        out.append(Instances.cat([p, sel]))
    return out


def select_foreground_proposals(proposals, bg_label):
    """D2/modeling/roi_heads/roi_heads.py:46-76.  Proposals that come out of label_and_sample_proposals carry their foreground
    rows FIRST and the count on the host (`_dgx_num_fg`): the selection is then a slice, not a `nonzero` (no device->host read)."""
    fg, masks = [], []
    for p in proposals:
        n = p.__dict__.get("_dgx_num_fg")
        if n is not None:
            fg.append(p[:n])
            masks.append(None)               # (the selection mask is only computed where somebody needs it)
            continue
        m = (p.gt_classes != -1) & (p.gt_classes != bg_label)
        fg.append(p[m.nonzero().squeeze(1)])
        masks.append(m)
    return fg, masks


@ROI_HEADS_REGISTRY.register()
class DeticCascadeROIHeads(nn.Module):
    @configurable
    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_append_gt, box_in_features,
                 box_pooler, box_heads, box_predictors, cascade_ious, mask_in_features=None, mask_pooler=None,
                 mask_head=None, mult_proposal_score=False, mask_weight=1.0, divergen_mask_loss=True,
                 one_class_per_proposal=False, **unused):
        super().__init__()
        self.num_classes, self.batch_size_per_image, self.positive_fraction = num_classes, batch_size_per_image, positive_fraction
        self.proposal_append_gt = proposal_append_gt
        self.box_in_features, self.box_pooler = box_in_features, box_pooler
        self.box_head, self.box_predictor = nn.ModuleList(box_heads), nn.ModuleList(box_predictors)
        self.num_cascade_stages, self.cascade_ious = len(box_heads), list(cascade_ious)
        self.mask_on = mask_head is not None
        if self.mask_on:
            self.mask_in_features, self.mask_pooler, self.mask_head = mask_in_features, mask_pooler, mask_head
        self.mult_proposal_score, self.mask_weight = mult_proposal_score, mask_weight
        self.divergen_mask_loss, self.one_class_per_proposal = divergen_mask_loss, one_class_per_proposal

    @classmethod
    def from_config(cls, cfg, input_shape):
        rh, bh, ch = cfg.MODEL.ROI_HEADS, cfg.MODEL.ROI_BOX_HEAD, cfg.MODEL.ROI_BOX_CASCADE_HEAD
        in_features = rh.IN_FEATURES
        scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        in_ch = [input_shape[f].channels for f in in_features]
        assert len(set(in_ch)) == 1 and bh.CLS_AGNOSTIC_BBOX_REG and ch.IOUS[0] == rh.IOU_THRESHOLDS[0]
        assert len(ch.BBOX_REG_WEIGHTS) == len(ch.IOUS)
        pooled = ShapeSpec(channels=in_ch[0], width=bh.POOLER_RESOLUTION, height=bh.POOLER_RESOLUTION)
        heads, preds = [], []
        for w in ch.BBOX_REG_WEIGHTS:
            h = build_box_head(cfg, pooled)
            heads.append(h)
            preds.append(DeticFastRCNNOutputLayers(cfg, h.output_shape, box2box_transform=Box2BoxTransform(weights=w)))
        ret = dict(num_classes=rh.NUM_CLASSES, batch_size_per_image=rh.BATCH_SIZE_PER_IMAGE,
                   positive_fraction=rh.POSITIVE_FRACTION, proposal_append_gt=rh.PROPOSAL_APPEND_GT,
                   box_in_features=in_features,
                   # channels-last pooled rows: the box heads' first FC keeps its columns in (h, w, c) order (box_head.py)
                   box_pooler=ROIPooler(bh.POOLER_RESOLUTION, scales, bh.POOLER_SAMPLING_RATIO, bh.POOLER_TYPE, out_nhwc=True),
                   box_heads=heads, box_predictors=preds, cascade_ious=ch.IOUS,
                   mult_proposal_score=bh.MULT_PROPOSAL_SCORE, mask_weight=rh.MASK_WEIGHT,
                   divergen_mask_loss=cfg.MODEL.USE_DIVERGEN_MASK_LOSS and cfg.MODEL.get("USE_XPASTE_MASK_LOSS", True), one_class_per_proposal=rh.ONE_CLASS_PER_PROPOSAL)
        if cfg.MODEL.MASK_ON:
            mh = cfg.MODEL.ROI_MASK_HEAD
            ret.update(mask_in_features=in_features,
                       mask_pooler=ROIPooler(mh.POOLER_RESOLUTION, scales, mh.POOLER_SAMPLING_RATIO, mh.POOLER_TYPE, out_nhwc=True),
                       mask_head=build_mask_head(cfg, ShapeSpec(channels=in_ch[0], width=mh.POOLER_RESOLUTION, height=mh.POOLER_RESOLUTION)))
        return ret

    # ------------------------------------------------------------ sampling / matching
    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, targets, only_gt_proposals=False):
        """only_gt_proposals (BS/bsgal/modeling/roi_heads/detic_roi_heads.py:334-358, BSGAL's held-out pass): an image WITH
        ground truth keeps exactly its ground-truth boxes as proposals (the rows appended last), labelled with their own
        classes; an image without any is sampled as usual."""
        fused = self._label_and_sample_fused(proposals, targets) if not only_gt_proposals else None
        if fused is not None:
            return fused
        if self.proposal_append_gt:
            proposals = add_ground_truth_to_proposals(targets, proposals)
        # pass 1 (device only): labels of every image and, for the reference's sampler, its positive / negative index lists as
        # stable argsorts of the two masks; the four list LENGTHS of the batch are read back together -- one device->host read
        # per step here instead of one `nonzero` per list (the lengths are what torch.randperm(n) needs on the host)
        labels, pre = [], [None] * len(proposals)
        for p, t in zip(proposals, targets):
            midx, mlab = iou_match(t.gt_boxes.tensor, p.proposal_boxes.tensor, self.cascade_ious[0])
            if len(t) > 0:
                gtc = t.gt_classes[midx]
                gtc[mlab == 0] = self.num_classes
            else:
                gtc = torch.zeros_like(midx) + self.num_classes
            if p.has("proposal_valid"):
                # padding rows of a fixed-length proposal list: label -1 = "ignore", never sampled
                gtc = torch.where(p.proposal_valid, gtc, torch.full_like(gtc, -1))
                p.remove("proposal_valid")
            labels.append((midx, gtc))
        if subsample_labels is _subsample_labels and labels and labels[0][1].is_cuda:
            orders, counts = [], []
            for _, gtc in labels:
                pm, nm = (gtc != -1) & (gtc != self.num_classes), gtc == self.num_classes
                orders.append((torch.sort(pm.to(torch.int8), descending=True, stable=True)[1],
                               torch.sort(nm.to(torch.int8), descending=True, stable=True)[1]))
                counts += [pm.sum(), nm.sum()]
            counts = torch.stack(counts).tolist()
            pre = [(o[0][:counts[2 * i]], o[1][:counts[2 * i + 1]]) for i, o in enumerate(orders)]
        out, nfg, nbg = [], [], []
        for i, (p, t) in enumerate(zip(proposals, targets)):
            has_gt = len(t) > 0
            if only_gt_proposals and has_gt:
                assert self.proposal_append_gt
                p = p[len(p) - len(t):]
                if p.has("proposal_valid"):
                    p.remove("proposal_valid")
                p.gt_classes = t.gt_classes
                for name, val in t.get_fields().items():
                    if (name.startswith("gt_") or name == "instance_source") and not p.has(name):
                        p.set(name, val)
                nfg.append(torch.tensor(float(len(t)), device=t.gt_classes.device))
                nbg.append(torch.zeros((), device=t.gt_classes.device))
                p.__dict__["_dgx_num_fg"] = len(t)
                out.append(p)
                continue
            midx, gtc = labels[i]
            if pre[i] is not None:
                fg_idx, bg_idx = subsample_labels(gtc, self.batch_size_per_image, self.positive_fraction, self.num_classes, pre=pre[i])
            else:
                fg_idx, bg_idx = subsample_labels(gtc, self.batch_size_per_image, self.positive_fraction, self.num_classes)
            sidx = torch.cat([fg_idx, bg_idx], dim=0)
            p = p[sidx]
            p.gt_classes = gtc[sidx]
            p.__dict__["_dgx_num_fg"] = int(fg_idx.numel())       # foreground rows first: select_foreground_proposals slices
            if has_gt:
                st = midx[sidx]
                for name, val in t.get_fields().items():
                    if (name.startswith("gt_") or name == "instance_source") and not p.has(name):
                        p.set(name, val[st])
            nbg.append((p.gt_classes == self.num_classes).sum())
            nfg.append(p.gt_classes.numel() - nbg[-1])
            out.append(p)
        st = get_event_storage()
        st.put_scalar("roi_head/num_fg_samples", torch.stack([x.float() for x in nfg]).mean())
        st.put_scalar("roi_head/num_bg_samples", torch.stack([x.float() for x in nbg]).mean())
        return out

    @staticmethod
    def _cat_targets(targets, dev, G, has_src, offs):
        from ...utils.h2d import upload_i32
        if G:
            gt_boxes = torch.cat([t.gt_boxes.tensor for t in targets]).float().contiguous()
            gt_classes = torch.cat([t.gt_classes for t in targets]).contiguous()
            gt_src = torch.cat([t.instance_source for t in targets]).contiguous() if has_src else None
        else:
            gt_boxes, gt_classes, gt_src = torch.zeros(1, 4, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), None
        return gt_boxes, gt_classes, gt_src, upload_i32(offs, dev)

    @torch.no_grad()
    def prepare_targets(self, targets):
        """The batch form of the ground truth the fused sampler reads (boxes, classes, sources of all images concatenated + the image
        offsets), built by the meta-architecture BEFORE the proposal generator runs: off the path between the proposal decode and the
        sampler's device->host read."""
        if not self.training or not len(targets) or not all(t.gt_boxes.tensor.is_cuda for t in targets):
            return
        offs = [0]
        for t in targets:
            offs.append(offs[-1] + len(t))
        has_src = all(t.has("instance_source") for t in targets)
        self.__dict__["_gt_batch"] = (targets,) + DeticCascadeROIHeads._cat_targets(targets, targets[0].gt_boxes.tensor.device, offs[-1], has_src, offs)

    def _label_and_sample_fused(self, proposals, targets):
        """label_and_sample_proposals for the whole batch in two launches around the step's one device->host read
        (dgx_roi_label -> counts -> torch.randperm draws in the reference's order -> dgx_roi_gather); None when the inputs do
        not come as a batch (no fixed-length proposal tensors, CPU tensors, a substituted `subsample_labels`)."""
        import ctypes
        from ... import _lib as L
        from ...structures import BitMasks
        from ...utils.h2d import upload_i32
        batch = getattr(proposals, "batch", None)
        if (batch is None or subsample_labels is not _subsample_labels or not self.proposal_append_gt or not len(targets)
                or not batch[0].is_cuda or len(targets) > 16):
            return None
        boxes, scores, valid = batch                                  # (B, K, 4) f32, (B, K) f32, (B, K) bool
        B, K = int(boxes.shape[0]), int(boxes.shape[1])
        dev = boxes.device
        gts = [len(t) for t in targets]
        offs = [0]
        for n in gts:
            offs.append(offs[-1] + n)
        G = offs[-1]
        has_src = all(t.has("instance_source") for t in targets)
        pre = self.__dict__.pop("_gt_batch", None)
        if pre is not None and pre[0] is targets:
            # concatenated ahead of the proposal generator (prepare_targets): these launches sat between the decode and the step's
            # device->host read otherwise
            _, gt_boxes, gt_classes, gt_src, offs_t = pre
        else:
            gt_boxes, gt_classes, gt_src, offs_t = DeticCascadeROIHeads._cat_targets(targets, dev, G, has_src, offs)
        Nmax = K + max(gts)
        midx = torch.empty(B, Nmax, dtype=torch.int32, device=dev)
        labels = torch.empty(B, Nmax, dtype=torch.int64, device=dev)
        pos_idx = torch.empty(B, Nmax, dtype=torch.int32, device=dev)
        neg_idx = torch.empty(B, Nmax, dtype=torch.int32, device=dev)
        counts = torch.empty(2 * B, dtype=torch.int32, device=dev)
        boxes, scores = boxes.float().contiguous(), scores.float().contiguous()
        valid_u8 = valid.view(torch.uint8).contiguous() if valid is not None else None
        lib = L.lib()
        L.check(lib.dgx_roi_label(L.ptr(boxes), L.ptr(valid_u8), B, K, L.ptr(gt_boxes), L.ptr(gt_classes), L.ptr(offs_t), max(gts),
                                  float(self.cascade_ious[0]), self.num_classes, 1, Nmax, L.ptr(midx), L.ptr(labels), L.ptr(pos_idx),
                                  L.ptr(neg_idx), L.ptr(counts), L.stream()), "dgx_roi_label")
        before = self.__dict__.pop("_before_host_read", None)
        if before is not None:
            # (meta_arch/custom_rcnn.py early_proposal_backward) the counts start their way to the host NOW; the work queued by
            # `before` runs behind them, so the device is busy while the host reads and issues the rest of the heads
            host = self.__dict__.get("_counts_host")
            if host is None or host.numel() < 2 * B:
                host = self.__dict__["_counts_host"] = torch.empty(max(2 * B, 32), dtype=torch.int32).pin_memory()
                self.__dict__["_counts_event"] = torch.cuda.Event()
            host[:2 * B].copy_(counts, non_blocking=True)
            ev = self.__dict__["_counts_event"]
            ev.record()
            before()
            ev.synchronize()
            c = host[:2 * B].tolist()
        else:
            c = counts.tolist()                                       # THE device->host read of the step
        perms, npos, nneg = [], [], []
        for i in range(B):
            k_pos = min(c[2 * i], int(self.batch_size_per_image * self.positive_fraction))
            k_neg = min(c[2 * i + 1], self.batch_size_per_image - k_pos)
            perms.append((draw_permutation(c[2 * i], k_pos, dev), draw_permutation(c[2 * i + 1], k_neg, dev)))
            npos.append(k_pos)
            nneg.append(k_neg)
        R = sum(npos) + sum(nneg)
        o_box = torch.empty(R, 4, dtype=torch.float32, device=dev)
        o_cls = torch.empty(R, dtype=torch.int64, device=dev)
        o_gtb = torch.empty(R, 4, dtype=torch.float32, device=dev)
        o_gti = torch.empty(R, dtype=torch.int64, device=dev)
        o_src = torch.empty(R, dtype=torch.int64, device=dev) if has_src else None
        o_log = torch.empty(R, dtype=torch.float32, device=dev)
        pp = (ctypes.c_void_p * B)(*[p1.data_ptr() if p1.numel() else None for p1, _ in perms])
        pn = (ctypes.c_void_p * B)(*[p2.data_ptr() if p2.numel() else None for _, p2 in perms])
        gt_logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
        L.check(lib.dgx_roi_gather(B, K, Nmax, pp, pn, (ctypes.c_int * B)(*npos), (ctypes.c_int * B)(*nneg), L.ptr(boxes), L.ptr(scores),
                                   L.ptr(gt_boxes), L.ptr(gt_src), L.ptr(offs_t), gt_logit, L.ptr(midx), L.ptr(labels), L.ptr(pos_idx),
                                   L.ptr(neg_idx), L.ptr(o_box), L.ptr(o_cls), L.ptr(o_gtb), L.ptr(o_gti), L.ptr(o_src), L.ptr(o_log),
                                   L.stream()), "dgx_roi_gather")
        out, r0 = ProposalBatch(), 0
        for i, (p, t) in enumerate(zip(proposals, targets)):
            n = npos[i] + nneg[i]
            inst = Instances(p.image_size, proposal_boxes=Boxes(o_box[r0:r0 + n]), objectness_logits=o_log[r0:r0 + n],
                             gt_classes=o_cls[r0:r0 + n])
            if gts[i] > 0:
                inst.gt_boxes = Boxes(o_gtb[r0:r0 + n])
                if has_src:
                    inst.instance_source = o_src[r0:r0 + n]
                if t.has("gt_masks"):
                    inst.gt_masks = t.gt_masks[o_gti[r0:r0 + n]]      # lazy: an index into the image's mask stack
                for name, val in t.get_fields().items():
                    if name.startswith("gt_") and name not in ("gt_boxes", "gt_masks", "gt_classes") and not inst.has(name):
                        inst.set(name, val[o_gti[r0:r0 + n]])
            inst.__dict__["_dgx_num_fg"] = npos[i]
            out.append(inst)
            r0 += n
        out.train = dict(prop=o_box, gt_classes=o_cls, gt_boxes=o_gtb, src=o_src, counts=[a + b for a, b in zip(npos, nneg)],
                         t_boxes=gt_boxes, t_classes=gt_classes, t_src=gt_src, gts=gts)
        st = get_event_storage()
        st.put_scalar("roi_head/num_fg_samples", sum(npos) / float(B))
        st.put_scalar("roi_head/num_bg_samples", sum(nneg) / float(B))
        return out

    @torch.no_grad()
    def _match_and_label_boxes(self, proposals, stage, targets):
        nfg, nbg = [], []
        for p, t in zip(proposals, targets):
            midx, mlab = iou_match(t.gt_boxes.tensor, p.proposal_boxes.tensor, self.cascade_ious[stage])
            src = None
            if len(t) > 0:
                gtc = t.gt_classes[midx]
                gtc[mlab == 0] = self.num_classes
                if t.has("instance_source"):
                    src = t.instance_source[midx]
                    src[mlab == 0] = 0
                gtb = t.gt_boxes[midx]
            else:
                gtc = torch.zeros_like(midx) + self.num_classes
                gtb = Boxes(t.gt_boxes.tensor.new_zeros((len(p), 4)))
                if t.has("instance_source"):
                    src = torch.zeros_like(midx)
            p.gt_classes, p.gt_boxes = gtc, gtb
            if src is not None:
                p.instance_source = src
            f = (mlab == 1).sum()
            nfg.append(f.float())
            nbg.append(mlab.numel() - f.float())
        st = get_event_storage()
        st.put_scalar("stage{}/roi_head/num_fg_samples".format(stage), torch.stack(nfg).mean())
        st.put_scalar("stage{}/roi_head/num_bg_samples".format(stage), torch.stack(nbg).mean())
        return proposals

    def _create_proposals_from_boxes(self, boxes, image_sizes, logits):
        out = []
        for b, size, logit in zip(boxes, image_sizes, logits):
            b = Boxes(b.detach())
            b.clip(size)
            if self.training:
                inds = b.nonempty()
                b, logit = b[inds], logit[inds]
            out.append(Instances(size, proposal_boxes=b, objectness_logits=logit))
        return out

    def _run_stage(self, features, proposals, stage):
        R = sum(len(p) for p in proposals)
        x = self.box_pooler(features, [p.proposal_boxes for p in proposals], pad_to=256 if self.training else 0)
        x = _ScaleGradient.apply(x, 1.0 / self.num_cascade_stages)
        scores, deltas = self.box_predictor[stage](self.box_head[stage](x))
        return scores[:R], deltas[:R]

    def _forward_box_train(self, features, proposals, targets):
        """Training cascade with the stage hand-over (decode, clip, re-match, gather: `_create_proposals_from_boxes` +
        `_match_and_label_boxes` + `predict_boxes`) as ONE kernel per stage over the whole batch (dgx_cascade_refine) and
        the losses of each stage as one kernel pair (dgx_detic_losses).  An empty refined box -- a row the reference
        drops -- becomes an "ignore" row instead (label -1), so no shape depends on the data."""
        import ctypes
        from ... import _lib as L
        dev = proposals[0].proposal_boxes.tensor.device
        B = len(proposals)
        tr = getattr(proposals, "train", None)
        counts = tr["counts"] if tr is not None else [len(p) for p in proposals]
        gts = tr["gts"] if tr is not None else [len(t) for t in targets]
        row0 = (ctypes.c_int * (B + 1))(*([0] + [sum(counts[:i + 1]) for i in range(B)]))
        gt0 = (ctypes.c_int * (B + 1))(*([0] + [sum(gts[:i + 1]) for i in range(B)]))
        img_h = (ctypes.c_float * B)(*[float(p.image_size[0]) for p in proposals])
        img_w = (ctypes.c_float * B)(*[float(p.image_size[1]) for p in proposals])
        R = sum(counts)
        has_src = all(t.has("instance_source") for t in targets)
        if tr is not None:      # the sampler's batch-level tensors: no re-concatenation
            gt_boxes, gt_classes, gt_src = tr["t_boxes"], tr["t_classes"], tr["t_src"]
            prop, gtc, gtb, src = tr["prop"], tr["gt_classes"], tr["gt_boxes"], tr["src"]
        else:
            gt_boxes = torch.cat([t.gt_boxes.tensor for t in targets]).float().contiguous()
            gt_classes = torch.cat([t.gt_classes for t in targets]).contiguous()
            gt_src = torch.cat([t.instance_source for t in targets]).contiguous() if has_src else None
            prop = torch.cat([p.proposal_boxes.tensor for p in proposals]).float().contiguous()
            gtc = torch.cat([p.gt_classes for p in proposals])
            gtb = torch.cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals]).float()
            src = torch.cat([p.instance_source for p in proposals]) if all(p.has("instance_source") for p in proposals) else None
        feats = [features[f] for f in self.box_in_features]
        st = get_event_storage()
        losses, valid, deltas = {}, None, None
        wts = self.box_predictor[0].box2box_transform
        for k in range(self.num_cascade_stages):
            if k > 0:
                tr = self.box_predictor[k - 1].box2box_transform
                d = deltas.detach().contiguous()
                nb = torch.empty(R, 4, dtype=torch.float32, device=dev)
                nvalid = torch.empty(R, dtype=torch.uint8, device=dev)
                gtc = torch.empty(R, dtype=torch.int64, device=dev)
                gtb = torch.empty(R, 4, dtype=torch.float32, device=dev)
                src = torch.empty(R, dtype=torch.int64, device=dev) if has_src else None
                nfg = torch.empty(1, dtype=torch.int32, device=dev)
                L.check(L.lib().dgx_cascade_refine(L.ptr(prop), L.ptr(d), L.ptr(valid), B, row0, gt0, img_h, img_w, L.ptr(gt_boxes),
                                                   L.ptr(gt_classes), L.ptr(gt_src), float(self.cascade_ious[k]), self.num_classes,
                                                   float(tr.weights[0]), float(tr.weights[1]), float(tr.weights[2]), float(tr.weights[3]),
                                                   float(tr.scale_clamp), L.ptr(nb), L.ptr(nvalid), L.ptr(gtc), L.ptr(gtb), L.ptr(src),
                                                   L.ptr(nfg), L.dtype_code(d), L.stream()), "dgx_cascade_refine")
                prop, valid = nb, nvalid
                st.put_scalar("stage{}/roi_head/num_fg_samples".format(k), DeferredScalar(lambda f, B=B: f[0] / B, nfg))
                st.put_scalar("stage{}/roi_head/num_bg_samples".format(k), DeferredScalar(lambda f, B=B, R=R: (R - f[0]) / B, nfg))
            obs = self.__dict__.get("stage_observer")
            if obs is not None:      # tests: the labels this stage trains on (hand-over to the CPU oracle)
                obs(k, dict(boxes=prop, valid=valid, gt_classes=gtc, gt_boxes=gtb, counts=counts))
            pred = self.box_predictor[k]
            fused = box_stage_supported(self.box_head[k], pred)
            # _ScaleGradient (cascade_rcnn.py:20-28, :150) sits on the pooled features: the fused stage leaves the factor to the pooler's
            # backward, which folds it into its interpolation table (no pass over the 25 MB pooled gradient)
            x = self.box_pooler.forward_rows(feats, prop, counts, pad_to=256, grad_scale=1.0 / self.num_cascade_stages if fused else 1.0)
            if fused:
                # flatten -> fc1 -> ReLU -> fc2 -> ReLU -> cls_score | bbox_pred -> losses: one autograd node (layers/box_stage.py)
                C = pred.num_classes
                w = pred._class_weight(gtc, C)
                loss_cls, loss_box, out, deltas = box_stage(x, self.box_head[k], pred, gtc, w, prop, gtb,
                                                            None if pred.divergen_box_loss else src, R, 1.0)
                with st.name_scope("stage{}".format(k)):
                    st.put_scalar("fast_rcnn/cls_accuracy", out[11])
                    st.put_scalar("fast_rcnn/fg_cls_accuracy", out[12])
                    st.put_scalar("fast_rcnn/false_negative", out[13])
                losses["loss_cls_stage{}".format(k)], losses["loss_box_reg_stage{}".format(k)] = loss_cls, loss_box
                continue
            x = _ScaleGradient.apply(x, 1.0 / self.num_cascade_stages)
            scores, deltas = pred(self.box_head[k](x))
            scores, deltas = scores[:R], deltas[:R]
            with st.name_scope("stage{}".format(k)):
                sl = pred.losses_from_tensors(scores, deltas, gtc, prop, gtb, src)
            losses.update({n + "_stage{}".format(k): v for n, v in sl.items()})
        return losses

    def _forward_box(self, features, proposals, targets=None, only_gt_proposals=False):
        if (self.training and not only_gt_proposals and targets is not None and len(proposals) and sum(len(p) for p in proposals) > 0
                and proposals[0].proposal_boxes.tensor.is_cuda and all(bp.fused_supported for bp in self.box_predictor)):
            return self._forward_box_train(features, proposals, targets)
        if (not self.training) and self.mult_proposal_score:
            pscores = [p.get("scores") if p.has("scores") else p.get("objectness_logits") for p in proposals]
        features = [features[f] for f in self.box_in_features]
        outs, prev = [], None
        image_sizes = [x.image_size for x in proposals]
        for k in range(self.num_cascade_stages):
            if k > 0:
                proposals = self._create_proposals_from_boxes(prev, image_sizes, [p.objectness_logits for p in proposals])
                if self.training:
                    proposals = self._match_and_label_boxes(proposals, k, targets)
            preds = self._run_stage(features, proposals, k)
            prev = self.box_predictor[k].predict_boxes(preds, proposals)
            outs.append((self.box_predictor[k], preds, proposals))
        if self.training:
            losses = {}
            st = get_event_storage()
            for stage, (pred, preds, props) in enumerate(outs):
                with st.name_scope("stage{}".format(stage)):
                    sl = pred.no_grad_losses(preds, props) if only_gt_proposals else pred.losses(preds, props)
                losses.update({k + "_stage{}".format(stage): v for k, v in sl.items()})
            return losses
        scores_per_stage = [h[0].predict_probs(h[1], h[2]) for h in outs]
        scores = [sum(list(s)) * (1.0 / self.num_cascade_stages) for s in zip(*scores_per_stage)]
        if self.mult_proposal_score:
            scores = [(s * ps[:, None]) ** 0.5 for s, ps in zip(scores, pscores)]
        if self.one_class_per_proposal:
            scores = [s * (s == s[:, :-1].max(dim=1)[0][:, None]).float() for s in scores]
        pred, preds, props = outs[-1]
        boxes = pred.predict_boxes(preds, props)
        inst, _ = fast_rcnn_inference(boxes, scores, image_sizes, pred.test_score_thresh, pred.test_nms_thresh,
                                      pred.test_topk_per_image)
        return inst

    def _forward_mask(self, features, instances):
        if not self.mask_on:
            return {} if self.training else instances
        if self.training:
            instances, _ = select_foreground_proposals(instances, self.num_classes)
            if not self.divergen_mask_loss:
                instances = [i[i.instance_source == 0] for i in instances]
        feats = [features[f] for f in self.mask_in_features]
        boxes = [x.proposal_boxes if self.training else x.pred_boxes for x in instances]
        if self.training and boxes and boxes[0].tensor.is_cuda:
            rows = torch.cat([b.tensor for b in boxes]) if len(boxes) > 1 else boxes[0].tensor
            x = self.mask_pooler.forward_rows(feats, rows.float(), [len(b) for b in boxes], pad_to=64)
        else:
            x = self.mask_pooler(feats, boxes, pad_to=64 if self.training else 0)
        return self.mask_head(x, instances)

    def forward(self, images, features, proposals, targets=None, ann_type="box", only_gt_proposals=False, **kwargs):
        if self.training:
            assert ann_type == "box", "image-label / caption co-training is outside the shipped configs"
            proposals = self.label_and_sample_proposals(proposals, targets, only_gt_proposals)
            if only_gt_proposals:
                losses = self._forward_box(features, proposals, targets, only_gt_proposals=True)
                if targets[0].has("gt_masks"):
                    losses.update({k: v * self.mask_weight for k, v in self._forward_mask(features, proposals).items()})
                return proposals, losses
            losses = self._forward_box(features, proposals, targets)
            stop_at = self.__dict__.pop("_early_box_backward", None)
            if stop_at is not None and losses and all(v.requires_grad for v in losses.values()):
                # (meta_arch/custom_rcnn.py early_proposal_backward) the cascade's losses are back-propagated HERE, up to the feature maps in
                # `stop_at` -- whose gradient maps the poolers add into in place -- so that the device has the three stages' backward to run
                # while the host issues the mask head, the loss sum and the start of the final backward (the host-bound stretch of the step)
                tot = torch.stack([v.float().reshape(()) for v in losses.values()]).sum()
                params = [q for m in (self.box_head, self.box_predictor) for q in m.parameters() if q.requires_grad]
                torch.autograd.backward(tot, inputs=list(stop_at) + params)       # (parameters that take their gradient through autograd)
                losses = {k: v.detach() for k, v in losses.items()}
            if targets[0].has("gt_masks"):
                losses.update({k: v * self.mask_weight for k, v in self._forward_mask(features, proposals).items()})
            elif self.mask_on:
                losses["loss_mask"] = torch.zeros((1,), device=proposals[0].objectness_logits.device)[0]
            return proposals, losses
        inst = self._forward_box(features, proposals)
        return self._forward_mask(features, inst), {}
