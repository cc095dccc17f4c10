"""CenterNet proposal generator (ONLY_PROPOSAL + WITH_AGN_HM configuration of DiverGen's YAMLs).
Mirrors CN/modeling/dense_heads/centernet.py:30-737.

MI355X-first changes: dense target assignment is one HIP kernel without M x N temporaries
(dgx_centernet_targets); NMS is the on-device bitmask kernel (dgx_nms_sorted); the loss
normalisers stay on the device (the reference's two `.item()` round trips after all-reduce,
centernet.py:259-260,289, become device-side divisions); post-NMS top-k is taken on the GPU
(no `.cpu()` kthvalue, centernet.py:727-731) with the same ">= k-th score" tie rule.
"""
import ctypes
import os

import torch
import torch.distributed as dist
from torch import nn

from .. import PROPOSAL_GENERATOR_REGISTRY
from ...config import configurable
from ...layers import centernet_targets, nms, nms_batched_sorted
from ...structures import Boxes, Instances, ProposalBatch
from ...utils.comm import get_world_size
from .centernet_head import CenterNetHead

_FUSED_CN_LOSSES = True      # dgx_centernet_losses; the composed form is the reference of its parity test
_OWN_TOPK = True             # dgx_topk_index_rows / dgx_sort_rows_desc; torch.topk / torch.sort are the reference of their parity tests

INF = 100000000


def reduce_sum(t):
    if get_world_size() < 2:
        return t
    t = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


@PROPOSAL_GENERATOR_REGISTRY.register()
class CenterNet(nn.Module):
    @configurable
    def __init__(self, in_channels=256, *, num_classes=80, in_features=("p3", "p4", "p5", "p6", "p7"),
                 strides=(8, 16, 32, 64, 128), score_thresh=0.05, hm_min_overlap=0.8, loc_loss_type="giou",
                 min_radius=4, hm_focal_alpha=0.25, hm_focal_beta=4, loss_gamma=2.0, reg_weight=2.0,
                 not_norm_reg=True, with_agn_hm=False, only_proposal=False, as_proposal=False, not_nms=False,
                 pos_weight=1.0, neg_weight=1.0, sigmoid_clamp=1e-4, ignore_high_fp=-1.0, center_nms=False,
                 sizes_of_interest=((0, 80), (64, 160), (128, 320), (256, 640), (512, 10000000)), more_pos=False,
                 pre_nms_topk_train=1000, pre_nms_topk_test=1000, post_nms_topk_train=100, post_nms_topk_test=100,
                 nms_thresh_train=0.6, nms_thresh_test=0.6, no_reduce=False, centernet_head=None, **unused):
        super().__init__()
        if not (only_proposal and with_agn_hm) or more_pos or center_nms or loc_loss_type != "giou":
            raise NotImplementedError("only the shipped CenterNet2 proposal configuration "
                                      "(ONLY_PROPOSAL, WITH_AGN_HM, giou, no MORE_POS/CENTER_NMS) is built")
        self.num_classes, self.in_features, self.strides = num_classes, tuple(in_features), tuple(strides)
        self.score_thresh, self.min_radius, self.hm_min_overlap = score_thresh, min_radius, hm_min_overlap
        self.hm_focal_alpha, self.hm_focal_beta, self.loss_gamma = hm_focal_alpha, hm_focal_beta, loss_gamma
        self.reg_weight, self.not_norm_reg, self.not_nms = reg_weight, not_norm_reg, not_nms
        self.with_agn_hm, self.only_proposal = with_agn_hm, only_proposal
        self.pos_weight, self.neg_weight = pos_weight, neg_weight
        self.sigmoid_clamp, self.ignore_high_fp = sigmoid_clamp, ignore_high_fp
        self.sizes_of_interest = [list(s) for s in sizes_of_interest]
        self.pre_nms_topk_train, self.pre_nms_topk_test = pre_nms_topk_train, pre_nms_topk_test
        self.post_nms_topk_train, self.post_nms_topk_test = post_nms_topk_train, post_nms_topk_test
        self.nms_thresh_train, self.nms_thresh_test, self.no_reduce = nms_thresh_train, nms_thresh_test, no_reduce
        self.centernet_head = centernet_head if centernet_head is not None else CenterNetHead(
            in_channels=in_channels, num_levels=len(in_features), with_agn_hm=with_agn_hm, only_proposal=only_proposal)

    @classmethod
    def from_config(cls, cfg, input_shape):
        c = cfg.MODEL.CENTERNET
        return dict(in_channels=input_shape[c.IN_FEATURES[0]].channels, num_classes=c.NUM_CLASSES,
                    in_features=c.IN_FEATURES, strides=c.FPN_STRIDES, score_thresh=c.INFERENCE_TH,
                    loc_loss_type=c.LOC_LOSS_TYPE, hm_min_overlap=c.HM_MIN_OVERLAP, min_radius=c.MIN_RADIUS,
                    hm_focal_alpha=c.HM_FOCAL_ALPHA, hm_focal_beta=c.HM_FOCAL_BETA, loss_gamma=c.LOSS_GAMMA,
                    reg_weight=c.REG_WEIGHT, not_norm_reg=c.NOT_NORM_REG, with_agn_hm=c.WITH_AGN_HM,
                    only_proposal=c.ONLY_PROPOSAL, as_proposal=c.AS_PROPOSAL, not_nms=c.NOT_NMS,
                    pos_weight=c.POS_WEIGHT, neg_weight=c.NEG_WEIGHT, sigmoid_clamp=c.SIGMOID_CLAMP,
                    ignore_high_fp=c.IGNORE_HIGH_FP, center_nms=c.CENTER_NMS, sizes_of_interest=c.SOI,
                    more_pos=c.MORE_POS, pre_nms_topk_train=c.PRE_NMS_TOPK_TRAIN, pre_nms_topk_test=c.PRE_NMS_TOPK_TEST,
                    post_nms_topk_train=c.POST_NMS_TOPK_TRAIN, post_nms_topk_test=c.POST_NMS_TOPK_TEST,
                    nms_thresh_train=c.NMS_TH_TRAIN, nms_thresh_test=c.NMS_TH_TEST, no_reduce=c.NO_REDUCE,
                    centernet_head=CenterNetHead(cfg, [input_shape[f] for f in c.IN_FEATURES]))

    # ---------------------------------------------------------------- forward
    def forward(self, images, features_dict, gt_instances):
        features = [features_dict[f] for f in self.in_features]
        flat = self._run_head_flat(features) if self.training and not self.not_nms else None
        if flat is not None:
            # graphed training path: the captured segment ends in the flattened (sum_l B h_l w_l, 4) / (sum_l B h_l w_l) fp32
            # tensors the losses take (the per-level float() / permute / reshape / cat of centernet.py:179-235 run inside the
            # graph), so the replayed backward receives TWO contiguous gradients instead of ten strided per-level ones (round 2:
            # 83 copy / fill launches per step inside torch's GraphedBackward); the proposal decode reads per-level NHWC views
            reg_pred, agn_hm_pred = flat
            shapes = [(int(x.shape[2]), int(x.shape[3])) for x in features]
            B = int(features[0].shape[0])
            raw_hm, raw_reg, off = [], [], 0
            for h, w in shapes:
                n = B * h * w
                raw_reg.append(reg_pred.detach()[off:off + n].view(B, h, w, 4).permute(0, 3, 1, 2))
                raw_hm.append(agn_hm_pred.detach()[off:off + n].view(B, h, w, 1).permute(0, 3, 1, 2))
                off += n
            pos_inds, reg_targets, flattened_hms = self._get_ground_truth(shapes, gt_instances)
            reg_pred_per_level = agn_hm_pred_per_level = grids = None
        else:
            reg_pred_per_level, agn_hm_pred_per_level = self._run_head(features)
            raw_hm = raw_reg = None
            if self.training and reg_pred_per_level[0].is_cuda and not self.not_nms:
                raw_hm, raw_reg = [a.detach() for a in agn_hm_pred_per_level], [r.detach() for r in reg_pred_per_level]
            reg_pred_per_level = [r.float() for r in reg_pred_per_level]
            agn_hm_pred_per_level = [a.float() for a in agn_hm_pred_per_level]
            grids = self.compute_grids(features)
            shapes = [(int(x.shape[2]), int(x.shape[3])) for x in reg_pred_per_level]
            if not self.training:
                hms = [x.sigmoid() for x in agn_hm_pred_per_level]
                proposals = self.predict_instances(grids, hms, reg_pred_per_level, images.image_sizes)
                for p in proposals:
                    p.proposal_boxes = p.get("pred_boxes")
                    p.objectness_logits = p.get("scores")
                    p.remove("pred_boxes")
                return proposals, {}
            pos_inds, reg_targets, flattened_hms = self._get_ground_truth(shapes, gt_instances)
            reg_pred = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, 4) for x in reg_pred_per_level], dim=0)
            agn_hm_pred = torch.cat([x.permute(0, 2, 3, 1).reshape(-1) for x in agn_hm_pred_per_level], dim=0)
        losses = self.losses(pos_inds, reg_targets, flattened_hms, reg_pred, agn_hm_pred)
        with torch.no_grad():
            proposals = self._predict_instances_fused(raw_hm, raw_reg, images.image_sizes) if raw_hm is not None else None
            if proposals is None:
                if agn_hm_pred_per_level is None:      # flattened path whose layout the fused decode refused: per-level copies
                    agn_hm_pred_per_level, reg_pred_per_level = [t.contiguous() for t in raw_hm], [t.contiguous() for t in raw_reg]
                    grids = self.compute_grids(features)
                hms = [x.detach().sigmoid() for x in agn_hm_pred_per_level]
                proposals = self.predict_instances(grids, hms, [r.detach() for r in reg_pred_per_level], images.image_sizes)
        for p in proposals:
            p.proposal_boxes = p.get("pred_boxes")
            p.objectness_logits = p.get("scores")
            p.remove("pred_boxes")
            p.remove("scores")
            if p.has("pred_classes"):
                p.remove("pred_classes")
        return proposals, losses

    def _run_head_flat(self, features):
        """Tower + predictors on every level as one captured hipGraph pair ending in the flattened loss operands
        (reg_pred (M, 4) fp32, agn_hm_pred (M,) fp32); None when the segment cannot be replayed (eval, capture in progress,
        graphs switched off, CPU tensors)."""
        seg = self.__dict__.get("_segment")
        if seg is None:
            from ...utils.graphs import GraphedSegment
            seg = self.__dict__["_segment"] = GraphedSegment(_HeadSegment(self.centernet_head))
        if self.with_agn_hm and seg.usable(features):
            return seg(*features)
        return None

    def _run_head(self, features):
        _, reg, hm = self.centernet_head(features)
        return reg, hm

    @staticmethod
    def _topk_indices_torch(scores, sizes, offs, big, pre_topk, small_idx, cache, key, B, dev):
        """torch.topk form of the candidate indices (levels beyond the own kernel's row limit; the reference of its parity test)."""
        if len(big) > 1:
            # the per-level top-k of all large levels as ONE torch.topk over rows padded with -inf to the largest level (the selection runs one
            # workgroup per row and takes as long for 4 096 as for 16 384 entries: two launches of ~80 us were one too many); rows are
            # independent, the padding can never be selected (n > k real entries, all > -inf)
            nmax = max(sizes[l] for l in big)
            pk = ("topk_pad", key)
            if pk not in cache:
                cache[pk] = (torch.full((B, len(big), nmax), float("-inf"), dtype=torch.float32, device=dev),
                             torch.tensor([offs[l] for l in big], dtype=torch.int64, device=dev).view(1, len(big), 1))
            pad, boff = cache[pk]
            for j, l in enumerate(big):
                pad[:, j, :sizes[l]].copy_(scores[:, offs[l]:offs[l] + sizes[l]])
            parts = [(pad.view(B * len(big), nmax).topk(pre_topk, dim=1)[1].view(B, len(big), pre_topk) + boff).view(B, len(big) * pre_topk)]
        else:
            parts = [scores[:, offs[l]:offs[l] + sizes[l]].topk(pre_topk, dim=1)[1] + offs[l] for l in big]
        if small_idx is not None:
            parts.append(small_idx)
        return torch.cat(parts, 1).contiguous() if len(parts) > 1 else parts[0].contiguous()

    def compute_grids(self, features):
        key = tuple((int(f.shape[-2]), int(f.shape[-1]), str(f.device)) for f in features)
        cache = self.__dict__.setdefault("_grid_cache", {})
        if key not in cache:       # geometry constants: built once per feature-map signature
            cache[key] = self._compute_grids(features)
        return cache[key]

    def _compute_grids(self, features):
        grids = []
        for level, f in enumerate(features):
            h, w = f.shape[-2:]
            s = self.strides[level]
            xs = torch.arange(0, w * s, step=s, dtype=torch.float32, device=f.device)
            ys = torch.arange(0, h * s, step=s, dtype=torch.float32, device=f.device)
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            grids.append(torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1) + s // 2)
        return grids

    @torch.no_grad()
    def _get_ground_truth(self, shapes, gt_instances):
        boxes = [g.gt_boxes.tensor for g in gt_instances]
        if boxes[0].is_cuda and boxes[0].dtype == torch.float32:
            # dense targets + the fixed-length (index, cared) list of the positives: two launches over the same box list
            reg, hm, inds = centernet_targets(boxes, shapes, self.strides, self.sizes_of_interest, self.hm_min_overlap,
                                              self.min_radius, label_inds=True)
            return inds, reg, hm
        reg, hm = centernet_targets(boxes, shapes, self.strides, self.sizes_of_interest, self.hm_min_overlap,
                                    self.min_radius)
        return self._get_label_inds(boxes, shapes, masked=boxes[0].is_cuda), reg, hm

    def _get_label_inds(self, boxes_list, shapes, masked=False):
        """centernet.py:439-483 on the device (n x L integers per image).  masked=True returns (indices, cared) of
        fixed length sum_i n_i * L instead of the compacted index list: the selection `ind[cared]` is the only
        data-dependent shape of the CenterNet losses and costs a device->host read per image."""
        dev = boxes_list[0].device
        L, B = len(self.strides), len(boxes_list)
        key = (tuple(map(tuple, shapes)), B, str(dev))
        cache = self.__dict__.setdefault("_label_consts", {})
        if key not in cache:     # constants of the feature-map geometry: uploaded once, not once per step
            hw = torch.tensor(shapes, dtype=torch.int64, device=dev)
            loc = hw[:, 0] * hw[:, 1]
            bases = torch.cumsum(torch.cat([loc.new_zeros(1), B * loc[:-1]]), 0)
            cache[key] = (hw, loc, bases, torch.tensor(self.strides, dtype=torch.float32, device=dev),
                          torch.tensor(self.sizes_of_interest, dtype=torch.float32, device=dev))
        hw, loc, bases, st, sr = cache[key]
        out, keep = [], []
        for i, bx in enumerate(boxes_list):
            c = (bx[:, :2] + bx[:, 2:]) / 2          # (slices, not list indexing: a list index is uploaded every call)
            ci = (c[:, None, :] / st[None, :, None]).long()
            ind = bases[None] + i * loc[None] + ci[:, :, 1] * hw[None, :, 1] + ci[:, :, 0]
            crit = ((bx[:, 2:] - bx[:, :2]) ** 2).sum(dim=1) ** 0.5 / 2
            cared = (crit[:, None] >= sr[None, :, 0]) & (crit[:, None] <= sr[None, :, 1])
            if masked:
                out.append(ind.reshape(-1))
                keep.append(cared.reshape(-1))
            else:
                out.append(ind[cared].reshape(-1))
        if masked:
            return torch.cat(out, dim=0).long(), torch.cat(keep, dim=0)
        return torch.cat(out, dim=0).long()

    def _losses_fused(self, pos_inds, reg_targets, flattened_hms, reg_pred, agn_hm_pred):
        """`losses` through libdgx's dgx_centernet_losses: raw sums + unscaled gradients from one pass; the normalisers
        (all-reduced across ranks unless NO_REDUCE) and the three divisions stay here, on device scalars."""
        world = get_world_size()
        idx, cared = pos_inds if isinstance(pos_inds, tuple) else (pos_inds, None)
        alpha = self.hm_focal_alpha
        loc_num, pos, neg, s, npos = _CenterNetLosses.apply(
            reg_pred, agn_hm_pred, reg_targets, flattened_hms, idx, cared, bool(self.not_norm_reg), float(self.hm_focal_beta),
            float(self.loss_gamma), float(self.sigmoid_clamp), float(self.ignore_high_fp),
            float(alpha) if alpha >= 0 else 1.0, float(1 - alpha) if alpha >= 0 else 1.0)
        num_pos_local = npos.reshape(1)
        total_num_pos = num_pos_local * world if self.no_reduce else reduce_sum(num_pos_local)
        num_pos_avg = torch.clamp(total_num_pos / world, min=1.0)[0]
        reg_norm = torch.clamp((s if self.no_reduce else reduce_sum(s)) / (1 if self.no_reduce else world), min=1.0)
        return {"loss_centernet_loc": self.reg_weight * loc_num / reg_norm,
                "loss_centernet_agn_pos": self.pos_weight * pos / num_pos_avg,
                "loss_centernet_agn_neg": self.neg_weight * neg / num_pos_avg}

    def losses(self, pos_inds, reg_targets, flattened_hms, reg_pred, agn_hm_pred):
        if (_FUSED_CN_LOSSES and reg_pred.is_cuda and reg_pred.dtype == torch.float32 and agn_hm_pred.dtype == torch.float32
                and self.with_agn_hm and agn_hm_pred.dim() == 1):      # (__init__ only admits the giou loss)
            return self._losses_fused(pos_inds, reg_targets, flattened_hms, reg_pred, agn_hm_pred)
        world = get_world_size()
        if isinstance(pos_inds, tuple):      # (indices, cared): fixed length, count stays on the device
            num_pos_local = pos_inds[1].sum().float().reshape(1)
        else:
            num_pos_local = torch.tensor([float(pos_inds.numel())], device=reg_pred.device)
        total_num_pos = num_pos_local * world if self.no_reduce else reduce_sum(num_pos_local)
        num_pos_avg = torch.clamp(total_num_pos / world, min=1.0)[0]
        losses = {}
        reg_mask = (reg_targets.max(dim=1)[0] >= 0).float()
        w = flattened_hms.max(dim=1)[0]
        reg_weight_map = (w * 0 + 1 if self.not_norm_reg else w) * reg_mask
        s = reg_weight_map.sum()
        reg_norm = torch.clamp((s if self.no_reduce else reduce_sum(s)) / (1 if self.no_reduce else world), min=1.0)
        # masked (instead of index-selected) GIoU: identical sum, no nonzero() sync
        tgt = torch.where(reg_mask[:, None] > 0, reg_targets, torch.ones_like(reg_targets))
        prd = torch.where(reg_mask[:, None] > 0, reg_pred, torch.ones_like(reg_pred))
        losses["loss_centernet_loc"] = self.reg_weight * (_giou(prd, tgt) * reg_weight_map).sum() / reg_norm
        pos, neg = _binary_heatmap_focal_loss(agn_hm_pred, w, pos_inds, self.hm_focal_alpha, self.hm_focal_beta,
                                              self.loss_gamma, self.sigmoid_clamp, self.ignore_high_fp)
        losses["loss_centernet_agn_pos"] = self.pos_weight * pos / num_pos_avg
        losses["loss_centernet_agn_neg"] = self.neg_weight * neg / num_pos_avg
        return losses

    # ---------------------------------------------------------------- decoding
    @staticmethod
    def _nhwc_layout(t):
        """(pixel stride, ok): a logical (B, C, h, w) tensor whose channels of a pixel sit `stride(1) == 1` apart in a dense
        (B, h, w, pixel_stride) image -- what the channels-last head produces, including channel slices of it."""
        B, C, h, w = t.shape
        ps = t.stride(3) if w > 1 else (t.stride(2) if h > 1 else max(C, 1))
        ok = (C == 1 or t.stride(1) == 1) and (w == 1 or t.stride(3) == ps) and (h == 1 or t.stride(2) == w * ps) \
            and (B == 1 or t.stride(0) == h * w * ps) and ps >= C
        return ps, ok

    @torch.no_grad()
    def _predict_instances_fused(self, hm_logits, reg_maps, image_sizes):
        """predict_instances for training on the GPU: libdgx kernels only (dgx_centernet_scores / dgx_topk_index_rows / _decode / dgx_sort_rows_desc /
        dgx_nms_batched / _finalize); fixed-length lists + validity flags, no host round trip.  None = layouts it does not take."""
        import ctypes
        from ... import _lib as L
        Lv, B = len(hm_logits), int(hm_logits[0].shape[0])
        lay_h = [self._nhwc_layout(t) for t in hm_logits]
        lay_r = [self._nhwc_layout(t) for t in reg_maps]
        if (not all(ok for _, ok in lay_h + lay_r) or len({ps for ps, _ in lay_h}) != 1 or len({ps for ps, _ in lay_r}) != 1
                or len({t.dtype for t in hm_logits + reg_maps}) != 1 or hm_logits[0].dtype not in (torch.float32, torch.bfloat16)):
            return None
        dev = hm_logits[0].device
        pre_topk, post_topk, thr_nms = self.pre_nms_topk_train, self.post_nms_topk_train, self.nms_thresh_train
        sizes = [int(t.shape[2] * t.shape[3]) for t in hm_logits]
        M = sum(sizes)
        hw = (ctypes.c_int32 * (2 * Lv))(*[v for t in hm_logits for v in (int(t.shape[2]), int(t.shape[3]))])
        st = (ctypes.c_int32 * Lv)(*self.strides[:Lv])
        hp = (ctypes.c_void_p * Lv)(*[t.data_ptr() for t in hm_logits])
        rp = (ctypes.c_void_p * Lv)(*[t.data_ptr() for t in reg_maps])
        code = L.dtype_code(hm_logits[0])
        scores = torch.empty(B, M, dtype=torch.float32, device=dev)
        n_valid = torch.empty(B, dtype=torch.int32, device=dev)
        lib = L.lib()
        L.check(lib.dgx_centernet_scores(hp, lay_h[0][0], 0, hw, st, Lv, B, float(self.score_thresh), L.ptr(scores), L.ptr(n_valid), code,
                                         L.stream()), "dgx_centernet_scores")
        key = (tuple(sizes), B, pre_topk, str(dev), "fused")
        cache = self.__dict__.setdefault("_decode_consts", {})
        if key not in cache:
            offs, tot = [], 0
            for n in sizes:
                offs.append(tot)
                tot += n
            small = [torch.arange(offs[l], offs[l] + sizes[l], device=dev) for l in range(Lv) if sizes[l] <= pre_topk]
            cache[key] = (offs, torch.cat(small)[None].expand(B, -1).contiguous() if small else None)
        offs, small_idx = cache[key]
        big = [l for l, n in enumerate(sizes) if n > pre_topk]
        if big and _OWN_TOPK and max(sizes[l] for l in big) <= 32768:
            # the per-level top-k of all large levels of all images in ONE launch (dgx_topk_index_rows: radix select + order-preserving
            # compaction per (image, level)), written straight into the candidate index buffer whose tail -- the levels taken whole --
            # is constant
            tk = ("own_topk", key)
            if tk not in cache:
                nb = len(big)
                Kc0 = nb * pre_topk + (small_idx.shape[1] if small_idx is not None else 0)
                buf = torch.empty(B, Kc0, dtype=torch.int64, device=dev)
                if small_idx is not None:
                    buf[:, nb * pre_topk:] = small_idx
                cache[tk] = (buf, torch.tensor([offs[l] for l in big], dtype=torch.int32, device=dev),
                             torch.tensor([sizes[l] for l in big], dtype=torch.int32, device=dev), (ctypes.c_int32 * nb)(*[sizes[l] for l in big]))
            idx, d_off, d_n, h_n = cache[tk]
            L.check(lib.dgx_topk_index_rows(L.ptr(scores), M, B, L.ptr(d_off), L.ptr(d_n), h_n, len(big), pre_topk, L.ptr(idx), idx.shape[1],
                                            L.stream()), "dgx_topk_index_rows")
        else:
            idx = self._topk_indices_torch(scores, sizes, offs, big, pre_topk, small_idx, cache, key, B, dev)
        Kc = int(idx.shape[1])
        boxes = torch.empty(B, Kc, 4, dtype=torch.float32, device=dev)
        sc = torch.empty(B, Kc, dtype=torch.float32, device=dev)
        L.check(lib.dgx_centernet_decode(rp, lay_r[0][0], 0, hw, st, Lv, B, L.ptr(idx), Kc, L.ptr(scores), float(self.score_thresh),
                                         L.ptr(boxes), L.ptr(sc), L.ptr(n_valid), code, L.stream()), "dgx_centernet_decode")
        if _OWN_TOPK and Kc <= 16384:
            sc_sorted, order = torch.empty_like(sc), torch.empty(B, Kc, dtype=torch.int64, device=dev)
            L.check(lib.dgx_sort_rows_desc(L.ptr(sc), B, Kc, L.ptr(sc_sorted), L.ptr(order), L.stream()), "dgx_sort_rows_desc")
            sc = sc_sorted
        else:
            sc, order = torch.sort(sc, dim=1, descending=True, stable=True)
        sorted_boxes = torch.empty_like(boxes)
        L.check(lib.dgx_gather_boxes(L.ptr(boxes), L.ptr(order.contiguous()), B, Kc, L.ptr(sorted_boxes), L.stream()), "dgx_gather_boxes")
        boxes = sorted_boxes
        cap = min(Kc, post_topk + 64)
        keep_idx, num_keep = nms_batched_sorted(boxes, sc, n_valid, thr_nms, max_keep=post_topk, cap=cap)
        o_box = torch.empty(B, cap, 4, dtype=torch.float32, device=dev)
        o_sc = torch.empty(B, cap, dtype=torch.float32, device=dev)
        o_valid = torch.empty(B, cap, dtype=torch.uint8, device=dev)
        L.check(lib.dgx_centernet_finalize(L.ptr(boxes), L.ptr(sc), L.ptr(keep_idx.contiguous()), L.ptr(num_keep.contiguous()), B, Kc, cap,
                                           L.ptr(o_box), L.ptr(o_sc), L.ptr(o_valid), L.stream()), "dgx_centernet_finalize")
        valid = o_valid.view(torch.bool)
        results = ProposalBatch()
        for i in range(B):
            inst = Instances(image_sizes[i])
            inst.pred_boxes = Boxes(o_box[i])
            inst.scores = o_sc[i]
            inst.proposal_valid = valid[i]
            results.append(inst)
        results.batch = (o_box, o_sc, valid)
        return results

    @torch.no_grad()
    def predict_instances(self, grids, hms, reg_pred, image_sizes):
        B = hms[0].shape[0]
        pre_topk = self.pre_nms_topk_train if self.training else self.pre_nms_topk_test
        post_topk = self.post_nms_topk_train if self.training else self.post_nms_topk_test
        thr_nms = self.nms_thresh_train if self.training else self.nms_thresh_test
        # All levels decoded together (the reference loops over levels, centernet.py:640-688): per-level top-k only where a
        # level has more locations than PRE_NMS_TOPK (its other levels keep every location), then ONE gather of the grid
        # centres / strides / regression rows over the concatenated candidate list and ONE box decode.
        sizes = [int(h.shape[2] * h.shape[3]) for h in hms]
        key = (tuple(sizes), B, pre_topk, str(hms[0].device))
        cache = self.__dict__.setdefault("_decode_consts", {})
        if key not in cache:
            offs, tot = [], 0
            for n in sizes:
                offs.append(tot)
                tot += n
            grid_all = torch.cat(grids, 0)                                              # (M,2)
            stride_all = torch.cat([grids[l].new_full((sizes[l],), float(self.strides[l])) for l in range(len(sizes))])
            small = [torch.arange(offs[l], offs[l] + sizes[l], device=grid_all.device) for l in range(len(sizes))
                     if sizes[l] <= pre_topk]
            small_idx = torch.cat(small)[None].expand(B, -1).contiguous() if small else None
            cache[key] = (offs, grid_all, stride_all, small_idx)
        offs, grid_all, stride_all, small_idx = cache[key]
        hm_all = torch.cat([h.reshape(B, -1) for h in hms], 1)                            # (B,M), C == 1
        reg_all = torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, 4) for r in reg_pred], 1)   # (B,M,4)
        hm_all = torch.where(hm_all > self.score_thresh, hm_all, hm_all.new_full((), -1.0))
        idx_parts, val_parts = [], []
        for l, n in enumerate(sizes):
            if n > pre_topk:
                v, i = hm_all[:, offs[l]:offs[l] + n].topk(pre_topk, dim=1)
                idx_parts.append(i + offs[l])
                val_parts.append(v)
        if small_idx is not None:
            idx_parts.append(small_idx)
            val_parts.append(torch.gather(hm_all, 1, small_idx))
        idx = torch.cat(idx_parts, 1)                                                     # (B,K)
        sc = torch.cat(val_parts, 1)
        g = grid_all[idx]                                                                 # (B,K,2)
        r = torch.gather(reg_all, 1, idx[:, :, None].expand(-1, -1, 4)) * stride_all[idx][:, :, None]
        x0, y0 = g[..., 0] - r[..., 0], g[..., 1] - r[..., 1]
        boxes = torch.stack([x0, y0, torch.max(g[..., 0] + r[..., 2], x0 + 0.01), torch.max(g[..., 1] + r[..., 3], y0 + 0.01)], -1)
        # Fixed-shape, sync-free from here: every image keeps its K candidates; the ones at or below the score threshold
        # sort to the end and the device-side count tells the NMS kernel where to stop.
        ok = sc > self.score_thresh
        n_valid = ok.sum(1).to(torch.int32)
        sc = torch.where(ok, torch.sqrt(sc.clamp(min=0)), sc.new_full((), -1.0))
        sc, order = torch.sort(sc, dim=1, descending=True, stable=True)
        boxes = torch.gather(boxes, 1, order[:, :, None].expand(-1, -1, 4))
        K = sc.shape[1]
        if not self.not_nms:
            cap = min(K, post_topk + 64)
            keep_idx, num_keep = nms_batched_sorted(boxes, sc, n_valid, thr_nms, max_keep=post_topk, cap=cap)
            kidx = keep_idx.clamp(min=0).long()
            boxes = torch.gather(boxes, 1, kidx[:, :, None].expand(-1, -1, 4))
            sc = torch.gather(sc, 1, kidx)
        else:
            # no NMS: the reference's top-k-with-ties rule applied to the score-sorted candidates
            cap = min(K, post_topk + 64)
            kth = sc[:, min(post_topk, K) - 1:min(post_topk, K)]
            num_keep = torch.minimum(n_valid, ((sc >= kth) & (sc >= 0)).sum(1).to(torch.int32)).clamp(max=cap)
            boxes, sc = boxes[:, :cap], sc[:, :cap]
        valid = torch.arange(boxes.shape[1], device=sc.device)[None, :] < num_keep[:, None]
        boxes = boxes * valid[:, :, None]
        sc = torch.where(valid, sc, torch.zeros_like(sc))
        results = []
        counts = None if self.training else num_keep.tolist()      # inference returns exact-length lists (one sync)
        for i in range(B):
            inst = Instances(image_sizes[i])
            if counts is None:      # training: fixed length + validity flags (the sampler ignores padded rows)
                inst.pred_boxes = Boxes(boxes[i])
                inst.scores = sc[i]
                inst.proposal_valid = valid[i]
            else:
                inst.pred_boxes = Boxes(boxes[i, :counts[i]])
                inst.scores = sc[i, :counts[i]]
            inst.pred_classes = torch.zeros_like(inst.scores, dtype=torch.int64)
            results.append(inst)
        if counts is None:      # training: the batch-level tensors travel with the list (the RoI heads label / sample them in one go)
            results = ProposalBatch(results)
            results.batch = (boxes, sc, valid)
        return results


class _HeadSegment(nn.Module):
    """CenterNetHead with tensors in / (flattened regression rows, flattened heat-map logits) out: the unit captured as a hipGraph."""

    def __init__(self, head):
        super().__init__()
        self.head = head
        self.amp = False

    def forward(self, *feats):
        # centernet.py:179-235: per level (B, C, h, w) -> (B h w, C), levels stacked
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp, cache_enabled=False):
            return self.head.forward_flat(list(feats))


class _CenterNetLosses(torch.autograd.Function):
    """(reg_pred, logit) -> (weighted GIoU sum, pos loss, neg loss | sum of weights, #positives): dgx_centernet_losses."""

    @staticmethod
    def forward(ctx, reg_pred, logit, reg_targets, hms, idx, cared, not_norm_reg, beta, gamma, clampv, ihf, pos_mul, neg_mul):
        from ... import _lib as L
        reg_pred, logit = reg_pred.contiguous(), logit.contiguous()
        reg_targets, hms = reg_targets.float().contiguous(), hms.float().contiguous()
        M, C, P = reg_pred.shape[0], hms.shape[1] if hms.dim() > 1 else 1, idx.numel()
        idx = idx.contiguous()
        cared_u8 = cared.to(torch.uint8).contiguous() if cared is not None else None
        dev = reg_pred.device
        g_reg = torch.empty(M, 4, dtype=torch.float32, device=dev)
        g_neg = torch.empty(M, dtype=torch.float32, device=dev)
        g_pos = torch.empty(M, dtype=torch.float32, device=dev)
        out = torch.empty(8, dtype=torch.float32, device=dev)
        part = torch.empty(3 * L.lib().dgx_centernet_losses_blocks(M), dtype=torch.float32, device=dev)
        L.check(L.lib().dgx_centernet_losses(L.ptr(reg_pred), L.ptr(reg_targets), L.ptr(hms), L.ptr(logit), L.ptr(idx) if P else None,
                                             L.ptr(cared_u8), M, C, P, int(not_norm_reg), beta, gamma, clampv, ihf, pos_mul, neg_mul,
                                             L.ptr(g_reg), L.ptr(g_neg), L.ptr(g_pos), L.ptr(out), L.ptr(part), L.stream()),
                "dgx_centernet_losses")
        ctx.save_for_backward(g_reg, g_neg, g_pos)
        s_w, n_pos = out[0], out[4]
        ctx.mark_non_differentiable(s_w, n_pos)
        return out[1], out[3], out[2], s_w, n_pos

    @staticmethod
    def backward(ctx, d_loc, d_pos, d_neg, _s, _n):
        g_reg, g_neg, g_pos = ctx.saved_tensors
        d_reg = g_reg * d_loc
        d_logit = torch.addcmul(g_neg * d_neg, g_pos, d_pos)
        return (d_reg, d_logit) + (None,) * 11


def _giou(pred, target):
    """CN/modeling/layers/iou_loss.py:10-63 ('giou'), per-row loss."""
    pl, pt, pr, pb = pred.unbind(1)
    tl, tt, tr, tb = target.unbind(1)
    ta, pa = (tl + tr) * (tt + tb), (pl + pr) * (pt + pb)
    wi = torch.min(pl, tl) + torch.min(pr, tr)
    hi = torch.min(pb, tb) + torch.min(pt, tt)
    ac = (torch.max(pl, tl) + torch.max(pr, tr)) * (torch.max(pb, tb) + torch.max(pt, tt))
    ai = wi * hi
    au = ta + pa - ai
    return 1 - ((ai + 1.0) / (au + 1.0) - (ac - au) / ac)


def _binary_heatmap_focal_loss(inputs, targets, pos_inds, alpha, beta, gamma, sigmoid_clamp, ignore_high_fp):
    """CN/modeling/layers/heatmap_focal_loss.py:51-85."""
    pred = torch.clamp(inputs.sigmoid(), min=sigmoid_clamp, max=1 - sigmoid_clamp)
    neg_weights = torch.pow(1 - targets, beta)
    if isinstance(pos_inds, tuple):          # masked form of the index selection: same sum, no data-dependent shape
        pos_pred = pred[pos_inds[0]]
        pos_loss = torch.log(pos_pred) * torch.pow(1 - pos_pred, gamma) * pos_inds[1].to(pred.dtype)
    else:
        pos_pred = pred[pos_inds]
        pos_loss = torch.log(pos_pred) * torch.pow(1 - pos_pred, gamma)
    neg_loss = torch.log(1 - pred) * torch.pow(pred, gamma) * neg_weights
    if ignore_high_fp > 0:
        neg_loss = (pred < ignore_high_fp).float() * neg_loss
    pos_loss, neg_loss = -pos_loss.sum(), -neg_loss.sum()
    if alpha >= 0:
        pos_loss, neg_loss = alpha * pos_loss, (1 - alpha) * neg_loss
    return pos_loss, neg_loss
