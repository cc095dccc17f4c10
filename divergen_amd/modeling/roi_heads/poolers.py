"""ROIPooler over the multi-level HIP ROIAlign.  Mirrors D2/modeling/poolers.py:93-245 (ROIAlignV2)."""
import math

import torch
from torch import nn

from ...layers import roi_pooler


class ROIPooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4,
                 out_nhwc=False):
        super().__init__()
        if pooler_type != "ROIAlignV2":
            raise NotImplementedError("only POOLER_TYPE ROIAlignV2 (the shipped configs) is built, got %s" % pooler_type)
        assert canonical_box_size == 224 and canonical_level == 4
        self.output_size = output_size if isinstance(output_size, int) else output_size[0]
        self.scales, self.sampling_ratio, self.out_nhwc = tuple(scales), sampling_ratio, out_nhwc
        mn, mx = -math.log2(scales[0]), -math.log2(scales[-1])
        assert math.isclose(mn, int(mn)) and math.isclose(mx, int(mx)), "Featuremap stride is not power of 2!"
        self.min_level, self.max_level = int(mn), int(mx)
        assert len(scales) == self.max_level - self.min_level + 1

    def forward_rows(self, x, boxes, counts, pad_to=0, grad_scale=1.0):
        """`forward` for RoIs that already sit in ONE (R, 4) tensor with `counts` rows per image (the cascade keeps them that way):
        the image-index column and the shape-padding rows are constants of (counts, pad_to), built once."""
        R = int(boxes.shape[0])
        Rp = R if pad_to <= 0 else -(-R // pad_to) * pad_to
        key = (tuple(counts), Rp, str(boxes.device))
        cache = self.__dict__.setdefault("_rows_cache", {})
        idx = cache.get(key)
        if idx is None:
            if len(cache) > 64:
                cache.clear()
            col = torch.cat([torch.full((n,), float(i)) for i, n in enumerate(counts)] + [torch.zeros(Rp - R)])
            idx = cache[key] = col.to(boxes.device).view(Rp, 1)
        if Rp > R:       # shape-padding rows: empty boxes of image 0
            rois = torch.zeros(Rp, 5, dtype=torch.float32, device=boxes.device)
            rois[:R] = torch.cat([idx[:R], boxes.float()], dim=1)
        else:
            rois = torch.cat([idx, boxes.float()], dim=1)
        return roi_pooler(list(x), rois, self.output_size, self.scales, self.sampling_ratio, self.out_nhwc, grad_scale)

    def forward(self, x, box_lists, pad_to=0):
        """x: list of (N,C,H,W); box_lists: list[Boxes] per image -> (R, C, S, S).
        pad_to > 0 rounds R up to a multiple of pad_to with empty boxes (their output rows are zeros):
        downstream GEMM shapes then repeat from step to step instead of changing with every RoI count,
        which keeps the GEMM library's per-shape solution lookup out of the step."""
        assert len(x) == len(self.scales) and len(box_lists) == x[0].size(0)
        boxes = torch.cat([b.tensor for b in box_lists], dim=0)
        idx = torch.cat([torch.full((len(b),), float(i), dtype=boxes.dtype, device=boxes.device)
                         for i, b in enumerate(box_lists)])
        rois = torch.cat([idx[:, None], boxes], dim=1)
        R = rois.shape[0]
        if pad_to > 0 and R % pad_to:
            rois = torch.cat([rois, rois.new_zeros(pad_to - R % pad_to, 5)], dim=0)
        return roi_pooler(list(x), rois, self.output_size, self.scales, self.sampling_ratio, self.out_nhwc)
