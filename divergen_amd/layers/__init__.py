"""Thin torch wrappers (autograd Functions / modules) over the libdgx C ABI.

Mirrors the role of detectron2/layers (ROIAlign, nms, ...) plus the Swin window ops.  Every op
needs ROCm tensors and libdgx.so; nothing here falls back to eager PyTorch or the CPU oracle.
"""
from .window_ops import window_attention_core, window_gather, window_scatter, shift_regions  # noqa
from .roi_ops import roi_align, roi_pooler, mask_crop  # noqa
from .box_ops import nms, batched_nms, iou_match, nms_batched_sorted  # noqa
from .dense_ops import centernet_targets  # noqa
from .copy_paste import PackedPastes, copy_paste, pack_pastes, pack_pastes_host  # noqa
from .optim_ops import adamw_ema_step, clip_coef, sgd_ema_step  # noqa
