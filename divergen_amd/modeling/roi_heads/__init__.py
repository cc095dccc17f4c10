from .poolers import ROIPooler  # noqa
from .box_head import FastRCNNConvFCHead, build_box_head  # noqa
from .mask_head import MaskRCNNConvUpsampleHead, build_mask_head, mask_rcnn_loss  # noqa
from .detic_fast_rcnn import DeticFastRCNNOutputLayers  # noqa
from .detic_roi_heads import DeticCascadeROIHeads  # noqa
