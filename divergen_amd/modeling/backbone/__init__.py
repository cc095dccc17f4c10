from .swintransformer import SwinTransformer, build_swintransformer_backbone, build_swintransformer_fpn_backbone  # noqa
from .fpn import FPN, LastLevelP6P7_P5  # noqa
from .timm import TIMM, build_timm_backbone, build_p67_timm_fpn_backbone, build_p35_timm_fpn_backbone  # noqa
