from .swintransformer import SwinTransformer, build_swintransformer_backbone, build_swintransformer_fpn_backbone  # noqa
from .fpn import FPN, LastLevelP6P7_P5  # noqa
