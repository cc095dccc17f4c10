"""ResNet-50 (timm 0.4.9 layout) bottom-up + P3-P7 FPN of the R50 configurations (BASELINE configs[0]).

Mirrors DG/divergen/modeling/backbone/timm.py: `CustomResNet` (:27-57), `TIMM` (:109-151), `build_timm_backbone` (:154-162),
`build_p67_timm_fpn_backbone` (:165-182), `build_p35_timm_fpn_backbone` (:184-201) over timm 0.4.9's `ResNet` / `Bottleneck`
(stride on the 3x3, 1x1-stride downsample, no bias; timm itself is not vendored by the reference: restated here and in
oracle/resnet.py, "parity unpinned" by reference vectors).  Module / parameter / buffer names are timm's, so
`resnet50_miil_21k` checkpoints load (`base.conv1.weight`, `base.layer1.0.bn1.running_mean`, `base.layer2.0.downsample.0.weight` ...).

MI355X-first: activations are channels-last bf16 end to end; every convolution is a libdgx GEMM (1x1 = dgx_gemm_bf16_nt over
pixels, 3x3 stride 1 = implicit GEMM, 3x3 stride 2 = im2col + GEMM, the 7x7 stem = dgx_stem_im2col7x7 + GEMM); every
FrozenBatchNorm2d is folded with the ReLU (and the block's residual add) that follows it into one pass (dgx_affine_act_fwd/bwd);
max-pool keeps its arg-max as a byte and runs its backward as a gather (dgx_maxpool3x3s2_fwd/bwd)."""
import torch
from torch import nn

from .. import BACKBONE_REGISTRY
from .swintransformer import Backbone
from ... import _lib as L
from ...layers.conv_ops import Conv2d, _nhwc
from ...layers.linear_ops import accumulate_grad, shadow, wgrad_into

BF16 = torch.bfloat16


class FrozenBatchNorm2d(nn.Module):
    """D2/layers/batch_norm.py:13-111: fixed statistics and affine parameters (buffers, not parameters), eps 1e-5."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def scale_shift(self):
        """(scale, shift) f32: computed once per state of the buffers (they only change through load_state_dict / .to())."""
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version, self.weight.device)
        c = self.__dict__.get("_ss")
        if c is None or c[0] != key:
            scale = self.weight * (self.running_var + self.eps).rsqrt()
            c = (key, scale.float().contiguous(), (self.bias - self.running_mean * scale).float().contiguous())
            self.__dict__["_ss"] = c
        return c[1], c[2]

    def forward(self, x, residual=None, relu=False):
        """x logical (N,C,H,W) over NHWC storage -> same; optional residual (same layout) and ReLU in the same pass."""
        scale, shift = self.scale_shift()
        xh = _nhwc(x)
        if not (xh.is_cuda and xh.shape[-1] % 8 == 0):
            raise L.DgxError("FrozenBatchNorm2d: GPU input with C %% 8 == 0 required (%s, C %d)" % (xh.device, xh.shape[-1]))
        rh = _nhwc(residual).to(BF16).contiguous() if residual is not None else None
        return _AffineAct.apply(xh.to(BF16).contiguous(), scale, shift, rh, relu).permute(0, 3, 1, 2)


class _AffineAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, res, relu):
        y = torch.empty_like(x)
        C = x.shape[-1]
        L.check(L.lib().dgx_affine_act_fwd(L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(res), L.ptr(y), x.numel() // C, C, int(relu),
                                           L.stream()), "dgx_affine_act_fwd")
        ctx.save_for_backward(y if relu else None, scale)
        ctx.relu, ctx.has_res = relu, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        dy = dy.contiguous()
        C = dy.shape[-1]
        dx = torch.empty_like(dy)
        dres = torch.empty_like(dy) if (ctx.has_res and ctx.relu) else None
        L.check(L.lib().dgx_affine_act_bwd(L.ptr(dy), L.ptr(y), L.ptr(scale), L.ptr(dx), L.ptr(dres), dy.numel() // C, C, int(ctx.relu),
                                           L.stream()), "dgx_affine_act_bwd")
        return dx, None, None, (dres if dres is not None else dy) if ctx.has_res else None, None


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, H, W, C = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(N, Ho, Wo, C, dtype=x.dtype, device=x.device)
        idx = torch.empty(N, Ho, Wo, C, dtype=torch.uint8, device=x.device)
        L.check(L.lib().dgx_maxpool3x3s2_fwd(L.ptr(x), L.ptr(y), L.ptr(idx), N, H, W, C, L.stream()), "dgx_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (N, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        N, H, W, C = ctx.shape
        dx = torch.empty(N, H, W, C, dtype=dy.dtype, device=dy.device)
        L.check(L.lib().dgx_maxpool3x3s2_bwd(L.ptr(dy.contiguous()), L.ptr(idx), L.ptr(dx), N, H, W, C, L.stream()), "dgx_maxpool3x3s2_bwd")
        return dx


def maxpool3x3s2(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on a logical (N,C,H,W) tensor over NHWC storage."""
    xh = _nhwc(x)
    if not (xh.is_cuda and xh.shape[-1] % 8 == 0):
        raise L.DgxError("maxpool3x3s2: GPU input with C %% 8 == 0 required (%s, C %d)" % (xh.device, xh.shape[-1]))
    return _MaxPool.apply(xh.to(BF16).contiguous()).permute(0, 3, 1, 2)


class _StemConv(torch.autograd.Function):
    """conv1 of the ResNet: 7x7 / 2 / pad 3, 3 -> 64, no bias; no input gradient (the image)."""

    @staticmethod
    def forward(ctx, x, weight):
        from ...layers.gemm_ops import gemm_nt
        N, _, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        rows = torch.empty(N * Ho * Wo, 152, dtype=BF16, device=x.device)
        L.check(L.lib().dgx_stem_im2col7x7(L.ptr(x.float().contiguous()), L.ptr(rows), N, H, W, L.stream()), "dgx_stem_im2col7x7")
        w = torch.zeros(weight.shape[0], 152, dtype=BF16, device=x.device)
        w[:, :147] = shadow(weight).reshape(weight.shape[0], 147)
        y = gemm_nt(rows, w)
        ctx.save_for_backward(rows)
        ctx.weight = weight
        return y.view(N, Ho, Wo, weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        rows, = ctx.saved_tensors
        weight = ctx.weight
        if not ctx.needs_input_grad[1]:
            return None, None
        dy2 = dy.reshape(-1, dy.shape[-1]).to(BF16).contiguous()

        def grad():
            g = torch.zeros(weight.shape[0], 152, dtype=torch.float32, device=dy.device)
            wgrad_into(g, dy2, rows, 0.0)
            return g[:, :147].reshape(weight.shape)
        return None, accumulate_grad(weight, grad, gemm_into=lambda g: g.add_(grad()))


class StemConv(nn.Conv2d):
    def forward(self, x):
        if not (x.is_cuda and self.kernel_size == (7, 7) and self.stride == (2, 2) and self.padding == (3, 3)
                and self.in_channels == 3 and self.bias is None and self.out_channels % 8 == 0):
            raise L.DgxError("StemConv: only the ResNet stem (7x7 / 2 / pad 3, 3 -> 8k channels, no bias) on a GPU tensor is built")
        with torch.autocast("cuda", enabled=False):
            return _StemConv.apply(x, self.weight).permute(0, 3, 1, 2)


class Bottleneck(nn.Module):
    """timm.models.resnet.Bottleneck (0.4.9) with cardinality 1, base_width 64, no attention / anti-aliasing: 1x1 -> 3x3 (the
    block's stride) -> 1x1 (x4), each followed by a norm, ReLU after the first two and after the residual add."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x
        y = self.bn1(self.conv1(x), relu=True)
        y = self.bn2(self.conv2(y), relu=True)
        y = self.conv3(y)
        if self.downsample is not None:
            s = x[:, :, ::self.stride, ::self.stride] if self.stride > 1 else x      # a 1x1 convolution with stride s reads every s-th pixel
            shortcut = self.downsample[1](self.downsample[0](s))
        return self.bn3(y, residual=shortcut, relu=True)


class CustomResNet(nn.Module):
    """DG timm.py:27-47 over timm's ResNet(block=Bottleneck, layers=[3, 4, 6, 3]): returns the features picked by out_indices
    from [stem after max-pool, layer1, layer2, layer3, layer4]."""
    feature_info = [dict(num_chs=64, reduction=2, module="act1"), dict(num_chs=256, reduction=4, module="layer1"),
                    dict(num_chs=512, reduction=8, module="layer2"), dict(num_chs=1024, reduction=16, module="layer3"),
                    dict(num_chs=2048, reduction=32, module="layer4")]

    def __init__(self, layers=(3, 4, 6, 3), out_indices=(2, 3, 4)):
        super().__init__()
        self.out_indices = list(out_indices)
        self.conv1 = StemConv(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        inplanes = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            stride = 1 if i == 0 else 2
            blocks = []
            for b in range(n):
                ds = None
                if b == 0 and (stride != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(Conv2d(inplanes, planes * 4, 1, bias=False), FrozenBatchNorm2d(planes * 4))
                blocks.append(Bottleneck(inplanes, planes, stride if b == 0 else 1, ds))
                inplanes = planes * 4
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
        for m in self.modules():                       # timm's init: kaiming_normal_(fan_out, relu) on every convolution
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.bn1(self.conv1(x), relu=True)
        x = maxpool3x3s2(x)
        ret = [x]
        for i in range(4):
            x = getattr(self, "layer%d" % (i + 1))(x)
            ret.append(x)
        return [ret[i] for i in self.out_indices]


class TIMM(Backbone):
    """DG timm.py:109-151 for the ResNet base names; norm 'FrozenBN' is the only one the shipped configs use."""

    def __init__(self, base_name, out_levels, freeze_at=0, norm="FrozenBN"):
        super().__init__()
        if "resnet50" not in base_name:
            raise NotImplementedError("MODEL.TIMM.BASE_NAME '%s': resnet50 / resnet50_in21k are built" % base_name)
        if norm != "FrozenBN":
            raise NotImplementedError("MODEL.TIMM.NORM '%s': FrozenBN (every shipped R50 configuration) is built" % norm)
        self.base = CustomResNet(out_indices=[x - 1 for x in out_levels])
        fi = self.base.feature_info
        self._out_features = ["layer{}".format(x) for x in out_levels]
        self._out_feature_channels = {"layer{}".format(l): fi[l - 1]["num_chs"] for l in out_levels}
        self._out_feature_strides = {"layer{}".format(l): fi[l - 1]["reduction"] for l in out_levels}
        self._size_divisibility = max(self._out_feature_strides.values())
        self.freeze(freeze_at)

    def freeze(self, freeze_at=0):
        mods = ([self.base.conv1] if freeze_at >= 1 else []) + ([self.base.layer1] if freeze_at >= 2 else [])
        for m in mods:
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, x):
        return dict(zip(self._out_features, self.base(x)))

    @property
    def size_divisibility(self):
        return self._size_divisibility


@BACKBONE_REGISTRY.register()
def build_timm_backbone(cfg, input_shape):
    t = cfg.MODEL.TIMM
    return TIMM(t.BASE_NAME, t.OUT_LEVELS, freeze_at=t.FREEZE_AT, norm=t.NORM)


@BACKBONE_REGISTRY.register()
def build_p67_timm_fpn_backbone(cfg, input_shape):
    from .fpn import FPN, LastLevelP6P7_P5
    oc = cfg.MODEL.FPN.OUT_CHANNELS
    return FPN(bottom_up=build_timm_backbone(cfg, input_shape), in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=oc,
               norm=cfg.MODEL.FPN.NORM, top_block=LastLevelP6P7_P5(oc, oc), fuse_type=cfg.MODEL.FPN.FUSE_TYPE)


@BACKBONE_REGISTRY.register()
def build_p35_timm_fpn_backbone(cfg, input_shape):
    from .fpn import FPN
    oc = cfg.MODEL.FPN.OUT_CHANNELS
    return FPN(bottom_up=build_timm_backbone(cfg, input_shape), in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=oc,
               norm=cfg.MODEL.FPN.NORM, top_block=None, fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
