"""Swin backbone with HIP window ops.  Module / parameter names follow the reference so its
checkpoints load: DG/divergen/modeling/backbone/swintransformer.py (classes :28-634, size table
:636-693, builders :695-731).

Differences that are deliberate (MI355X-first):
  * pad + roll + window_partition (and the inverse) are ONE index-mapped copy each (libdgx
    dgx_window_gather/scatter) instead of 4 full-tensor passes;
  * QK^T + rel-pos bias + shift mask + softmax + PV is ONE MFMA kernel per block
    (dgx_window_attention_fwd/bwd); the (B_,nH,N,N) score tensor never exists in HBM;
  * the SW-MSA mask is never materialised: an int8 region id per token (cached per (H,W)) is all
    the kernel needs.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import BACKBONE_REGISTRY, ShapeSpec
from ...layers.conv_ops import patch_embed4x4, patch_embed_rows
from ...structures import PatchRows
from ...layers import linear_ops
from ...layers.linear_ops import GELU, Linear
from ...layers.norm_ops import layernorm_bf16, layernorm_f32out, layernorm_window_gather, patch_merge_layernorm, residual_add
from ...layers import shift_regions, window_attention_core
from ...layers.swin_block import arena_resident, swin_block

from ..._lib import DgxError

_FUSED_BLOCK = True      # tests flip this to compare the one-node block with the composed path (same kernels)


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std)


class DropPath(nn.Module):
    """timm==0.4.9 DropPath: per-sample Bernoulli keep, scale by 1/keep (not vendored by the reference)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
        return x.div(keep) * mask.floor_()


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0):
        super().__init__()
        self.fc1 = Linear(in_features, hidden_features or in_features)
        self.act = GELU()
        self.fc2 = Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        head_dim = dim // num_heads
        if head_dim != 32:
            raise ValueError("libdgx window attention is built for head_dim 32 (every Swin size); got %d" % head_dim)
        self.scale = qk_scale or head_dim ** -0.5
        ws = window_size[0]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        r = torch.arange(ws)
        ii, jj = torch.meshgrid(r, r, indexing="ij")
        pos = torch.stack([ii.reshape(-1), jj.reshape(-1)])
        rel = pos[:, :, None] - pos[:, None, :]
        self.register_buffer("relative_position_index", (rel[0] + ws - 1) * (2 * ws - 1) + (rel[1] + ws - 1))
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = Linear(dim, dim)
        trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x, region=None, nW=1):
        """x (B_, N, C); region int8 (nW, N) or None (= the reference's `mask` argument)."""
        qkv = self.qkv(x)
        o = window_attention_core(qkv.to(torch.bfloat16), self.relative_position_bias_table, region, nW,
                                  self.num_heads, self.window_size[0], self.scale)
        return self.proj(o)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_path=0.0):
        super().__init__()
        assert 0 <= shift_size < window_size
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, (window_size, window_size), num_heads, qkv_bias, qk_scale)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.H = self.W = None

    def forward(self, x, region):
        B, Ltok, C = x.shape
        H, W = self.H, self.W
        assert Ltok == H * W, "input feature has wrong size"
        ws, sh = self.window_size, self.shift_size
        if not (x.is_cuda and C <= 1536 and C % 8 == 0):
            raise DgxError("SwinTransformerBlock: GPU input with C <= 1536, C %% 8 == 0 required (C %d on %s): there is no "
                           "CPU / eager path" % (C, x.device))
        if _FUSED_BLOCK and torch.is_grad_enabled():
            params = (self.norm1.weight, self.norm1.bias, self.attn.qkv.weight, self.attn.qkv.bias,
                      self.attn.relative_position_bias_table, self.attn.proj.weight, self.attn.proj.bias,
                      self.norm2.weight, self.norm2.bias, self.mlp.fc1.weight, self.mlp.fc1.bias,
                      self.mlp.fc2.weight, self.mlp.fc2.bias)
            if arena_resident(params):   # one autograd node for the whole block
                s1, s2 = self._drop_scales(B, x.device)
                cfg = (B, H, W, ws, sh, self.num_heads, self.attn.scale, self.norm1.eps, self.norm2.eps)
                return swin_block(x, region if sh > 0 else None, s1, s2, cfg, params)
        # composed path (parameters outside an arena, or no gradients: evaluation): the same kernels, one autograd node each
        # LN + bf16 cast + pad + roll + partition in one pass
        xw = layernorm_window_gather(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, B, H, W, ws, sh)
        nW = (-(-H // ws)) * (-(-W // ws))
        aw = self.attn(xw, region if sh > 0 else None, nW)
        # window_reverse + roll + crop + DropPath + residual add: one pass each
        s1, s2 = self._drop_scales(B, x.device)
        x = residual_add(x, aw, s1, B, H, W, ws, sh)
        h2 = layernorm_bf16(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return residual_add(x, self.mlp(h2), s2, B, H, W)

    def _drop_scales(self, B, device):
        """Per-sample DropPath factors floor(keep + U)/keep for the attention and MLP branches of this
        block (timm drop_path semantics; call sites swintransformer.py:254-255), one RNG draw for both."""
        p = self.drop_path.drop_prob if isinstance(self.drop_path, DropPath) else 0.0
        if p == 0.0 or not self.training:
            return None, None
        # drawn for all blocks at once by SwinTransformer.forward and KEPT until the next draw overwrites it: with
        # MODEL.SWIN.USE_CHECKPOINT the block runs a second time in backward and must see the same factors (the RNG state
        # torch.utils.checkpoint restores cannot reproduce a draw that happened outside the block)
        pre = self.__dict__.get("_dp_scales", None)
        if pre is not None and pre.shape[1] == B:
            return pre[0], pre[1]
        keep = 1.0 - p
        s = torch.floor(keep + torch.rand(2, B, device=device, dtype=torch.float32)) / keep
        return s[0], s[1]


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, H, W):
        B, Ltok, C = x.shape
        if not (x.is_cuda and C % 4 == 0 and 4 * C <= 3072 and x.dtype in (torch.float32, torch.bfloat16)):
            raise DgxError("PatchMerging: GPU f32 / bf16 input with 4 C <= 3072 required (C %d, %s, %s)" % (C, x.dtype, x.device))
        # pad + 2x2 gather + LayerNorm in one pass, bf16 out = the reduction GEMM's operand
        return self.reduction(patch_merge_layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps, B, H, W))


# The blocks of a stage replayed as hipGraph pairs (round 6).  A Swin block is ~10 launches forward and ~14 backward through one autograd
# node; 24 blocks are ~560 of the step's ~820 launches and ~14 ms of host time -- the Swin-T step is host-bound outright (14.4 ms against
# ~9 of GPU work) and the Swin-L step is host-PACED (host issue = step time within 0.3 ms), so anything that takes CPU from the training
# thread (the loader's pin thread and workers) lengthens it.  The blocks' shapes depend only on (B, H, W): groups of <= GRAPH_GROUP blocks
# (+ the stage's PatchMerging) are captured with utils.graphs.GraphedSegment like the FPN and the CenterNet tower; the DropPath factors of
# a step are a graph INPUT.  Several groups per stage rather than one graph for the backbone: a group's parameters are signalled to the
# data-parallel reducer behind ITS backward replay, so the all-reduce still overlaps the rest of backward.
GRAPH_BLOCKS = True
GRAPH_GROUP = 6
# ... but only where the host is the limit.  A block is ~24 launches = ~0.5 ms of host time forward + backward; its GPU time is
# 72 T C^2 FLOP (12 C^2 per token forward, x 3) at ~0.7 PFLOP/s.  Swin-L at 1024^2 (348 GFLOP per block in every stage) keeps the GPU
# busy for as long as the host needs to issue it, and there the replayed groups were SLOWER in a same-box A/B (26.8 -> 28.0 ms/step,
# profiles/r06_ab_compact_blockgraphs.txt: the replay's node-to-node gaps and the lost first-writer protocol -- a replayed group
# accumulates into zero-filled segments); Swin-T at 1024^2 (87 GFLOP per block) went 14.4 -> 12.3 ms/step.  Groups are replayed when a
# block is below this much work, issued eagerly above it.
GRAPH_BLOCKS_MAX_GFLOP = 150.0


class _BlockGroup(nn.Module):
    """blocks[a:b] of a BasicLayer (+ its downsample when it ends the stage), tensors in / tuple out: the captured unit."""

    def __init__(self, layer, a, b, with_down):
        super().__init__()
        self.__dict__["layer"] = layer          # not a registered child: the BasicLayer owns the parameters
        self.a, self.b, self.with_down, self.amp = a, b, with_down, False
        self.hw = None

    def parameters(self, recurse=True):
        layer = self.__dict__["layer"]
        for blk in layer.blocks[self.a:self.b]:
            yield from blk.parameters()
        if self.with_down:
            yield from layer.downsample.parameters()

    def forward(self, x, scales):
        layer = self.__dict__["layer"]
        H, W = self.hw
        region = layer._region(H, W, x.device)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp, cache_enabled=False):
            for k, blk in enumerate(layer.blocks[self.a:self.b]):
                blk.H, blk.W = H, W
                blk.__dict__["_dp_scales"] = scales[k]        # (2, B): this step's DropPath factors (ignored by a block without DropPath)
                x = blk(x, region)
            if self.with_down:
                return x, layer.downsample(x, H, W)
        return (x,)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_path=0.0, downsample=None, use_checkpoint=False):
        super().__init__()
        self.window_size, self.shift_size, self.depth, self.use_checkpoint = window_size, window_size // 2, depth, use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio,
                                 qkv_bias, qk_scale, drop_path[i] if isinstance(drop_path, list) else drop_path)
            for i in range(depth)])
        self.downsample = downsample(dim=dim) if downsample is not None else None
        self._regions = {}

    def _region(self, H, W, device):
        key = (H, W, str(device))
        if key not in self._regions:
            self._regions[key] = shift_regions(H, W, self.window_size).to(device)
        return self._regions[key]

    def _groups(self):
        g = self.__dict__.get("_graph_groups")
        if g is None:
            from ...utils.graphs import GraphedSegment
            cuts = list(range(0, self.depth, GRAPH_GROUP)) + [self.depth]
            g = self.__dict__["_graph_groups"] = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                mod = _BlockGroup(self, a, b, self.downsample is not None and b == self.depth)
                g.append((mod, GraphedSegment(mod)))
        return g

    def _graphable(self, x):
        if not (GRAPH_BLOCKS and _FUSED_BLOCK and self.training and torch.is_grad_enabled() and x.is_cuda and x.requires_grad
                and not self.use_checkpoint):
            return False
        sc = self.__dict__.get("_dp_stage")
        if sc is None or sc.shape[2] != x.shape[0] or sc.device != x.device:
            return False
        if 72.0 * x.shape[0] * x.shape[1] * x.shape[2] * x.shape[2] > GRAPH_BLOCKS_MAX_GFLOP * 1e9:      # the GPU is the limit here
            return False
        if not self.__dict__.get("_arena_ok"):          # (checked until it holds: the optimizer builds the arena after the model)
            params = [p for blk in self.blocks for n, p in blk.named_parameters()] + \
                ([p for p in self.downsample.parameters()] if self.downsample is not None else [])
            if not arena_resident(tuple(params)):
                return False
            self.__dict__["_arena_ok"] = True
        return True

    def forward(self, x, H, W):
        if self._graphable(x):
            scales, down = self.__dict__["_dp_stage"], None           # (depth, 2, B) this step's DropPath factors of the stage
            for mod, seg in self._groups():
                sc = scales[mod.a:mod.b]
                mod.hw = (H, W)
                if seg.usable((x, sc), tag=(H, W)):
                    out = seg(x, sc, tag=(H, W))
                else:
                    out = mod(x, sc)
                x = out[0]
                if mod.with_down:
                    down = out[1]
            if self.downsample is not None:
                return x, H, W, down, (H + 1) // 2, (W + 1) // 2
            return x, H, W, x, H, W
        for _, seg in self.__dict__.get("_graph_groups") or ():      # issued eagerly this time: the reducer's signal counts depend on it
            linear_ops.SEGMENT_MODES[id(seg)] = "e"
        region = self._region(H, W, x.device)
        for blk in self.blocks:
            blk.H, blk.W = H, W
            if self.use_checkpoint:
                x = torch.utils.checkpoint.checkpoint(blk, x, region, use_reentrant=False)
            else:
                x = blk(x, region)
        if self.downsample is not None:
            return x, H, W, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
        return x, H, W, x, H, W


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, patch_norm=True):
        super().__init__()
        self.patch_size, self.embed_dim = (patch_size, patch_size), embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim) if patch_norm else None

    def forward(self, x):
        if isinstance(x, PatchRows):       # normalised + padded + unfolded by dgx_preprocess_patches: only the projection is left
            x, Wh, Ww = patch_embed_rows(x, self.proj.weight, self.proj.bias)
            return self._norm(x), Wh, Ww
        _, _, H, W = x.shape
        p = self.patch_size[0]
        if W % p:
            x = F.pad(x, (0, p - W % p))
        if H % p:
            x = F.pad(x, (0, 0, 0, p - H % p))
        x, Wh, Ww = patch_embed4x4(x, self.proj.weight, self.proj.bias, p)
        return self._norm(x), Wh, Ww

    def _norm(self, x):
        if self.norm is not None:
            if not (x.is_cuda and self.embed_dim % 4 == 0 and self.embed_dim <= 768 and x.dtype in (torch.float32, torch.bfloat16)):
                raise DgxError("PatchEmbed.norm: GPU f32 / bf16 input with embed_dim <= 768 required (%d, %s, %s)"
                               % (self.embed_dim, x.dtype, x.device))
            x = layernorm_f32out(x, self.norm.weight, self.norm.bias, self.norm.eps)     # fp32 out: the stage-0 residual stream
        return x


class Backbone(nn.Module):
    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    @property
    def size_divisibility(self):
        return 0


class SwinTransformer(Backbone):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24),
                 window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_path_rate=0.2, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False):
        super().__init__()
        self.num_layers, self.embed_dim, self.out_indices, self.frozen_stages = len(depths), embed_dim, tuple(out_indices), frozen_stages
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, patch_norm)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                downsample=PatchMerging if i < self.num_layers - 1 else None, use_checkpoint=use_checkpoint))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in self.out_indices:
            self.add_module("norm%d" % i, nn.LayerNorm(self.num_features[i]))
        self._out_features = ["swin%d" % i for i in self.out_indices]
        self._out_feature_channels = {"swin%d" % i: embed_dim * 2 ** i for i in self.out_indices}
        self._out_feature_strides = {"swin%d" % i: 2 ** (i + 2) for i in self.out_indices}

    def init_weights(self, pretrained=None):
        def _init(m):
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self.apply(_init)

    def _draw_drop_path(self, B, device):
        """All blocks' DropPath factors floor(keep_i + U)/keep_i in one RNG call (4 tiny kernels per step instead
        of 4 per block); same per-sample Bernoulli(keep_i) law as timm's drop_path at each of the 2 call sites."""
        blocks = [blk for layer in self.layers for blk in layer.blocks]
        probs = [blk.drop_path.drop_prob if isinstance(blk.drop_path, DropPath) else 0.0 for blk in blocks]
        if not self.training:
            return
        if not any(probs):                             # no DropPath anywhere: the graphed block groups still take a factor tensor (ones)
            ones = self.__dict__.get("_dp_ones")
            if ones is None or ones.shape[2] != B or ones.device != device:
                ones = self.__dict__["_dp_ones"] = torch.ones(len(blocks), 2, B, device=device, dtype=torch.float32)
            i0 = 0
            for layer in self.layers:
                layer.__dict__["_dp_stage"] = ones[i0:i0 + layer.depth]
                i0 += layer.depth
            return
        keep = getattr(self, "_dp_keep", None)
        if keep is None or keep.device != device:
            keep = self._dp_keep = (1.0 - torch.tensor(probs, dtype=torch.float32, device=device)).view(-1, 1, 1)
        s = torch.floor(keep + torch.rand(len(blocks), 2, B, device=device, dtype=torch.float32)) / keep
        for i, blk in enumerate(blocks):
            if probs[i] > 0:
                blk.__dict__["_dp_scales"] = s[i]
        i0 = 0
        for layer in self.layers:                      # the stage's slice: the input of its graphed block groups
            layer.__dict__["_dp_stage"] = s[i0:i0 + layer.depth]
            i0 += layer.depth

    def forward(self, x):
        self._draw_drop_path(x.shape[0], x.device)
        x, Wh, Ww = self.patch_embed(x)
        outs = {}
        for i, layer in enumerate(self.layers):
            x_out, H, W, x, Wh, Ww = layer(x, Wh, Ww)
            if i in self.out_indices:
                nm = getattr(self, "norm%d" % i)
                if not (x_out.is_cuda and self.num_features[i] <= 1536 and self.num_features[i] % 4 == 0):
                    raise DgxError("SwinTransformer out-norm: GPU input with C <= 1536 required (C %d, %s)" % (self.num_features[i], x_out.device))
                y = layernorm_bf16(x_out, nm.weight, nm.bias, nm.eps)      # fused LN -> bf16 (the FPN laterals' operand)
                # NHWC in memory, NCHW as a logical view: the registry contract sees (B,C,H,W)
                outs["swin%d" % i] = y.view(-1, H, W, self.num_features[i]).permute(0, 3, 1, 2)
        return outs


size2config = {
    "T": dict(window_size=7, embed_dim=96, depth=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], drop_path_rate=0.2),
    "S": dict(window_size=7, embed_dim=96, depth=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], drop_path_rate=0.2),
    "B": dict(window_size=7, embed_dim=128, depth=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=0.3),
    "B-22k": dict(window_size=7, embed_dim=128, depth=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=0.3),
    "B-22k-384": dict(window_size=12, embed_dim=128, depth=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=0.3),
    "L-22k": dict(window_size=7, embed_dim=192, depth=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], drop_path_rate=0.3),
    "L-22k-384": dict(window_size=12, embed_dim=192, depth=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], drop_path_rate=0.3),
}


@BACKBONE_REGISTRY.register()
def build_swintransformer_backbone(cfg, input_shape):
    c = size2config[cfg.MODEL.SWIN.SIZE]
    model = SwinTransformer(embed_dim=c["embed_dim"], window_size=c["window_size"], depths=c["depth"],
                            num_heads=c["num_heads"], drop_path_rate=c["drop_path_rate"],
                            out_indices=cfg.MODEL.SWIN.OUT_FEATURES, frozen_stages=-1,
                            use_checkpoint=cfg.MODEL.SWIN.USE_CHECKPOINT)
    model.init_weights(None)
    return model


@BACKBONE_REGISTRY.register()
def build_swintransformer_fpn_backbone(cfg, input_shape):
    from .fpn import FPN, LastLevelP6P7_P5
    bottom_up = build_swintransformer_backbone(cfg, input_shape)
    oc = cfg.MODEL.FPN.OUT_CHANNELS
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=oc, norm=cfg.MODEL.FPN.NORM,
               top_block=LastLevelP6P7_P5(oc, oc), fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
