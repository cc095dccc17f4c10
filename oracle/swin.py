"""Oracle: Swin backbone on torch-CPU fp32, functional over a reference-keyed state dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
DG/divergen/modeling/backbone/swintransformer.py (DG = /root/reference/DiverGen):
  window_partition :49-60, window_reverse :63-76, WindowAttention.forward :126-157,
  relative_position_index :105-116, SwinTransformerBlock.forward :201-257,
  PatchMerging.forward :272-298, BasicLayer.forward (shift mask) :361-400,
  PatchEmbed.forward :426-442, SwinTransformer.forward :602-629.
"""
import math

import torch
import torch.nn.functional as F

from .quant import active as _bf16_on
from .quant import rb


def relative_position_index(ws):
    """(ws*ws, ws*ws) int64 index into the (2ws-1)^2 bias table.  swintransformer.py:105-116."""
    r = torch.arange(ws)
    ii, jj = torch.meshgrid(r, r, indexing="ij")
    pos = torch.stack([ii.reshape(-1), jj.reshape(-1)])  # 2, N
    rel = pos[:, :, None] - pos[:, None, :]  # 2, N, N
    return (rel[0] + ws - 1) * (2 * ws - 1) + (rel[1] + ws - 1)


def partition(x, ws):
    """(B,Hp,Wp,C) -> (B*nW, ws*ws, C).  swintransformer.py:49-60,233."""
    B, Hp, Wp, C = x.shape
    x = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws * ws, C)


def unpartition(w, ws, Hp, Wp):
    """inverse of partition.  swintransformer.py:63-76."""
    C = w.shape[-1]
    B = w.shape[0] // ((Hp // ws) * (Wp // ws))
    x = w.reshape(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, Hp, Wp, C)


def shift_mask(H, W, ws):
    """SW-MSA additive mask (nW, N, N) in {0,-100}.  swintransformer.py:368-387."""
    shift = ws // 2
    Hp = int(math.ceil(H / ws)) * ws
    Wp = int(math.ceil(W / ws)) * ws
    region = torch.zeros(Hp, Wp)
    cuts_h = [0, Hp - ws, Hp - shift, Hp]
    cuts_w = [0, Wp - ws, Wp - shift, Wp]
    cnt = 0
    for a in range(3):
        for b in range(3):
            region[cuts_h[a]:cuts_h[a + 1], cuts_w[b]:cuts_w[b + 1]] = cnt
            cnt += 1
    ids = partition(region[None, :, :, None], ws)[..., 0]  # nW, N
    diff = ids[:, None, :] - ids[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def window_attention(x, mask, p, prefix, num_heads):
    """x (B_,N,C) -> (B_,N,C).  swintransformer.py:126-157."""
    B_, N, C = x.shape
    hd = C // num_heads
    ws = int(round(math.sqrt(N)))
    qkv = rb(F.linear(x, p[prefix + "qkv.weight"], p[prefix + "qkv.bias"]))
    qkv = qkv.reshape(B_, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    s = q @ k.transpose(-2, -1)  # B_, nH, N, N
    idx = p.get(prefix + "relative_position_index")
    if idx is None:
        idx = relative_position_index(ws)
    bias = p[prefix + "relative_position_bias_table"][idx.reshape(-1).long()]
    s = s + bias.reshape(N, N, num_heads).permute(2, 0, 1)[None]
    if mask is not None:
        nW = mask.shape[0]
        s = (s.reshape(B_ // nW, nW, num_heads, N, N) + mask[None, :, None]).reshape(B_, num_heads, N, N)
    if _bf16_on():
        # the product's kernel: e = exp(s - max) in fp32, the row sum from the fp32 values, P V on the bf16-rounded e, the
        # division behind it, the output stored as bf16
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        o = rb((rb(e) @ v) / e.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(B_, N, C)
    else:
        a = torch.softmax(s, dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B_, N, C)
    return rb(F.linear(o, p[prefix + "proj.weight"], p[prefix + "proj.bias"]))


def drop_path(x, rate, training):
    """timm==0.4.9 drop_path (per-sample Bernoulli keep, scale 1/keep).  Not vendored in
    the reference; call site swintransformer.py:193,254-255.  PARITY UNPINNED by reference tests."""
    if rate == 0.0 or not training:
        return x
    keep = 1.0 - rate
    r = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype)
    return x.div(keep) * r.floor()


def swin_block(x, H, W, mask, p, prefix, num_heads, ws, shift, drop=0.0, training=False, stream=None):
    """swintransformer.py:201-257.  stream: rounding of the residual stream (quant.rb for the stages whose stream the product
    keeps in bf16; None = fp32)."""
    B, L, C = x.shape
    rs = stream if stream is not None else (lambda t: t)
    shortcut = x
    h = rb(F.layer_norm(x, (C,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"])).reshape(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    h = F.pad(h, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = H + pad_b, W + pad_r
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    a = window_attention(partition(h, ws), mask if shift > 0 else None, p, prefix + "attn.", num_heads)
    h = unpartition(a, ws, Hp, Wp)
    if shift > 0:
        h = torch.roll(h, shifts=(shift, shift), dims=(1, 2))
    h = h[:, :H, :W, :].reshape(B, H * W, C)
    x = rs(shortcut + drop_path(h, drop, training))
    m = rb(F.layer_norm(x, (C,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"]))
    m = rb(F.linear(m, p[prefix + "mlp.fc1.weight"], p[prefix + "mlp.fc1.bias"]))
    m = rb(F.gelu(m))
    m = rb(F.linear(m, p[prefix + "mlp.fc2.weight"], p[prefix + "mlp.fc2.bias"]))
    return rs(x + drop_path(m, drop, training))


def patch_merging(x, H, W, p, prefix):
    """swintransformer.py:272-298."""
    B, L, C = x.shape
    x = x.reshape(B, H, W, C)
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.reshape(B, -1, 4 * C)
    x = rb(F.layer_norm(x, (4 * C,), p[prefix + "norm.weight"], p[prefix + "norm.bias"]))
    return rb(F.linear(x, p[prefix + "reduction.weight"]))


def basic_layer(x, H, W, p, prefix, depth, num_heads, ws, downsample, drops=None, training=False, stream=None):
    """swintransformer.py:361-400."""
    mask = shift_mask(H, W, ws)
    for i in range(depth):
        x = swin_block(x, H, W, mask, p, "%sblocks.%d." % (prefix, i), num_heads, ws,
                       0 if i % 2 == 0 else ws // 2,
                       drops[i] if drops else 0.0, training, stream)
    if downsample:
        return x, patch_merging(x, H, W, p, prefix + "downsample."), (H + 1) // 2, (W + 1) // 2
    return x, x, H, W


def swin_forward(img, p, embed_dim, depths, num_heads, ws, out_indices=(1, 2, 3), prefix="",
                 drop_path_rate=0.0, training=False):
    """img (B,3,H,W) fp32 -> {'swin{i}': (B,C_i,H_i,W_i)}.  swintransformer.py:426-442,602-629."""
    _, _, H, W = img.shape
    if W % 4:
        img = F.pad(img, (0, 4 - W % 4))
    if H % 4:
        img = F.pad(img, (0, 0, 0, 4 - H % 4))
    x = rb(F.conv2d(img, p[prefix + "patch_embed.proj.weight"], p[prefix + "patch_embed.proj.bias"], stride=4))
    Wh, Ww = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (embed_dim,), p[prefix + "patch_embed.norm.weight"], p[prefix + "patch_embed.norm.bias"])
    dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
    outs = {}
    for i, depth in enumerate(depths):
        C = embed_dim * 2 ** i
        x_out, x, nWh, nWw = basic_layer(
            x, Wh, Ww, p, "%slayers.%d." % (prefix, i), depth, num_heads[i], ws,
            i < len(depths) - 1, dpr[sum(depths[:i]):sum(depths[:i + 1])], training,
            rb if i >= 1 else None)                # the product's residual stream: fp32 in stage 0, bf16 from stage 1 on
        if i in out_indices:
            y = rb(F.layer_norm(x_out, (C,), p["%snorm%d.weight" % (prefix, i)], p["%snorm%d.bias" % (prefix, i)]))
            outs["swin%d" % i] = y.reshape(-1, Wh, Ww, C).permute(0, 3, 1, 2).contiguous()
        Wh, Ww = nWh, nWw
    return outs


SIZE2CONFIG = {  # swintransformer.py:636-693 (values only)
    "T": dict(ws=7, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], drop_path_rate=0.2),
    "S": dict(ws=7, embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], drop_path_rate=0.2),
    "B": dict(ws=7, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=0.3),
    "B-22k": dict(ws=7, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=0.3),
    "B-22k-384": dict(ws=12, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=0.3),
    "L-22k": dict(ws=7, embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], drop_path_rate=0.3),
    "L-22k-384": dict(ws=12, embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], drop_path_rate=0.3),
}
