"""Deterministic input/parameter recipes shared by make_golden.py (authoring) and the tests."""
import torch


def fill_state(shapes, seed, scale=0.05):
    """shapes: ordered list of (name, shape) in named_parameters() order.  Mirrors
    tests/golden/make_golden.py:fill_params."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in shapes:
        v = torch.randn(shape, generator=g) * scale
        if name.endswith("norm.weight") or (".norm" in name and name.endswith("weight")) \
                or (name.startswith("norm") and name.endswith("weight")):
            v = v + 1.0
        out[name] = v
    return out


def swin_param_shapes(embed_dim, depths, num_heads, ws, out_indices=(1, 2, 3), mlp_ratio=4):
    """named_parameters() order of the reference SwinTransformer (swintransformer.py:473-557)."""
    s = [("patch_embed.proj.weight", (embed_dim, 3, 4, 4)), ("patch_embed.proj.bias", (embed_dim,)),
         ("patch_embed.norm.weight", (embed_dim,)), ("patch_embed.norm.bias", (embed_dim,))]
    for i, d in enumerate(depths):
        C = embed_dim * 2 ** i
        for j in range(d):
            p = "layers.%d.blocks.%d." % (i, j)
            s += [(p + "norm1.weight", (C,)), (p + "norm1.bias", (C,)),
                  (p + "attn.relative_position_bias_table", ((2 * ws - 1) ** 2, num_heads[i])),
                  (p + "attn.qkv.weight", (3 * C, C)), (p + "attn.qkv.bias", (3 * C,)),
                  (p + "attn.proj.weight", (C, C)), (p + "attn.proj.bias", (C,)),
                  (p + "norm2.weight", (C,)), (p + "norm2.bias", (C,)),
                  (p + "mlp.fc1.weight", (mlp_ratio * C, C)), (p + "mlp.fc1.bias", (mlp_ratio * C,)),
                  (p + "mlp.fc2.weight", (C, mlp_ratio * C)), (p + "mlp.fc2.bias", (C,))]
        if i < len(depths) - 1:
            p = "layers.%d.downsample." % i
            s += [(p + "reduction.weight", (2 * C, 4 * C)), (p + "norm.weight", (4 * C,)), (p + "norm.bias", (4 * C,))]
    for i in out_indices:
        s += [("norm%d.weight" % i, (embed_dim * 2 ** i,)), ("norm%d.bias" % i, (embed_dim * 2 ** i,))]
    return s
