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


def fill_by_name(module, seed, scale):
    """Mirrors tests/golden/make_golden.py:fill_by_name: parameters from a per-NAME seeded stream, bf16-exact values."""
    import zlib
    with torch.no_grad():
        for name, p in module.named_parameters():
            g = torch.Generator().manual_seed(seed + (zlib.crc32(name.encode()) & 0xFFFFFF))
            v = torch.randn(p.shape, generator=g) * scale
            if p.dim() == 1 and name.endswith("weight"):
                v = v + 1.0
            v = v.to(torch.bfloat16).float()
            if hasattr(p, "_dgx_sd_perm"):      # a parameter stored in another column order than its reference / state-dict form
                v = p._dgx_sd_perm[1](v)        # (the values are defined in the reference's order)
            p.copy_(v)


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


# ---- the two random draws of a training step replaced by a deterministic rule (shared by the e2e parity test and the
# cpu_baseline leg of bench.py, which times the assembled oracle)
def det_sample(labels, num_samples, positive_fraction, bg_label):
    pos = ((labels != -1) & (labels != bg_label)).nonzero().squeeze(1)
    neg = (labels == bg_label).nonzero().squeeze(1)
    npos = min(pos.numel(), int(num_samples * positive_fraction))
    return pos[:npos], neg[:min(neg.numel(), num_samples - npos)]


def det_fed_mask(gt_classes, K, C, weight):
    app = torch.zeros(C + 1, dtype=torch.bool, device=gt_classes.device)
    app[gt_classes] = True
    app[C] = False if not bool((gt_classes == C).any()) else True
    cand = (~app[:C]) & (weight > 0)
    extra = cand.nonzero().squeeze(1)[:max(K - int(app.sum()), 0)]
    m = app.clone()
    m[extra] = True
    return m


def assembled_oracle_losses(p, images, gts, image_sizes, swin, num_classes, freq_weight, batch_per_image=512, pos_fraction=0.25,
                            fed_num=50, mask_weight=1.0, proposals=None, score_thresh=0.0001, pre_topk=4000, nms_thresh=0.9,
                            post_topk=2000, stage_labels=None):
    """The whole training forward of oracle/model.py on CPU: Swin + FPN + CenterNet head -> CenterNet losses -> proposals
    (the oracle's own decode + NMS unless `proposals` are handed in) -> cascade RoI heads + mask head.  Returns the loss
    dict (differentiable w.r.t. the entries of `p` that require grad)."""
    from oracle import model as OM
    fp, regs, hms = OM.backbone_and_dense(p, images, swin)
    losses = dict(OM.centernet_losses(regs, hms, [g["boxes"] for g in gts]))
    if proposals is None:
        with torch.no_grad():
            proposals = [b for b, _ in (OM.proposals_from_heatmaps([r.detach() for r in regs], [h.detach() for h in hms],
                                                                   score_thresh, pre_topk, nms_thresh, post_topk))]
    losses.update(OM.roi_head_losses(p, fp, proposals, gts, image_sizes, num_classes, batch_per_image, pos_fraction, freq_weight,
                                     fed_num, lambda i, labels, n, frac, bg: det_sample(labels, n, frac, bg),
                                     lambda k, gtc, K, Cn, w: det_fed_mask(gtc, K, Cn, w).nonzero().squeeze(1),
                                     mask_weight=mask_weight, stage_labels=stage_labels))
    return losses
