"""The reference-owned arithmetic of the generation -> filter factory: inter-image similarity filtering of generated
instances against real crops of the same category.

Mirrors DG/filteration/get_image_similarity_from_feature.py:63-78,225-254 (for every real crop, `torch.cosine_similarity` with
every generated image of the category) and DG/filteration/filter_image_by_similarity.py:139-212 (per generated image the MEAN over
the real crops; kept when mean >= threshold).  The reference loops crop by crop on the host; here one category is ONE libdgx GEMM
over the row-normalised bf16 features (dgx_gemm_bf16_nt), the mean and the threshold follow on the device.  What is NOT here:
the feature extractors and generators themselves (CLIP / DINOv2 / Stable Diffusion / DeepFloyd / SAM are third-party models that
are not in the reference tree) and the directory / csv shuffling of the scripts.  Categories are sharded over ranks like the
reference's `--dist` loops (every rank takes categories rank::world)."""
import torch

from ..layers.gemm_ops import gemm_nt

BF16 = torch.bfloat16


def cosine_similarity_matrix(real, gen, eps=1e-8):
    """(R, D), (G, D) fp32 feature rows -> (R, G) fp32 cosine similarities, torch.cosine_similarity's definition
    (x . y / max(|x| |y|, eps)); the contraction runs on the MFMA GEMM over bf16 copies of the unit rows."""
    rn = real.float() / real.float().norm(dim=1, keepdim=True).clamp_min(eps ** 0.5)
    gn = gen.float() / gen.float().norm(dim=1, keepdim=True).clamp_min(eps ** 0.5)
    D = rn.shape[1]
    pad = (-D) % 8                                  # the GEMM wants K and N in multiples of 8
    G = gn.shape[0]
    gpad = (-G) % 8
    if pad:
        rn, gn = torch.nn.functional.pad(rn, (0, pad)), torch.nn.functional.pad(gn, (0, pad))
    if gpad:
        gn = torch.nn.functional.pad(gn, (0, 0, 0, gpad))
    return gemm_nt(rn.to(BF16).contiguous(), gn.to(BF16).contiguous()).float()[:, :G]


def mean_similarity(real, gen):
    """filter_image_by_similarity.py:18-41 (`filename_dict_to_csv`): per generated image the average similarity over the real crops."""
    return cosine_similarity_matrix(real, gen).mean(dim=0)


def filter_category(real, gen, threshold):
    """-> (keep mask (G,) bool, mean similarity (G,)): a generated image stays when its mean similarity is >= threshold (:205)."""
    sim = mean_similarity(real, gen)
    return sim >= threshold, sim


def filter_pool(features_real, features_gen, threshold, rank=0, world=1):
    """{category: (R, D)}, {category: (names, (G, D))} -> {category: {name: similarity}} of the kept images for the categories of
    this rank (sorted category order, rank::world)."""
    out = {}
    cats = sorted(set(features_real) & set(features_gen))
    for c in cats[rank::world]:
        names, gen = features_gen[c]
        if len(names) == 0 or features_real[c].shape[0] == 0:
            out[c] = {}
            continue
        keep, sim = filter_category(features_real[c], gen, threshold)
        k, s = keep.cpu().tolist(), sim.cpu().tolist()
        out[c] = {n: v for n, v, kk in zip(names, s, k) if kk}
    return out
