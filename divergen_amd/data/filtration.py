"""The reference-owned arithmetic of the generation -> filter factory: inter-image similarity filtering of generated
instances against real crops of the same category.

Mirrors DG/filteration/get_image_similarity_from_feature.py:63-78,225-254 (for every real crop, `torch.cosine_similarity` with
every generated image of the category) and DG/filteration/filter_image_by_similarity.py:139-212 (per generated image the MEAN over
the real crops; kept when mean >= threshold).  The reference loops crop by crop on the host; here one category is ONE libdgx MFMA
contraction with fp32 accumulation and fp32 output (dgx_linear_wgrad_grouped over hi / lo bf16 halves of the features), the
norms, the mean and the threshold follow on the device in fp32.  What is NOT here:
the feature extractors and generators themselves (CLIP / DINOv2 / Stable Diffusion / DeepFloyd / SAM are third-party models that
are not in the reference tree) and the directory / csv shuffling of the scripts.  Categories are sharded over ranks like the
reference's `--dist` loops (every rank takes categories rank::world)."""
import torch

from ..layers.linear_ops import wgrad_into

BF16 = torch.bfloat16


def _split_bf16(x):
    """fp32 -> (hi, lo) bf16 with hi + lo = x to ~2^-17 relative: the two-term split that lets a bf16 MFMA contraction with an fp32
    accumulator reproduce an fp32 product."""
    hi = x.to(BF16)
    return hi, (x - hi.float()).to(BF16)


def cosine_similarity_matrix(real, gen, eps=1e-8):
    """(R, D), (G, D) fp32 feature rows -> (R, G) fp32 cosine similarities, torch.cosine_similarity's definition
    (x . y / max(|x| |y|, eps)).  fp32 all the way, as the reference computes it: the dot products run on the fp32-accumulating,
    fp32-writing MFMA kernel (dgx_linear_wgrad_grouped: out = a^T b over the row index) with every operand split into hi + lo bf16
    halves (hi.hi + hi.lo + lo.hi as ONE contraction of length 3 D), so a keep / drop decision differs from the fp32 one only
    within ~1e-5 of the threshold -- not within the 4e-3 of a bf16 result."""
    real, gen = real.float(), gen.float()
    R, G, D = real.shape[0], gen.shape[0], real.shape[1]
    rp, gp = (-R) % 8, (-G) % 8                     # the kernel wants output widths in multiples of 8
    rh, rl = _split_bf16(torch.nn.functional.pad(real, (0, 0, 0, rp)).t().contiguous())     # (D, R8)
    gh, gl = _split_bf16(torch.nn.functional.pad(gen, (0, 0, 0, gp)).t().contiguous())      # (D, G8)
    dots = torch.empty(R + rp, G + gp, dtype=torch.float32, device=real.device)
    wgrad_into(dots, torch.cat([rh, rh, rl]), torch.cat([gh, gl, gh]), beta=0.0)
    norms = (real.norm(dim=1).view(-1, 1) * gen.norm(dim=1).view(1, -1)).clamp_min(eps)
    return dots[:R, :G] / norms


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
