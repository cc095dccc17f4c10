"""The reference-owned glue of the generation -> mask -> filter factory (BASELINE configuration 5).

What the reference's scripts do AROUND the third-party networks (DeepFloyd-IF / Stable Diffusion, SAM, CLIP -- none of them in the
reference tree, none of them built here: every function below takes the network as a callable):

  generation_plan            DG/generation/txt2img_diffusers_stages_from_txt.py:123-131,213-262: the per-rank batch arithmetic, prompt
                             order and file numbering of the rank-sharded generation loop (seed = args.seed + rank, :198)
  background_corner_points   DG/segmentation/get_background_sam_mask.py:150-161: the four background prompt points
  background_mask_from_sam   :166-170: third SAM proposal, inverted, as an 8-bit mask
  check_point_in_foreground  :28-30
  shard_indices              the `i % world_size == global_rank` sharding every script of the factory uses
  masked_image_and_area      DG/filteration/get_clip_score.py:133-145: foreground kept, background set to 1, mask area fraction
  clip_preprocess            :75-81 (torchvision Resize(224, bicubic) + CenterCrop + ToTensor + Normalize restated over PIL / torch)
  clip_scores_for_category   :113-166: batched scoring of this rank's images with the prompt 'a photo of a single <name>'
  gather_by_index            :170-204: all_gather of (indices, scores, areas) over the ranks, re-ordered by image index
  select_pool_entries        DG/filteration/clean_pool_if.py:157-213: per image the segmentation method with the best CLIP score, kept when
                             the score clears min(min_clip, best - tolerance) and the area lies in [min_area, max_area]
  largest_component_filled   :34-45 (cv2.findContours / contourArea / fillPoly restated over scipy.ndimage: cv2 is not in this image)
  crop_instance              :48-86 (`subwork`): alpha > 128, largest component, tight crop, alpha masked
  merge_inst_pools           DG/tools/merge_inst_pool_json.py:60-82

Pinning (tests/test_host_factory.py): `clip_scores_for_category` / `masked_image_and_area` / `merge_inst_pools` on goldens produced by
RUNNING the reference's own scripts on a synthetic tree with stub networks (tests/golden/make_golden_factory.py).  clean_pool_if.py and
get_background_sam_mask.py cannot run as shipped -- they read `args.enable_split` / `args.in_npy_dir`, which their parsers do not
define, and need cv2 / segment_anything -- so `select_pool_entries`, `crop_instance`, `largest_component_filled`, the SAM helpers and
`generation_plan` are PARITY UNPINNED by reference runs and held on hand-worked cases."""
import os

import numpy as np
import torch
import torch.distributed as dist

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ------------------------------------------------------------------------------------------------------------------ sharding / generation
def shard_indices(n, rank, world):
    return [i for i in range(n) if i % world == rank]


def generation_plan(prompts, category_id, n_samples, max_batch_size, rank=0, world=1, offset=0):
    """-> [(prompt, images in this call, [file names])] for one category's prompt file on one rank, in call order.
    txt2img_diffusers_stages_from_txt.py: total_batch_size = n_samples // world images per prompt and rank (:123-124), cut into
    ceil(total / max_batch_size) calls (:126-131) the FIRST of which takes the remainder (:229-236); the prompt list is repeated once
    per call and sorted so that equal prompts are neighbours (:222-225); file number = j + tmp + total * rank + offset +
    (i // calls) * n_samples (:251), `tmp` being the images written so far -- it restarts at every prompt's first call (:230)."""
    total = n_samples // world
    if total * world != n_samples:
        raise ValueError("n_samples must be divisible by world_size")
    calls, rem = divmod(total, max_batch_size)
    if rem > 0:
        calls += 1
    data = sorted(calls * list(prompts))
    plan, tmp = [], 0
    for i, prompt in enumerate(data):
        prompt = prompt.strip()
        if i % calls == 0:
            tmp = 0
            n = rem if rem != 0 else max_batch_size
        else:
            n = max_batch_size
        names = ["%s_%07d.png" % (category_id, j + tmp + total * rank + offset + (i // calls) * n_samples) for j in range(n)]
        plan.append((prompt, n, names))
        tmp += n
    return plan


def rank_seed(seed, rank):
    return seed + rank


# ------------------------------------------------------------------------------------------------------------------ SAM prompting
def check_point_in_foreground(coord, atten_map, threshold):
    x, y = coord
    return atten_map[x, y] > threshold


def background_corner_points(height, width, margin=5):
    """The four positive prompt points of --background_mode (SAM then segments the BACKGROUND, which is inverted afterwards).  As the
    reference writes them (the second point's first coordinate is 0, not the margin)."""
    pts = [[margin, margin], [0, width - 1 - margin], [height - 1 - margin, margin], [height - 1 - margin, width - 1 - margin]]
    return np.array(pts), np.array([1, 1, 1, 1])


def background_mask_from_sam(masks):
    """masks: SAM's three proposals (3, H, W) bool -> uint8 (H, W) in {0, 255}: the complement of the third (largest) proposal."""
    return (1 - np.asarray(masks[2]).astype(np.uint8)) * 255


# ------------------------------------------------------------------------------------------------------------------ CLIP score
def prompt_text(category_name):
    return "a photo of a single {}".format(" ".join(category_name.split("_")))


def masked_image_and_area(image, mask):
    """image (H, W, 3) uint8, mask (H, W) uint8 -> (uint8 image with the background set to 1, foreground fraction)."""
    m = np.expand_dims(np.asarray(mask), axis=2) > 128
    out = np.asarray(image) * m + np.ones_like(np.asarray(image)) * (1 - m)
    return out.astype(np.uint8), float(np.sum(m) / m.shape[0] / m.shape[1])


def clip_preprocess(pil_image, n_px=224):
    """Resize(n_px, bicubic) -> CenterCrop(n_px) -> ToTensor -> Normalize(CLIP statistics): (3, n_px, n_px) float32."""
    from PIL import Image
    w, h = pil_image.size
    if w <= h:
        nw, nh = n_px, int(n_px * h / w)
    else:
        nw, nh = int(n_px * w / h), n_px
    im = pil_image.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))
    im = im.crop((left, top, left + n_px, top + n_px))
    x = torch.from_numpy(np.asarray(im.convert("RGB"), dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
    mean, std = torch.tensor(CLIP_MEAN).view(3, 1, 1), torch.tensor(CLIP_STD).view(3, 1, 1)
    return (x - mean) / std


def clip_scores_for_category(sample_paths, category_name, score_fn, preprocess, max_batch_size=1, rank=0, world=1, mask_paths=None):
    """This rank's share of one category.  score_fn(batch (B, 3, H, W), text) -> B scores (CLIP's logits_per_text for the one prompt).
    -> (indices, scores, areas or None) in the order the images were scored.  A batch is flushed when it is full or at the rank's last
    image (get_clip_score.py:150-166)."""
    from PIL import Image
    picked = [(i, p) for i, p in enumerate(sample_paths) if i % world == rank]
    indices, clips, areas, batch = [], [], ([] if mask_paths is not None else None), []
    text = prompt_text(category_name)
    for i, path in picked:
        image = Image.open(path).convert("RGB")
        if mask_paths is not None:
            arr, area = masked_image_and_area(np.asarray(image), np.asarray(Image.open(mask_paths[i]).convert("L")))
            image = Image.fromarray(arr)
            areas.append(area)
        batch.append(preprocess(image))
        indices.append(i)
        if len(batch) < max_batch_size and i != picked[-1][0]:
            continue
        clips.extend(float(v) for v in torch.as_tensor(score_fn(torch.stack(batch, 0), text)).reshape(-1).tolist())
        batch = []
    return indices, clips, areas


def gather_by_index(indices, *values, device="cpu", group=None):
    """Every rank's (indices, values...) -> the values of ALL ranks ordered by index (lists).  One all_gather per tensor, as the
    reference does; ranks may hold different counts here (the reference assumes equal ones): the counts are exchanged first and the
    tensors padded to the longest."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        order = np.argsort(np.asarray(indices), kind="stable")
        return [[v[j] for j in order] for v in values]
    world = dist.get_world_size(group)
    n = torch.tensor([len(indices)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c) for c in counts]
    top = max(counts)

    def gather(vals, dtype):
        t = torch.zeros(top, dtype=dtype, device=device)
        t[:len(vals)] = torch.tensor(vals, dtype=dtype, device=device)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        return torch.cat([p[:c] for p, c in zip(parts, counts)])
    idx = gather(indices, torch.int64)
    _, order = idx.sort()
    return [gather(v, torch.float32)[order].tolist() for v in values]


# ------------------------------------------------------------------------------------------------------------------ pool cleaning
def select_pool_entries(results_by_method, seg_methods, image_dir, input_dir, stage, min_clip=0.0, min_area=0.0, max_area=1.0,
                        tolerance=1.0, keep_names=None):
    """results_by_method: one results.json list (categories with 'id', 'name', 'image_count', 'clip_scores', 'areas') per segmentation
    method.  -> {0-based category id: ['image path|mask path', ...]}.  keep_names: {category name: set of file names} from the
    similarity filter's csv (data/filtration.py), or None."""
    results = [sorted(r, key=lambda x: x["image_count"]) for r in results_by_method]
    out = {}
    for c in zip(*results):
        ids = [j["id"] for j in c]
        if ids.count(ids[0]) != len(ids):
            raise ValueError("id not match, {}".format(ids))
        npc = np.stack([np.array(j["clip_scores"]) for j in c], 0)
        areas = np.stack([np.array(j["areas"]) for j in c], 0)
        if npc.size == 0 or areas.size == 0:
            continue
        name, cid = c[0]["name"], c[0]["id"] - 1
        best = np.argmax(npc, 0)
        bar = min(min_clip, np.max(npc) - tolerance)
        for k in range(len(best)):
            fname = "%s_%07d" % (c[0]["id"], k)
            if keep_names is not None and (name not in keep_names or fname + ".png" not in keep_names[name]):
                continue
            m = best[k]
            if npc[m, k] < bar or areas[m, k] < min_area or areas[m, k] > max_area:
                continue
            out.setdefault(cid, []).append("|".join([os.path.join(image_dir, stage, name, fname + ".png"),
                                                     os.path.join(input_dir, stage, seg_methods[m], name, fname + ".png")]))
    return out


def largest_component_filled(mask):
    """uint8 (H, W) or (H, W, 1) in {0, 1} -> the connected component (8-connectivity) whose OUTER contour encloses the largest area,
    holes filled -- cv2.findContours(RETR_EXTERNAL) + max contourArea + fillPoly.  contourArea is the polygon area through the boundary
    pixel centres; it is restated as (filled pixel count - boundary pixel count / 2 - 1) (Pick's theorem) so that the choice between
    components follows the same measure."""
    from scipy import ndimage
    m = np.asarray(mask)
    shape = m.shape
    m2 = m.reshape(shape[0], shape[1]) > 0
    lab, n = ndimage.label(m2, structure=np.ones((3, 3), dtype=bool))
    if n == 0:
        return m
    best, best_area = None, -1.0
    for i in range(1, n + 1):
        filled = ndimage.binary_fill_holes(lab == i)
        inner = ndimage.binary_erosion(filled, structure=np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool), border_value=0)
        boundary = int(filled.sum() - inner.sum())
        area = float(filled.sum()) - boundary / 2.0 - 1.0
        if area > best_area:
            best, best_area = filled, area
    return best.astype(m.dtype).reshape(shape)


def crop_instance(rgba, mask=None):
    """(H, W, 4) uint8 (+ optional (H, W) uint8 alpha replacement) -> the cropped RGBA instance, or None when nothing is left."""
    img = np.array(rgba, dtype=np.uint8, copy=True)
    if mask is not None:
        img[:, :, -1] = np.asarray(mask)
    seg = (img[..., 3:] > 128).astype("uint8")
    seg = largest_component_filled(seg)
    if seg.size == 0:
        return None
    ys, xs = np.where(seg[..., 0])
    if ys.size == 0 or xs.size == 0:
        return None
    y0, y1, x0, x1 = ys.min(), ys.max(), xs.min(), xs.max()
    if y1 <= y0 or x1 <= x0:
        return None
    img[:, :, 3:] *= seg
    return img[y0:y1 + 1, x0:x1 + 1]


def merge_inst_pools(pools, before_prefix=None, after_prefix=None):
    """[{key: [paths]}, ...] -> one pool; with prefixes, every path of pool i has before_prefix[i] replaced by after_prefix[i]."""
    if before_prefix is not None and not (len(pools) == len(before_prefix) == len(after_prefix)):
        raise ValueError("length of inst_pool_path and before_prefix / after_prefix should be equal")
    out = {}
    for i, pool in enumerate(pools):
        for key, value in pool.items():
            if before_prefix is not None:
                value = [v.replace(before_prefix[i], after_prefix[i]) for v in value]
            if key in out:
                out[key].extend(value)
            else:
                out[key] = list(value)
    return out
