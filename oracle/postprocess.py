"""CPU restatement of the evaluation post-processing path.  TEST INFRASTRUCTURE ONLY.

* paste_masks      <- D2/layers/mask_ops.py:17-68 `_do_paste_mask` + :73-150 `paste_masks_in_image` (bilinear grid_sample,
                      align_corners=False, zero padding, `>= threshold`).  Pinned: tests/golden/paste_masks.npz holds the
                      reference function's own output on seeded inputs (tests/test_oracle_postprocess.py).
* rle_counts / rle_to_string / rle_from_string
                   <- pycocotools 2.0.x `maskApi.c` rleEncode / rleToString / rleFrString, called by the reference at
                      D2/evaluation/coco_evaluation.py:406 (`mask_util.encode(np.array(mask[:, :, None], order="F"))`).
                      pycocotools is a pip dependency, NOT vendored under /root/reference and not installed in this image:
                      the published algorithm is restated here.  PARITY UNPINNED for the string form (no pycocotools to
                      compare with); the run lengths themselves are pinned by the decode(encode(m)) == m property and by the
                      hand-worked vectors in the tests.
"""
import numpy as np


def paste_masks(masks, boxes, image_shape, threshold=0.5):
    """masks (N,S,S) f32, boxes (N,4) f32 -> (N,H,W) bool.  Same fp32 operation order as the reference + ATen grid_sample
    (`grid_sampler_unnormalize`: ((g + 1) * size - 1) / 2; weights from the floor corner; out-of-range taps contribute 0)."""
    masks = np.asarray(masks, np.float32)
    boxes = np.asarray(boxes, np.float32)
    H, W = image_shape
    N, S = masks.shape[0], masks.shape[-1]
    out = np.zeros((N, H, W), bool)
    f = np.float32
    with np.errstate(all="ignore"):
        for n in range(N):
            x0, y0, x1, y1 = boxes[n]
            gx = ((np.arange(W, dtype=np.float32) + f(0.5)) - x0) / (x1 - x0) * f(2) - f(1)
            gy = ((np.arange(H, dtype=np.float32) + f(0.5)) - y0) / (y1 - y0) * f(2) - f(1)
            ix = ((gx + f(1)) * f(S) - f(1)) / f(2)
            iy = ((gy + f(1)) * f(S) - f(1)) / f(2)
            fx, fy = np.floor(ix), np.floor(iy)
            wx1, wx0 = ix - fx, (fx + f(1)) - ix            # weight of the right / left tap
            wy1, wy0 = iy - fy, (fy + f(1)) - iy
            finite_x, finite_y = np.isfinite(fx), np.isfinite(fy)
            cx = np.where(finite_x, fx, -9).astype(np.int64)
            cy = np.where(finite_y, fy, -9).astype(np.int64)

            def tap(yy, xx):
                ok = ((yy >= 0) & (yy < S))[:, None] & ((xx >= 0) & (xx < S))[None, :]
                v = masks[n][np.clip(yy, 0, S - 1)[:, None], np.clip(xx, 0, S - 1)[None, :]]
                return v, ok
            acc = np.zeros((H, W), np.float32)
            for (yy, wy) in ((cy, wy0), (cy + 1, wy1)):            # ATen order: nw, ne, sw, se
                for (xx, wx) in ((cx, wx0), (cx + 1, wx1)):
                    v, ok = tap(yy, xx)
                    w = (wx[None, :] * wy[:, None]).astype(np.float32)
                    acc = np.where(ok, acc + v * w, acc).astype(np.float32)
            out[n] = acc >= f(threshold)            # NaN compares False, like the reference
    return out


def rle_counts(mask):
    """(H,W) binary -> run lengths in column-major order, first run counts zeros (maskApi.c rleEncode)."""
    flat = np.asarray(mask, np.uint8).reshape(mask.shape[0], mask.shape[1]).T.reshape(-1)       # column-major
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(edges).astype(np.int64)
    if flat.size and flat[0] == 1:
        counts = np.concatenate([[0], counts])
    return counts


def rle_decode(counts, H, W):
    vals = np.zeros(len(counts), np.uint8)
    vals[1::2] = 1
    flat = np.repeat(vals, np.asarray(counts, np.int64))
    assert flat.size == H * W, (flat.size, H, W)
    return flat.reshape(W, H).T.astype(bool)


def rle_to_string(counts):
    """maskApi.c rleToString: counts (delta against two back from the 4th on) in a 5-bit-per-char LEB-like code + 48."""
    out = bytearray()
    counts = [int(c) for c in counts]
    for i, x in enumerate(counts):
        if i > 2:
            x -= counts[i - 2]
        more = True
        while more:
            c = x & 0x1F
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
    return bytes(out)


def rle_from_string(s):
    """maskApi.c rleFrString."""
    counts, p = [], 0
    s = bytes(s)
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return np.asarray(counts, np.int64)
