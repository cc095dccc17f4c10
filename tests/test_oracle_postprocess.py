"""Oracle for the evaluation post-processing path (oracle/postprocess.py) against the reference's own output
(tests/golden/paste_masks.npz, made by tests/golden/make_golden.py from D2/layers/mask_ops.py) and RLE known answers."""
import os

import numpy as np
import pytest

from oracle import postprocess as P

G = os.path.join(os.path.dirname(__file__), "golden")


def _golden():
    z = np.load(os.path.join(G, "paste_masks.npz"))
    H, W = [int(v) for v in z["image_shape"]]
    N = z["masks"].shape[0]
    ref = np.unpackbits(z["out_bits"], axis=1)[:, :H * W].reshape(N, H, W).astype(bool)
    return z["masks"], z["boxes"], (H, W), float(z["threshold"]), ref


def test_paste_masks_matches_reference_output():
    masks, boxes, hw, thr, ref = _golden()
    got = P.paste_masks(masks, boxes, hw, thr)
    assert ref.sum() > 1000                       # fixture is not trivial
    assert ref[4].sum() > 0 and ref[5].sum() == 0
    # bit-exact: same fp32 operation order as the reference + ATen's CPU grid_sample
    assert np.array_equal(got, ref), int((got != ref).sum())


def test_rle_known_answers():
    # 3x3 example worked by hand, column-major: columns (0,1,1), (1,0,0), (0,0,1)
    m = np.array([[0, 1, 0], [1, 0, 0], [1, 0, 1]], bool)
    assert P.rle_counts(m).tolist() == [1, 3, 4, 1]
    # mask starting with a set pixel: leading zero-length run
    m2 = np.array([[1, 0], [1, 1]], bool)
    assert P.rle_counts(m2).tolist() == [0, 2, 1, 1]
    assert P.rle_counts(np.zeros((4, 5), bool)).tolist() == [20]
    assert P.rle_counts(np.ones((4, 5), bool)).tolist() == [0, 20]
    # string form worked by hand: 1,3,4 -> '1','3','4'; 4th count is stored as 1-3 = -2 -> 5-bit 0x1E, sign bit set,
    # remaining bits all ones -> single char 0x1E + 48 = 'N'
    assert P.rle_to_string([1, 3, 4, 1]) == b"134N"
    # 100 = 0b00011_00100 -> low group 4 with continuation (0x20) -> '4'+0x20 = 'T', then 3 -> '3'
    assert P.rle_to_string([100]) == b"T3"


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_rle_round_trip(seed):
    rng = np.random.default_rng(seed)
    H, W = 37, 53
    m = rng.random((H, W)) < (0.5 if seed else 0.02)
    if seed == 2:                                  # long runs -> multi-character codes and negative deltas
        m = np.zeros((H, W), bool)
        m[5:30, 10:40] = True
    c = P.rle_counts(m)
    assert c.sum() == H * W
    assert np.array_equal(P.rle_decode(c, H, W), m)
    s = P.rle_to_string(c)
    assert np.array_equal(P.rle_from_string(s), c)


def test_rle_of_golden_masks_round_trip():
    _, _, (H, W), _, ref = _golden()
    for n in range(ref.shape[0]):
        c = P.rle_counts(ref[n])
        assert np.array_equal(P.rle_decode(c, H, W), ref[n])
        assert np.array_equal(P.rle_from_string(P.rle_to_string(c)), c)
