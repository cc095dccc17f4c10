"""CPU: the worker half of the instance copy-paste (InstPool.draw / prepare) against tests/golden/pool_draws.npz -- the
reference's own InstPool.get_mix_result('cas_random') run end to end on the PNG fixtures (make_golden.py::gen_pool_draws).
Pinned: the ORDER of the np.random draws (sample count, class / instance pairs, every _load_RGBA before any placement), the
resize requests, the placements, the stream position afterwards; and, with the oracle compositor fed those pastes, the final
image / boxes / classes / masks / instance_source.  The GPU twin is tests/test_gpu_loader.py."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def make_pool(z, loader=None):
    from divergen_amd.data.copypaste import InstPool
    keys = [str(k) for k in np.load(os.path.join(GOLD, "pool_decode.npz"))["keys"]]
    pool = {}
    for k, c in zip(keys, z["pool_cats"].tolist()):
        pool.setdefault(str(c), []).append(k)
    ip = InstPool(pool, tuple(int(v) for v in z["hw"]), max_samples=int(z["max_samples"]), random_scale=False,
                  random_scale_min=0.5, random_scale_max=2.0, random_scale_min_size=5, use_largest_part=False, loader=loader)
    ip.HWms = {str(k): [float(a), float(b)] for k, (a, b) in zip(z["HWms_keys"], z["HWms_vals"])}
    return ip


def cases(z):
    ci = 0
    while "c%d_seed" % ci in z.files:
        yield ci
        ci += 1


@pytest.fixture()
def in_golden_dir():
    cwd = os.getcwd()
    os.chdir(GOLD)          # pool keys are relative to tests/golden/
    yield
    os.chdir(cwd)


def test_draw_order_and_values_equal_the_reference(in_golden_dir, monkeypatch):
    from divergen_amd.data.copypaste import InstPool
    z = np.load(os.path.join(GOLD, "pool_draws.npz"))
    ip = make_pool(z)
    H, W = (int(v) for v in z["hw"])
    seen = []
    real = InstPool._resize

    def rec(rgba, tw, th):
        seen.append((rgba.shape[0], rgba.shape[1], tw, th))
        return real(rgba, tw, th)
    monkeypatch.setattr(InstPool, "_resize", staticmethod(rec))
    total = 0
    for ci in cases(z):
        del seen[:]
        np.random.seed(int(z["c%d_seed" % ci]))
        pastes, names = ip.draw((H, W))
        after = np.random.randint(0, 2 ** 31 - 1)
        assert np.array_equal(np.array(seen, dtype=np.int64).reshape(-1, 4), z["c%d_resize" % ci]), ci
        got = np.array([[x0, y0, lab] for _, x0, y0, lab in pastes], dtype=np.int64).reshape(-1, 3)
        assert np.array_equal(got, z["c%d_place" % ci]), ci
        assert after == int(z["c%d_after" % ci]), "case %d leaves np.random at another position than the reference" % ci
        assert len(names) == len(pastes)
        total += len(pastes)
    assert total >= 12          # the fixture exercises real pastes, both size branches and rejections


def test_prepare_plus_oracle_compositor_equals_reference_mix_result(in_golden_dir):
    """Worker half (prepare) -> the packed form -> unpacked again -> oracle compositor == the reference's get_mix_result."""
    from oracle import compositor as OK
    from divergen_amd.structures import BitMasks, Boxes, Instances
    z = np.load(os.path.join(GOLD, "pool_draws.npz"))
    ip = make_pool(z)
    H, W = (int(v) for v in z["hw"])
    for ci in cases(z):
        inst = Instances((H, W), gt_boxes=Boxes(torch.from_numpy(z["c%d_boxes" % ci])), gt_classes=torch.from_numpy(z["c%d_labels" % ci]),
                         gt_masks=BitMasks(torch.from_numpy(z["c%d_masks" % ci])))
        np.random.seed(int(z["c%d_seed" % ci]))
        d = ip.prepare({"image": torch.from_numpy(z["c%d_image" % ci]), "instances": inst, "file_name": "case%d" % ci})
        pk = d["paste_pack"]
        assert pk["flat"].dtype == torch.uint8 and pk["desc"].dtype == torch.int32 and pk["labels"].dtype == torch.int64
        assert pk["desc"].shape == (pk["K"], 5) and d["paste_labels"] == pk["labels"].tolist() and len(d["paste_filename_list"]) == pk["K"]
        flat = pk["flat"].numpy()
        pastes = [(flat[o:o + h * w * 4].reshape(h, w, 4), x0, y0, int(lab))
                  for (o, h, w, x0, y0), lab in zip(pk["desc"].tolist(), pk["labels"].tolist())]
        ref = OK.composite(z["c%d_image" % ci], z["c%d_masks" % ci], z["c%d_boxes" % ci], z["c%d_labels" % ci], pastes)
        assert np.array_equal(ref["image"], z["c%d_out_image" % ci]), ci
        assert np.array_equal(ref["boxes"], z["c%d_out_boxes" % ci]) and np.array_equal(ref["labels"], z["c%d_out_labels" % ci])
        assert np.array_equal(ref["masks"], z["c%d_out_masks" % ci]) and np.array_equal(ref["source"], z["c%d_out_source" % ci])


def test_batch_ahead_on_cpu_passes_worker_results_through_finish():
    """BatchAhead without a GPU: no side stream, `finish` applied per sample, order kept, StopIteration passed on."""
    from divergen_amd.data.build import BatchAhead
    host = [[{"image": torch.full((3, 4, 4), i, dtype=torch.uint8)}, {"image": torch.full((3, 4, 4), 10 + i, dtype=torch.uint8)}] for i in range(3)]
    seen = []

    def finish(d, device):
        seen.append(int(d["image"][0, 0, 0]))
        return d
    out = list(BatchAhead(host, finish, "cpu"))
    assert [int(b[0]["image"][0, 0, 0]) for b in out] == [0, 1, 2] and seen == [0, 10, 1, 11, 2, 12]


class _FakeRing:
    """SlotRing without the page-locking (no GPU here): the same slot arithmetic over a shared-memory buffer."""

    def __init__(self, num_workers, per_worker, slot_bytes):
        self.num_workers, self.per_worker, self.slot_bytes = num_workers, per_worker, slot_bytes
        self.buf = torch.zeros(num_workers * per_worker * slot_bytes, dtype=torch.uint8).share_memory_()

    def offset(self, worker, k):
        return (worker * self.per_worker + k % self.per_worker) * self.slot_bytes


class _BlobDataset(torch.utils.data.Dataset):
    def __len__(self):
        return 64

    def __getitem__(self, i):
        n = 100 + 7 * (i % 5) if i % 9 else 5000          # every ninth sample does not fit a slot: it must travel the ordinary way
        return {"i": i, "blob": torch.full((n,), i % 251, dtype=torch.uint8)}


def test_ring_collate_writes_worker_slots_and_reuse_is_safe():
    """data/build.py _RingCollate + the SlotRing slot arithmetic with real worker processes: every sample's blob is found in the slot it
    reports, intact at the time the training process would upload it (batch t + 1 is pulled while batch t is in use: one ahead), with
    (prefetch_factor + 2) batches of slots per worker; a blob larger than a slot arrives as a tensor."""
    from divergen_amd.data.build import _RingCollate
    nw, pf, bs = 3, 2, 2
    ring = _FakeRing(nw, (pf + 2) * bs, 256)
    loader = torch.utils.data.DataLoader(_BlobDataset(), batch_size=bs, num_workers=nw, prefetch_factor=pf, collate_fn=_RingCollate(ring),
                                         persistent_workers=True)
    it = iter(loader)
    prev = next(it)
    seen, via_slot = 0, 0
    for _ in range(30):
        cur = next(it)                      # the loader is one batch ahead of the batch being "uploaded" ...
        for d in prev:                      # ... which must still be intact in its slot
            if d.get("blob_slot") is not None:
                off, n = d["blob_slot"]
                assert d["blob"] is None and n == 100 + 7 * (d["i"] % 5)
                assert bool((ring.buf[off:off + n] == d["i"] % 251).all()), d["i"]
                assert off % ring.slot_bytes == 0 and off // (ring.slot_bytes * ring.per_worker) < nw
                via_slot += 1
            else:
                assert d["blob"].numel() == 5000 and int(d["blob"][0]) == d["i"] % 251
            seen += 1
        prev = cur
    assert seen == 60 and via_slot >= 50
    del it, loader
