"""Instance-pool shard store on the CPU: the store returns exactly what the reference's per-sample decode
returns, and the InstPool mirror fed from it reaches cv2.resize with the array and target size the reference's
`_load_RGBA` reaches it with (tests/golden/pool_decode.npz, captured from the reference by tests/golden/make_golden.py)."""
import json
import os
import struct

import numpy as np
import pytest

from divergen_amd.data import pool_store as PS

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def golden_pool(tmp_path, monkeypatch):
    monkeypatch.chdir(G)                      # keys in the fixture are relative to tests/golden/
    z = np.load(os.path.join(G, "pool_decode.npz"))
    keys = [str(k) for k in z["keys"]]
    pool = {str(int(lab)): [k] for k, lab in zip(keys, z["labels"])}
    r = PS.build_shards(pool, str(tmp_path / "shards"))
    assert r["records"] == len(keys) and r["failed"] == []
    return z, keys, pool, str(tmp_path / "shards")


def test_store_returns_the_reference_decode(golden_pool):
    z, keys, pool, root = golden_pool
    store = PS.PoolStore(root, verify_crc=True)
    assert len(store) == len(keys) and sorted(store.keys()) == sorted(keys)
    for i, k in enumerate(keys):
        # random_scale mode hands the raw decode to cv2.resize: that array is the reference's decode of the key
        want = z["random_scale_%d_img" % i]
        got = store.view(k)
        assert got.dtype == np.uint8 and got.shape == want.shape
        assert np.array_equal(got, want), k
        assert not got.flags.writeable                       # zero-copy view of the mapping
        cp = store.loader(k)
        assert cp.flags.writeable and np.array_equal(cp, want)
        assert k in store
    assert "pool/nope.png" not in store
    with pytest.raises(KeyError):
        store.view("pool/nope.png")
    assert json.load(open(os.path.join(root, "inst_pool.json"))) == pool      # the index the sampler reads is unchanged
    store.close()


@pytest.mark.parametrize("mode", ["random_scale", "area_prior"])
def test_instpool_from_store_reaches_resize_like_the_reference(golden_pool, mode, monkeypatch):
    from divergen_amd.data.copypaste import InstPool
    z, keys, pool, root = golden_pool
    store = PS.PoolStore(root)
    ip = InstPool(pool, (256, 320), random_scale=(mode == "random_scale"), random_scale_min=0.5, random_scale_max=2.0,
                  use_largest_part=False, loader=store.loader)
    if mode == "area_prior":
        ip.HWms = {str(4 + i): [0.12, 0.03] for i in range(len(keys))}
    seen = {}

    class Stop(Exception):
        pass

    def rec(rgba, tw, th):
        seen["img"], seen["size"] = np.array(rgba), (tw, th)
        raise Stop()
    monkeypatch.setattr(InstPool, "_resize", staticmethod(rec))
    for i, k in enumerate(keys):
        assert ip.data_to_cat[k] == int(z["labels"][i])       # json key == label, as in the reference (mapper.py:147-152)
        np.random.seed(1000 + i)
        seen.clear()
        try:
            r = ip.load_rgba(k, (256, 320))
            assert r is None and ("%s_%d_rejected" % (mode, i)) in z.files
        except Stop:
            assert np.array_equal(seen["img"], z["%s_%d_img" % (mode, i)]), (mode, k)
            assert tuple(int(v) for v in z["%s_%d_size" % (mode, i)]) == seen["size"]


def test_pil_loader_and_store_agree_on_all_key_forms(golden_pool):
    from divergen_amd.data.copypaste import InstPool
    z, keys, pool, root = golden_pool
    store = PS.PoolStore(root)
    forms = {"pair": 0, "plain": 0, "star": 0}
    for k in keys:
        forms["star" if k.startswith("*") else "pair" if "|" in k else "plain"] += 1
        assert np.array_equal(InstPool._pil_loader(k), store.view(k))
    assert all(v > 0 for v in forms.values())


def test_ragged_pool_multi_shard_and_failures(tmp_path):
    rng = np.random.default_rng(0)
    arrays = {"k%03d" % i: rng.integers(0, 256, (int(rng.integers(1, 40)), int(rng.integers(1, 40)), 4), dtype=np.uint8)
              for i in range(57)}
    pool = {"0": list(arrays)[:20], "7": list(arrays)[20:] + ["missing"], "9": [], "11": [list(arrays)[0]]}   # dup + empty

    def decode(k):
        return arrays[k]                                      # KeyError for "missing": skipped, like mapper.py:374-378
    r = PS.build_shards(pool, str(tmp_path), decode=decode, shard_bytes=20000)
    assert r["records"] == 57 and r["failed"] == ["missing"] and r["shards"] > 3
    store = PS.PoolStore(str(tmp_path), verify_crc=True)
    assert len(store) == 57
    for k, a in arrays.items():
        assert np.array_equal(store.view(k), a)
    assert "missing" not in store
    # InstPool treats a loader failure as "image is None" (returns None) -- same as the reference
    from divergen_amd.data.copypaste import InstPool
    ip = InstPool({"3": ["missing"]}, (64, 64), random_scale=True, loader=store.loader)
    assert ip.load_rgba("missing", (64, 64)) is None


def test_empty_pool_and_corruption(tmp_path):
    r = PS.build_shards({}, str(tmp_path / "e"), decode=lambda k: None)
    assert r["records"] == 0 and r["shards"] == 1
    st = PS.PoolStore(str(tmp_path / "e"))
    assert len(st) == 0 and "x" not in st
    # corrupt one pixel: crc check fires only when asked for
    a = np.full((5, 6, 4), 7, np.uint8)
    PS.build_shards({"1": ["a"]}, str(tmp_path / "c"), decode=lambda k: a)
    p = os.path.join(str(tmp_path / "c"), "pool-00000.dgxpool")
    raw = bytearray(open(p, "rb").read())
    raw[64 + 32 + 16 + 3] ^= 0xFF
    open(p, "wb").write(bytes(raw))
    assert PS.PoolStore(p).view("a")[0, 0, 3] == 7 ^ 0xFF
    with pytest.raises(IOError):
        PS.PoolStore(p, verify_crc=True).view("a")
    # wrong magic / truncated file
    open(p, "wb").write(b"NOTAPOOL" + bytes(raw[8:]))
    with pytest.raises(ValueError):
        PS.PoolStore(p)
    open(p, "wb").write(b"abc")
    with pytest.raises(ValueError):
        PS.PoolStore(p)
    os.makedirs(str(tmp_path / "no_shards"))
    with pytest.raises(FileNotFoundError):
        PS.PoolStore(str(tmp_path / "no_shards"))


def test_hash_collision_is_resolved_by_the_stored_key(tmp_path, monkeypatch):
    monkeypatch.setattr(PS, "key_hash", lambda k: 42)         # every key collides
    a, b = np.zeros((2, 2, 4), np.uint8), np.ones((3, 1, 4), np.uint8)
    PS.build_shards({"1": ["a", "b"]}, str(tmp_path), decode=lambda k: {"a": a, "b": b}[k])
    st = PS.PoolStore(str(tmp_path))
    assert np.array_equal(st.view("a"), a) and np.array_equal(st.view("b"), b) and "c" not in st


def test_record_alignment(tmp_path):
    PS.build_shards({"1": ["kk", "a-much-longer-key/with/path.png|mask.png"]}, str(tmp_path),
                    decode=lambda k: np.zeros((3, 5, 4), np.uint8))
    raw = open(os.path.join(str(tmp_path), "pool-00000.dgxpool"), "rb").read()
    magic, ver, n, ioff = struct.unpack_from("<8sIIQ", raw, 0)
    assert magic == b"DGXPOOL1" and ver == 1 and n == 2 and ioff % 64 == 0
    idx = np.frombuffer(raw, dtype=PS.INDEX_DTYPE, count=n, offset=ioff)
    assert all(int(o) % 64 == 0 for o in idx["offset"]) and list(idx["hash"]) == sorted(idx["hash"])


def test_instpool_from_config_filters_by_frequency_and_uses_shards(golden_pool, tmp_path):
    """mapper.py:115-133,726-745: categories whose frequency is not in INST_POOL_FREQ are dropped (json key = id - 1)."""
    from divergen_amd.config import get_cfg
    from divergen_amd.data.copypaste import InstPool
    z, keys, pool, root = golden_pool
    labs = [int(v) for v in z["labels"]]
    infos = [{"id": lab + 1, "frequency": "rcf"[i % 3]} for i, lab in enumerate(labs)]
    json.dump(infos, open(tmp_path / "freq.json", "w"))
    json.dump(pool, open(tmp_path / "pool.json", "w"))
    cfg = get_cfg()
    cfg.INPUT.INST_POOL_PATH = str(tmp_path / "pool.json")
    cfg.MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH = str(tmp_path / "freq.json")
    cfg.INPUT.MEAN_STD2_PATH = str(tmp_path / "none.json")
    cfg.INPUT.INST_POOL_FREQ = ["r", "f"]
    cfg.INPUT.INST_POOL_SHARDS = root
    cfg.INPUT.RANDOM_SCALE = True
    ip = InstPool.from_config(cfg)
    assert sorted(ip.cats) == sorted(lab for i, lab in enumerate(labs) if "rcf"[i % 3] in "rf")
    assert ip.max_samples == cfg.INPUT.INST_POOL_MAX_SAMPLES and ip.random_scale
    k = ip.dataset[0]
    np.random.seed(0)
    r = ip.load_rgba(k, (256, 320))
    assert r is None or (r[0].shape[2] == 4 and r[1] == ip.data_to_cat[k])
    assert ip.loader.__self__.__class__.__name__ == "PoolStore"
