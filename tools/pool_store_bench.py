"""Instance-pool decode throughput on the host: reference-style PIL open of `path|maskpath` PNG pairs vs the
mmap shard store, on synthetic 512x512 RGBA instances (the size DeepFloyd-IF stage II emits; DG/DATA.md)."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from divergen_amd.data import pool_store as PS  # noqa: E402


def main(n=200, size=512):
    from PIL import Image
    rng = np.random.default_rng(0)
    d = tempfile.mkdtemp(prefix="poolbench_")
    keys = []
    yy, xx = np.mgrid[0:size, 0:size]
    for i in range(n):
        # smooth content (PNG-compressible like generated photos are not; this flatters PIL) + noise
        base = (rng.integers(0, 256, (size // 8, size // 8, 3)).repeat(8, 0).repeat(8, 1)).astype(np.uint8)
        rgb = np.clip(base.astype(np.int16) + rng.integers(-20, 20, (size, size, 3)), 0, 255).astype(np.uint8)
        alpha = (((xx - size / 2) ** 2 + (yy - size / 2) ** 2) < (size * (0.25 + 0.2 * rng.random())) ** 2).astype(np.uint8) * 255
        p, m = os.path.join(d, "i%05d.png" % i), os.path.join(d, "i%05d_mask.png" % i)
        Image.fromarray(rgb, "RGB").save(p)
        Image.fromarray(alpha, "L").save(m)
        keys.append(p + "|" + m)
    t0 = time.perf_counter()
    r = PS.build_shards({"0": keys}, os.path.join(d, "shards"))
    t_build = time.perf_counter() - t0
    store = PS.PoolStore(os.path.join(d, "shards"))
    order = rng.permutation(n)
    t0 = time.perf_counter()
    s = 0
    for i in order:
        s += int(PS.decode_key(keys[i])[0, 0, 0])
    t_pil = time.perf_counter() - t0
    t0 = time.perf_counter()
    s2 = 0
    for i in order:
        s2 += int(store.loader(keys[i])[0, 0, 0])
    t_store = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in order:
        store.view(keys[i])
    t_view = time.perf_counter() - t0
    assert s == s2
    png = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".png"))
    shard = sum(os.path.getsize(os.path.join(d, "shards", f)) for f in os.listdir(os.path.join(d, "shards")))
    print("%d instances %dx%d: build %.2f s; PNG %.1f MB -> shards %.1f MB" % (n, size, size, t_build, png / 1e6, shard / 1e6))
    print("PIL open + convert + mask (reference decode): %7.1f inst/s/core" % (n / t_pil))
    print("store.loader (mmap + private copy)          : %7.1f inst/s/core" % (n / t_store))
    print("store.view   (zero-copy lookup)             : %7.1f inst/s/core" % (n / t_view))


if __name__ == "__main__":
    main()
