"""Build the instance-pool shard store (divergen_amd/data/pool_store.py) from one or several INST_POOL_PATH jsons.
Merging and prefix replacement follow DG/tools/merge_inst_pool_json.py (:60-84): lists of equal category keys are
concatenated, `--before-prefix/--after-prefix` rewrite the paths of the matching input json.

    python tools/build_inst_pool.py --inst-pool-path a.json --inst-pool-path b.json --out datasets/inst_pool_shards
"""
import argparse
import json
import sys
import time

sys.path.insert(0, ".")
from divergen_amd.data.pool_store import build_shards  # noqa: E402


def merge(paths, before=(), after=()):
    out = {}
    for i, p in enumerate(paths):
        with open(p) as f:
            cur = json.load(f)
        for k, v in cur.items():
            if before:
                v = [x.replace(before[i], after[i]) for x in v]
            out.setdefault(k, []).extend(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inst-pool-path", action="append", default=[], required=True)
    ap.add_argument("--before-prefix", action="append", default=[])
    ap.add_argument("--after-prefix", action="append", default=[])
    ap.add_argument("--out", required=True)
    ap.add_argument("--shard-gib", type=float, default=1.0)
    a = ap.parse_args()
    if a.before_prefix or a.after_prefix:
        assert len(a.before_prefix) == len(a.after_prefix) == len(a.inst_pool_path), "one prefix pair per input json"
    t0 = time.time()
    r = build_shards(merge(a.inst_pool_path, a.before_prefix, a.after_prefix), a.out, shard_bytes=int(a.shard_gib * (1 << 30)),
                     log=print)
    print("%d records in %d shard(s), %d keys skipped, %.1f s" % (r["records"], r["shards"], len(r["failed"]), time.time() - t0))


if __name__ == "__main__":
    main()
