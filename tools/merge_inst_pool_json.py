"""CLI of DG/tools/merge_inst_pool_json.py (same arguments): merge instance-pool json files, optionally rewriting path prefixes."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from divergen_amd.data.factory import merge_inst_pools  # noqa: E402

if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--inst_pool_path", action="append", default=[])
    p.add_argument("--enable_replace", action="store_true")
    p.add_argument("--before_prefix", action="append", default=[])
    p.add_argument("--after_prefix", action="append", default=[])
    p.add_argument("--out_inst_pool_path", default="output/debug/230915_sim")
    a = p.parse_args()
    assert a.inst_pool_path, "inst_pool_path is empty"
    pools = [json.load(open(x)) for x in a.inst_pool_path]
    out = merge_inst_pools(pools, a.before_prefix, a.after_prefix) if a.enable_replace else merge_inst_pools(pools)
    os.makedirs(os.path.dirname(a.out_inst_pool_path) or ".", exist_ok=True)
    json.dump(out, open(a.out_inst_pool_path, "w"))
