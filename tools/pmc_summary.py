"""profiles/rNN_pmc.json from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE and WRITE_SIZE, separate runs):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline
    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <warm-up steps> <timed steps> profiles/r03_pmc.json

Only the TIMED steps are counted: the dispatches are walked in dispatch order and cut at the fused optimizer kernel
(`adamw_ema_kernel`, exactly one dispatch per step), so the hipGraph capture warm-ups and the warm-up steps drop out while the
graph-replayed launches of the timed steps stay in -- the same launch population bench.py's `algorithmic_bytes_per_launch`
covers (round 2 divided by different populations on the two sides).
Per kernel family (bench.FAMILY_KERNELS): dispatches, summed counters, HBM bytes per STEP and per LAUNCH of the family's entry
point (a launch of the wgrad entry point = one partial + its reduce dispatches; a split-K GEMM = kernel + fold), with the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE (KB) counts a 128-byte request of a wide (16 B/lane)
streaming read as 64 B, so it is doubled for kernels whose reads are 16 B/lane (`wide`); WRITE_SIZE is taken as reported."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FAMILY_KERNELS = {
    "gemm_nt": ("gemm_nt_kernel", "gemm_lw_kernel", "gemm_k192_kernel", "gemm_splitk_fold_kernel"),
    "wgrad": ("wgrad256_partial_kernel", "wgrad256_reduce_kernel", "wgrad256_bias_reduce_kernel", "wgrad_partial_kernel", "wgrad_reduce_kernel", "wgrad_lw_kernel"),
    "attn_fwd": ("win_attn_fwd_kernel",),
    "attn_bwd": ("win_attn_bwd_kernel",),
}
# entry-point launches are counted on the family's MAIN kernel
MAIN = {"gemm_nt": ("gemm_nt_kernel", "gemm_lw_kernel", "gemm_k192_kernel"), "wgrad": ("partial_kernel", "wgrad_lw_kernel"), "attn_fwd": "win_attn_fwd_kernel", "attn_bwd": "win_attn_bwd_kernel"}
# 16 B/lane streaming reads in every family (LDS-direct tile loads, float4 folds; the attention kernels load their q / k / v / dO head
# slices as 16-byte fragments, window_attention.hip ld_frag_global): the correction is applied UNIFORMLY (rounds 3-5 left the attention
# families undoubled and reported attn_fwd traffic at 0.62 of its algorithmic bytes, which is impossible)
WIDE = {"gemm_nt": True, "wgrad": True, "attn_fwd": True, "attn_bwd": True}


def collect(path, counter, warm, timed):
    out = {k: {"kb": 0.0, "dispatches": 0, "main": 0, "per_kernel": {}} for k in FAMILY_KERNELS}
    with open(path) as f:
        rows = [r for r in csv.DictReader(f) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    step = 0                                   # index of the training step a dispatch belongs to
    for r in rows:
        in_timed = warm <= step < warm + timed
        if "adamw_ema_kernel" in r["Kernel_Name"]:
            step += 1
        if in_timed:
            for fam, pats in FAMILY_KERNELS.items():
                if any(p in r["Kernel_Name"] for p in pats):
                    out[fam]["kb"] += float(r["Counter_Value"])
                    out[fam]["dispatches"] += 1
                    out[fam]["main"] += any(mk in r["Kernel_Name"] for mk in ((MAIN[fam],) if isinstance(MAIN[fam], str) else MAIN[fam]))
                    pk = out[fam]["per_kernel"].setdefault(r["Kernel_Name"].split("(")[0][-60:], [0, 0.0])
                    pk[0] += 1
                    pk[1] += float(r["Counter_Value"])
    assert step >= warm + timed, "fewer optimizer dispatches (%d) than warm-up + timed steps (%d)" % (step, warm + timed)
    return out


def main():
    fcsv, wcsv, warm, steps, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    F, W = collect(fcsv, "FETCH_SIZE", warm, steps), collect(wcsv, "WRITE_SIZE", warm, steps)
    res = {"_how": __doc__.strip(), "_timed_steps_counted": steps, "_warmup_steps_skipped": warm}
    for fam in FAMILY_KERNELS:
        if not F[fam]["main"]:
            continue
        fetch = F[fam]["kb"] * 1000.0 * (2.0 if WIDE[fam] else 1.0)
        write = W[fam]["kb"] * 1000.0
        res[fam] = {"entry_point_launches": F[fam]["main"], "dispatches": F[fam]["dispatches"],
                    "FETCH_SIZE_KB_raw_sum": F[fam]["kb"], "fetch_doubled": WIDE[fam], "WRITE_SIZE_KB_sum": W[fam]["kb"],
                    "per_kernel_dispatches_FETCH_KB_WRITE_KB": {k: [v[0], v[1], W[fam]["per_kernel"].get(k, [0, 0.0])[1]]
                                                                for k, v in F[fam]["per_kernel"].items()},
                    "launches_per_step": F[fam]["main"] / float(steps),
                    "hbm_bytes_per_step": (fetch + write) / steps,
                    "hbm_bytes_per_launch": (fetch + write) / F[fam]["main"]}
    with open(dst, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if not kk.startswith("per_kernel")} for k, v in res.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
