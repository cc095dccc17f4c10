#!/bin/bash
# what bounds the loader-wave weight-gradient kernel: kernel time, HBM fetch, L2 hit rate, LDS conflicts, wave stall split
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wlw; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PROBE_BIAS=${PROBE_BIAS:-0}
run() { # tag, counters
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pw_$1 -o p -- python $R/tools/wgrad_lw_probe.py 7 > /dev/null 2> $O/pmc_$1.err
  f=$(find /tmp/pw_$1 -name "*counter_collection.csv" | head -1)
  python - "$f" "$1" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    if "wgrad" not in k: continue
    acc[(k, r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in acc.items():
    print(sys.argv[2], k, "grid", g, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n", len(next(iter(d.values()))))
PY
}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_s -o p -- python $R/tools/wgrad_lw_probe.py 7 > /dev/null 2> $O/stats.err
head -4 $(find /tmp/pw_s -name "*kernel_stats.csv" | head -1) | cut -c1-200
run f "FETCH_SIZE"
run t "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run l "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES"
run w "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
run c "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum"
