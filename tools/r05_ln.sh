#!/bin/bash
# LayerNorm kernels: parity tests + in-situ time per kernel name from a short profiled bench run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ln; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_modules.py -x -q -k "layernorm or swin or patch_merge or block" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ln -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench.json 2>/dev/null
f=$(find /tmp/p_ln -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/$O/kernel_stats.csv
python - <<PY
import csv,json
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/$O/kernel_stats.csv")))
tot=0
for r in rows:
    n=r["Name"]
    if "ln_" in n:
        ms=float(r["TotalDurationNs"])/1e6/9; tot+=ms
        print("%.3f ms/step %5.1f calls avg %.1f us  %s" % (ms, int(r["Calls"])/9, float(r["AverageNs"])/1e3, n[:90]))
print("ln family total %.3f ms/step (9 steps in the trace)" % tot)
d=json.loads(open("$GRAFT_REPO_ROOT/$O/bench.json").read().strip().splitlines()[-1]); print("ms/step under profiler", d["ms_per_step"])
PY
