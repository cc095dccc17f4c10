#!/bin/bash
# steady-state kernel statistics of bench.py (rocprofv3 kernel trace, first 4 steps dropped) -> family summary + top kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o p -- python $R/bench.py --steps 8 --warmup 4 --no-roofline --no-cpu-baseline $BENCH_EXTRA > $O/bench_under_rocprof.json 2> $O/stats.err
n=$(python $R/tools/trace_stats.py $(find /tmp/p_stats -name "*kernel_trace.csv" | head -1) 4 $O/kernel_stats.csv)
python $R/tools/prof_summary.py $O/kernel_stats.csv $n | tee $O/summary.txt
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:45]:
    print("%.3f ms/step %6.1f calls  avg %7.1f us  %s" % (float(r["TotalDurationNs"])/1e6/$n, int(r["Calls"])/$n, float(r["AverageNs"])/1e3, r["Name"][:100]))
PY
