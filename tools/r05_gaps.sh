#!/bin/bash
# one kernel trace of bench.py -> idle gaps inside the steps (trace_gaps.py) and the weight-gradient launches by grid (trace_by_grid.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_gaps -o p -- python $R/bench.py --steps 8 --warmup 4 --no-roofline --no-cpu-baseline $BENCH_EXTRA > $O/bench_under_rocprof.json 2> $O/err.txt
T=$(find /tmp/p_gaps -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $T 4 > $O/gaps.txt 2>&1
python $R/tools/trace_by_grid.py $T 12 wgrad > $O/wgrad_by_grid.txt 2>&1
python $R/tools/trace_by_grid.py $T 12 cp_ > $O/cp_by_grid.txt 2>&1
head -70 $O/gaps.txt
