#!/bin/bash
# In-situ time per GEMM shape with the loader-wave kernel off / everywhere / only where K > 768 (one gpurun call, same box)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${1:-0 1}; do
  rm -rf /tmp/p_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$v -o p -- python $R/bench.py --dev gemm_lw=$v --dev gemm_log=/tmp/g_$v.log --no-graphs --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > $O/bench_$v.json 2> $O/trace_$v.err
  python $R/tools/gemm_insitu.py $(find /tmp/p_$v -name "*kernel_trace.csv" | head -1) /tmp/g_$v.log 9 > $O/insitu_lw$v.txt 2>&1
  tail -1 $O/insitu_lw$v.txt
done
