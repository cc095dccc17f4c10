#!/bin/bash
# In-situ vs standalone GEMM diagnosis (round 3): kernel trace joined with the launch log (hipGraph segments off so that every
# launch is logged), then one counter pass each over the step and over the back-to-back shape loop.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p0 -o p -- python $R/bench.py --dev gemm_log=/tmp/g0.log --no-graphs --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > $O/bench_trace.json 2> $O/trace.err
python $R/tools/gemm_insitu.py $(find /tmp/p0 -name "*kernel_trace.csv" | head -1) /tmp/g0.log 9 > $O/r03_gemm_insitu.txt 2>&1
if [ "$1" = "pmc" ]; then
C="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p1 -o p -- python $R/bench.py --dev gemm_log=/tmp/g1.log --no-graphs --steps 2 --warmup 2 --no-roofline --no-cpu-baseline > /dev/null 2> $O/pmc_insitu.err
python $R/tools/gemm_insitu_pmc.py $(find /tmp/p1 -name "*counter_collection.csv" | head -1) $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) /tmp/g1.log 4 > $O/r03_gemm_insitu_pmc.txt 2>&1
DGX_GEMM_LOG=/tmp/g2.log rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p2 -o p -- python $R/tools/gemm_shapes_probe.py --own-only > $O/standalone.txt 2> $O/pmc_standalone.err
python $R/tools/gemm_insitu_pmc.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) /tmp/g2.log 24 > $O/r03_gemm_standalone_pmc.txt 2>&1
fi
head -40 $O/r03_gemm_insitu.txt
