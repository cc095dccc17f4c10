cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r03i
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_stats -o p -- python $R/bench.py --steps 8 --warmup 4 --no-roofline --no-cpu-baseline > /dev/null 2>&1
n=$(python $R/tools/trace_stats.py $(find /tmp/p_stats -name "*kernel_trace.csv" | head -1) 4 $R/gpurun_out/r03i/steady_stats.csv)
python $R/tools/prof_summary.py $R/gpurun_out/r03i/steady_stats.csv $n
