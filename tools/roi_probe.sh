cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -o p -- python $R/bench.py --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/trace_by_grid.py $(find /tmp/rp -name "*kernel_trace.csv" | head -1) 9 roi_ 
python $R/tools/trace_by_grid.py $(find /tmp/rp -name "*kernel_trace.csv" | head -1) 9 direct_copy
