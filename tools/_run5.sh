cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -k "copy_paste" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p0 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > /tmp/b.json 2>/dev/null
f=$(find /tmp/p0 -name "*kernel_trace.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03i
python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $f 9 wgrad > $GRAFT_REPO_ROOT/gpurun_out/r03i/wgrad_by_grid.txt
python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $f 9 cp_ > $GRAFT_REPO_ROOT/gpurun_out/r03i/cp_by_grid.txt
python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $f 9 > $GRAFT_REPO_ROOT/gpurun_out/r03i/all_by_grid.txt
cat $GRAFT_REPO_ROOT/gpurun_out/r03i/cp_by_grid.txt
head -30 $GRAFT_REPO_ROOT/gpurun_out/r03i/wgrad_by_grid.txt
cd $GRAFT_REPO_ROOT; python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-330
