mkdir -p gpurun_out/r06x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_g -o p -- python $R/bench.py --steps 8 --warmup 6 --no-roofline --no-cpu-baseline > $R/gpurun_out/r06x/bench.json 2> $R/gpurun_out/r06x/prof.err
python $R/tools/step_gaps.py $(find /tmp/p_g -name "*kernel_trace.csv" | head -1) 3 16 > $R/gpurun_out/r06x/gaps.txt 2>&1
python $R/tools/decode_window.py $(find /tmp/p_g -name "*kernel_trace.csv" | head -1) 1 > $R/gpurun_out/r06x/window.txt 2>&1
