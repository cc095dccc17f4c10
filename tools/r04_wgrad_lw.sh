#!/bin/bash
# correctness + speed of the loader-wave weight-gradient kernel
mkdir -p gpurun_out/wlw
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wgrad" > gpurun_out/wlw/test.log 2>&1; tail -5 gpurun_out/wlw/test.log
DGX_WGRAD_LW=1 timeout 300 python tools/wgrad_lw_probe.py > gpurun_out/wlw/probe_lw.txt 2>&1; cat gpurun_out/wlw/probe_lw.txt
PROBE_BIAS=0 DGX_WGRAD_LW=1 timeout 300 python tools/wgrad_lw_probe.py > gpurun_out/wlw/probe_lw_nobias.txt 2>&1; cat gpurun_out/wlw/probe_lw_nobias.txt
DGX_WGRAD_LW=0 timeout 300 python tools/wgrad_lw_probe.py > gpurun_out/wlw/probe_256.txt 2>&1; cat gpurun_out/wlw/probe_256.txt
cd /tmp && export TMPDIR=/tmp
PROBE_BIAS=0 DGX_WGRAD_LW=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_wlw -o p -- python $GRAFT_REPO_ROOT/tools/wgrad_lw_probe.py 7 > /dev/null 2>&1
cp /tmp/p_wlw/p_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/wlw/kernel_stats.csv 2>/dev/null; head -5 /tmp/p_wlw/p_kernel_stats.csv
