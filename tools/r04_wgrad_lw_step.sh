#!/bin/bash
# loader-wave weight gradients inside the training step: tests, then A/B of the bench line (DGX_WGRAD_LW_MIN_M=0: round-3 launches only)
mkdir -p gpurun_out/wlw
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wgrad or swin_block or block" > gpurun_out/wlw/test.log 2>&1; tail -3 gpurun_out/wlw/test.log
PROBE_BETA=0 python tools/wgrad_lw_probe.py 7 4 2>&1 | grep -v amdgpu
for r in 1 2; do
for m in 4096 0 2048; do
  DGX_WGRAD_LW_MIN_M=$m python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; o={x['family']:x for x in d['roofline_other']}
print('LW_MIN_M=$m', 'ms/step %.2f' % d['ms_per_step'], 'gemm ms %.2f frac %.3f' % (r['total_ms_per_step'], r['frac']), 'wgrad ms %.2f' % o['wgrad']['total_ms_per_step'])
"
done
done
