#!/bin/bash
# loader-wave weight gradients inside the training step: tests, then A/B of the bench line (swin_block._LW_MIN_M = 0: round-3 launches only)
mkdir -p gpurun_out/wlw
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wgrad or swin_block or block" > gpurun_out/wlw/test.log 2>&1; tail -3 gpurun_out/wlw/test.log
PROBE_BETA=0 python tools/wgrad_lw_probe.py 7 4 2>&1 | grep -v amdgpu
# A/B inside the step: the loader-wave launches on (stage 2: _LW_MIN_M = 4096) / off (0) / also for stage 3 (2048)
bash tools/ab_module_const.sh divergen_amd.layers.swin_block _LW_MIN_M 4096 0 2 2>&1 | grep "_LW"
bash tools/ab_module_const.sh divergen_amd.layers.swin_block _LW_MIN_M 4096 2048 2 2>&1 | grep "_LW"
