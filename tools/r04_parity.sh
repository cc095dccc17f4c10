#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_swinL_geometry.py -x -q -s > $O/parity.txt 2>&1; echo "rc=$?"
grep -E "relative L2|product .* oracle|passed|failed|Error|assert" $O/parity.txt | cut -c1-220 | tail -70
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "found_inf or full_model or emit" 2>&1 | tail -3
