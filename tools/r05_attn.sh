#!/bin/bash
# attention kernels: standalone harness timings at the four Swin-L launch shapes (+ phase clocks of wave 0) and the parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5attn; mkdir -p $O
bash tools/probes/build_attn_bench.sh > $O/build.txt 2>&1 || tail -5 $O/build.txt
for s in "968 6" "242 12" "72 24" "18 48" "968 6 484" "242 12 121" "72 24 36" "18 48 9"; do ./tools/probes/attn_bwd_bench_w8 $s; done 2>&1 | tee $O/times.txt
./tools/probes/attn_bwd_bench_w0 968 6 2>&1 | tee $O/clocks_968x6.txt
./tools/probes/attn_bwd_bench_w0 72 24 2>&1 | tee $O/clocks_72x24.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_swinL_geometry.py -x -q -k "attention or geometry or swin" 2>&1 | tail -3
