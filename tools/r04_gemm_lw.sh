#!/bin/bash
# Round 4: loader-wave GEMM (gemm_lw.hip) -- correctness through the GEMM tests, standalone shape timing of both forms, phase stamps,
# and the in-step A/B.  One gpurun call: bash tools/r04_gemm_lw.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
export DGX_GEMM_LW=1
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q > $O/test_gemm_lw1.txt 2>&1; echo "gemm tests LW=1 rc=$?"; tail -3 $O/test_gemm_lw1.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_modules.py -x -q -k "conv or fpn or head or tower" > $O/test_conv_lw1.txt 2>&1; echo "conv tests LW=1 rc=$?"; tail -3 $O/test_conv_lw1.txt
for lw in 0 1; do DGX_GEMM_LW=$lw timeout 600 python tools/gemm_shapes_probe.py --own-only > $O/shapes_lw$lw.txt 2>&1; tail -1 $O/shapes_lw$lw.txt; done
DGX_GEMM_LW=1 timeout 300 python tools/gemm_phase_probe.py 256x192,192x192,128x192 0 > $O/phases_lw1.txt 2>&1
DGX_GEMM_LW=0 DGX_GEMM_2WG=0 timeout 300 python tools/gemm_phase_probe.py 256x192,192x192,128x192 0 > $O/phases_lw0.txt 2>&1
grep -h "per K-tile" $O/phases_lw1.txt $O/phases_lw0.txt
for r in 1 2; do for v in 0 1 2; do
  timeout 600 python bench.py --dev gemm_lw=$v --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; o={x['family']:x for x in d['roofline_other']}
print('LW=$v', 'ms/step %.2f' % d['ms_per_step'], 'gemm ms %.2f frac %.3f' % (r['total_ms_per_step'], r['frac']), ' '.join('%s %.2f' % (k, o[k]['total_ms_per_step']) for k in ('wgrad', 'attn_bwd', 'attn_fwd') if k in o))
"
done; done 2>&1 | tee $O/bench_ab.txt
