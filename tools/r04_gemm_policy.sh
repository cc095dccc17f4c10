#!/bin/bash
# Round 4: the shipped GEMM dispatch (gemm_lw where it wins, gemm_nt elsewhere) -- tests under the default policy and with the
# loader-wave kernel forced everywhere, then the in-step A/B against gemm_nt only (same box).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q > $O/test_gemm_policy.txt 2>&1; echo "gemm tests (policy) rc=$?"; tail -1 $O/test_gemm_policy.txt
DGX_GEMM_LW=1 timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q > $O/test_gemm_lw1.txt 2>&1; echo "gemm tests LW=1 rc=$?"; tail -1 $O/test_gemm_lw1.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_modules.py tests/test_gpu_swinL_geometry.py -x -q > $O/test_modules.txt 2>&1; echo "module tests rc=$?"; tail -1 $O/test_modules.txt
for r in 1 2; do for v in 0 2; do
  timeout 600 python bench.py --dev gemm_lw=$v --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; o={x['family']:x for x in d['roofline_other']}
print('LW=$v', 'ms/step %.2f' % d['ms_per_step'], 'gemm ms %.2f frac %.3f' % (r['total_ms_per_step'], r['frac']), ' '.join('%s %.2f' % (k, o[k]['total_ms_per_step']) for k in ('wgrad', 'attn_bwd', 'attn_fwd') if k in o))
"
done; done 2>&1 | tee $O/bench_ab.txt
bash tools/r04_insitu_ab.sh "2" > /dev/null 2>&1; tail -1 gpurun_out/r4b/insitu_lw2.txt
