#!/bin/bash
# Round 4: LayerNorm family -- kernel tests, block / model tests, and the in-step A/B is the bench line itself (compare with r4c)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "layernorm or ln or patch_merge" > $O/test_ln.txt 2>&1; echo "ln tests rc=$?"; tail -3 $O/test_ln.txt
timeout 1200 python -m pytest tests/test_gpu_parity_modules.py tests/test_gpu_swinL_geometry.py tests/test_gpu_model.py -x -q > $O/test_model.txt 2>&1; echo "model tests rc=$?"; tail -3 $O/test_model.txt
for r in 1 2; do
  timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; o={x['family']:x for x in d['roofline_other']}
print('ms/step %.2f' % d['ms_per_step'], 'gemm ms %.2f frac %.3f' % (r['total_ms_per_step'], r['frac']), ' '.join('%s %.2f' % (k, o[k]['total_ms_per_step']) for k in ('wgrad', 'attn_bwd', 'attn_fwd') if k in o))
"
done 2>&1 | tee $O/bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pln -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/pln -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/$O/kernel_stats.csv
grep -E "ln_|residual_|pm_ln" $f | cut -c1-150 | head -20
