#!/bin/bash
# A/B of one environment switch inside ONE gpurun call (same box):  bash tools/ab_env.sh DGX_CONV_WGRAD_MULTI [rounds]
cd $GRAFT_REPO_ROOT
for r in $(seq ${2:-2}); do
for v in 0 1; do
  env $1=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; o={x['family']:x for x in d['roofline_other']}
print('$1=$v', 'ms/step %.2f' % d['ms_per_step'], 'gemm ms %.2f frac %.3f' % (r['total_ms_per_step'], r['frac']), ' '.join('%s %.2f' % (k, o[k]['total_ms_per_step']) for k in ('wgrad', 'attn_bwd', 'attn_fwd') if k in o))
"
done
done
