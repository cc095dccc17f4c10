#!/bin/bash
# A/B of two builds of libdgx inside ONE gpurun call (same box): DGX_LIB selects the library (divergen_amd/_lib.py).
#   bash tools/ab_lib.sh divergen_amd/csrc/_obj/libdgx_old.so [rounds]
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/$1
for r in $(seq ${2:-2}); do
for tag in old new; do
  if [ $tag = old ]; then export DGX_LIB=$OLD; else unset DGX_LIB; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; o={x['family']:x for x in d['roofline_other']}
print('$tag', 'ms/step %.2f' % d['ms_per_step'], 'gemm ms %.2f frac %.3f' % (r['total_ms_per_step'], r['frac']), 'wgrad ms %.2f' % o['wgrad']['total_ms_per_step'], 'attn bwd %.2f fwd %.2f' % (o['attn_bwd']['total_ms_per_step'], o['attn_fwd']['total_ms_per_step']))
"
done
done
