cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 3 4 5; do echo -n "2wg=$v "; DGX_GEMM_2WG=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ms/step %.2f' % d['ms_per_step'], 'gemm ms %.2f frac %.3f' % (r['total_ms_per_step'], r['frac']))"; done; done
