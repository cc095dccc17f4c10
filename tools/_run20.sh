cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_swinL_geometry.py -x -q 2>&1 | tail -2
for v in 0 1; do echo "== DGX_GEMM_2WG=$v"; DGX_GEMM_2WG=$v python tools/gemm_shapes_probe.py --own-only 2>/dev/null | grep -E "s0\.|s1\." | cut -c1-110; done
for r in 1 2; do for v in 0 1; do echo -n "2wg=$v "; DGX_GEMM_2WG=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ms/step %.2f' % d['ms_per_step'], 'gemm ms %.2f frac %.3f' % (r['total_ms_per_step'], r['frac']))"; done; done
