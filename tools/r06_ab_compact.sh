# same-box A/B of round 6's two changes to the step itself: compact window order, graphed Swin block groups (each line: 60 timed steps)
O=gpurun_out/r06_ab; mkdir -p $O
for rep in 1 2; do
for size in 1024 896; do
python bench.py --size $size --steps 60 --warmup 10 --no-cpu-baseline > $O/final_${size}_$rep.json 2>> $O/err.txt
python bench.py --size $size --steps 60 --warmup 10 --no-cpu-baseline --no-compact > $O/nocompact_${size}_$rep.json 2>> $O/err.txt
python bench.py --size $size --steps 60 --warmup 10 --no-cpu-baseline --no-block-graphs > $O/noblockgraphs_${size}_$rep.json 2>> $O/err.txt
python bench.py --size $size --steps 60 --warmup 10 --no-cpu-baseline --no-compact --no-block-graphs > $O/round5form_${size}_$rep.json 2>> $O/err.txt
done
done
python - <<'PY'
import json, glob, os
rows = []
for f in sorted(glob.glob('gpurun_out/r06_ab/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    fam = {o['family']: o['total_ms_per_step'] for o in [d['roofline']] + d['roofline_other']}
    rows.append('%-28s %7.2f ms/step  gemm %6.3f  wgrad %6.3f  attn_fwd %5.3f  attn_bwd %5.3f' % (os.path.basename(f)[:-5], d['ms_per_step'], fam.get('gemm_nt', 0), fam.get('wgrad', 0), fam.get('attn_fwd', 0), fam.get('attn_bwd', 0)))
open('gpurun_out/r06_ab/table.txt', 'w').write('\n'.join(rows) + '\n')
print('\n'.join(rows))
PY
