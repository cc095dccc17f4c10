cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_loader.py -q -m gpu -k "do_train" 2>&1 | tail -30 > gpurun_out/r06_t4.log
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --through-loader --loader-shards --loader-record 8 > gpurun_out/r06_d1.json 2> gpurun_out/r06_d1.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_prof_loader -o loader --output-format csv -- python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --through-loader --loader-shards > gpurun_out/r06_d2.json 2> gpurun_out/r06_d2.err
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r06_prof_loader/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open('gpurun_out/r06_loader_kernel_top.txt', 'w') as o:
    o.write('total kernel ms (28 steps incl. warm-up): %.1f\n' % (tot / 1e6))
    for r in rows[:40]:
        o.write('%9.2f ms %6s calls  %s\n' % (float(r['TotalDurationNs']) / 1e6, r['Calls'], r['Name'][:110]))
PY
rm -rf gpurun_out/r06_prof_loader
