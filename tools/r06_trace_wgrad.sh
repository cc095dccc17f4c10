cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tr -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 4 --no-roofline --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r06_trace.err
f=$(find /tmp/p_tr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r06_wgrad_launches.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: between the last two adamw launches
idx = [i for i, r in enumerate(rows) if 'adamw_ema_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
print('kernels in the last step:', b - a)
for r in rows[a + 1:b]:
    n = r['Kernel_Name']
    if 'wgrad' in n or 'transpose_grouped' in n:
        print('%-28s grid %6s wg %4s  %8.1f us' % (n.split('(')[0][-28:], r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?')), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
