#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
for r in 1 2; do for f in "" "--inputs-resident"; do
  timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $f 2>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f', 'ms/step %.2f' % d['ms_per_step'], 'host issue %.2f' % d['host_issue_ms_per_step'], d['inputs'])
" || tail -5 $O/err.txt
done; done 2>&1 | tee $O/bench_h2d.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -s -k "end_to_end" 2>&1 | grep -E "parity report|passed|failed" | cut -c1-3000 | tee $O/e2e_report.txt
