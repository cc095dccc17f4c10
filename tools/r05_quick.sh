#!/bin/bash
# GEMM tests + two bench runs (same box): ms/step and the family times of the event clock
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_pins.py -x -q 2>&1 | tail -2
for r in 1 2; do
  timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $BENCH_EXTRA > $O/bench$r.json 2> $O/err.txt
  python - <<PY
import json
d=json.loads(open("$O/bench$r.json").read().strip().splitlines()[-1])
print("ms/step %.2f  img/s %.1f | gemm %.2f ms frac %.3f |" % (d["ms_per_step"], d["value"], d["roofline"]["total_ms_per_step"], d["roofline"]["frac"]), " ".join("%s %.2f" % (o["family"], o["total_ms_per_step"]) for o in d.get("roofline_other", [])))
PY
done
