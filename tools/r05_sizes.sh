#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5z; mkdir -p $O
for cfg in "--size 1024" "--size 896" "--swin T --size 1024"; do
  timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $cfg > $O/b.json 2> $O/err.txt
  python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("$cfg", "ms/step %.2f img/s %.1f | gemm %.2f ms frac %.3f |" % (d["ms_per_step"], d["value"], d["roofline"]["total_ms_per_step"], d["roofline"]["frac"]), " ".join("%s %.2f/%.3f" % (o["family"], o["total_ms_per_step"], o["frac"]) for o in d.get("roofline_other", [])))
PY
done
