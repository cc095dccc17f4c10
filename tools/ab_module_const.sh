#!/bin/bash
# A/B of a module constant of the package inside ONE gpurun call (same box), alternating:  bash tools/ab_module_const.sh divergen_amd.layers.swin_block _LW_ITEMS 1000 850 [rounds]
cd $GRAFT_REPO_ROOT
MOD=$1; NAME=$2; A=$3; B=$4
for r in $(seq ${5:-3}); do
for v in $A $B; do
  python - <<PY
import sys, json, runpy, io, contextlib
import $MOD as m
setattr(m, "$NAME", $v)
sys.argv = ["bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads(buf.getvalue().strip().splitlines()[-1])
o = {x["family"]: x for x in d["roofline_other"]}
print("$NAME=$v", "ms/step %.2f" % d["ms_per_step"], "gemm %.2f" % d["roofline"]["total_ms_per_step"], "wgrad %.2f" % o["wgrad"]["total_ms_per_step"])
PY
done
done
