#!/bin/bash
# Average duration of the kernels matching a pattern under the two values of an environment switch, same box:
#   bash tools/ab_env_kernel.sh DGX_ADAMW_NT adamw
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in 1 2; do
for v in 0 1; do
  rm -rf /tmp/abk_$v
  env $1=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$v -o p -- python $R/bench.py --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/abk_$v -name "*kernel_stats.csv" | head -1)
  echo "== $1=$v"; python - "$f" "$2" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-70.70s calls %6s avg %9.1f ns" % (r["Name"], r["Calls"], float(r["AverageNs"])))
PY
done
done
