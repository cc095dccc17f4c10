#!/bin/bash
# Average duration of the kernels matching a pattern with two builds of libdgx, same box:  bash tools/ab_kernel.sh <old lib> <pattern>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in old new; do
  if [ $tag = old ]; then export DGX_LIB=$R/$1; else unset DGX_LIB; fi
  rm -rf /tmp/abk_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$tag -o p -- python $R/bench.py --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/abk_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"; python - "$f" "$2" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-70.70s calls %6s avg %9.1f ns" % (r["Name"], r["Calls"], float(r["AverageNs"])))
PY
done
