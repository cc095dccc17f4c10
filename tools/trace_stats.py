"""Per-kernel totals over the STEADY-STATE steps of a rocprofv3 kernel trace of bench.py, in the column layout of rocprofv3's own
kernel_stats CSV (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev) so that tools/prof_summary.py and
bench.py's `profile_crosscheck` read it unchanged.  Steps are cut at the fused optimizer kernel (one `adamw_ema_kernel` per step);
the first `skip` steps -- warm-up, including the hipGraph capture passes of the FPN / tower segments, which rocprofv3's --stats
averages in -- are dropped.

    python tools/trace_stats.py <kernel_trace.csv> <skip steps> <out.csv>      prints the number of steps counted
"""
import csv
import math
import sys
from collections import defaultdict

path, skip, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
cuts = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r[2]]
assert len(cuts) > skip + 1, "not enough optimizer steps in the trace"
lo, hi = cuts[skip] + 1, cuts[-1] + 1          # kernels behind the skip-th optimizer launch, up to and including the last one
steps = len(cuts) - 1 - skip
agg = defaultdict(list)
for s, e, n in rows[lo:hi]:
    agg[n].append(e - s)
tot = sum(sum(v) for v in agg.values())
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        m = sum(v) / len(v)
        sd = math.sqrt(sum((x - m) ** 2 for x in v) / len(v))
        w.writerow([n, len(v), sum(v), "%.6f" % m, "%.4f" % (100.0 * sum(v) / tot), min(v), max(v), "%.3f" % sd])
print(steps)
