"""Dev helper: a rocprofv3 kernel trace grouped by (kernel, grid, workgroup, LDS): calls per step, average and total time.
usage: trace_by_grid.py <kernel_trace.csv> <steps> [name filter]"""
import csv
import sys
from collections import defaultdict

path, steps = sys.argv[1], float(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    key = (n.split("(")[0][-70:], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("LDS_Block_Size", ""))
    a = agg[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0.0
for key, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    tot += us
    print("%-72s grid %9s wg %5s lds %7s  calls/step %6.1f  avg %8.1f us  %7.3f ms/step" % (key + (c / steps, us / c, us / 1e3 / steps)))
print("shown total %.2f ms/step" % (tot / 1e3 / steps))
