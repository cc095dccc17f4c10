"""Dev helper: launches of a step that cannot fill the chip -- fewer workgroups than CUs -- and still take long: candidates for serial
or latency-bound code (the NMS sweep's resolver loop was found this way).  From a rocprofv3 kernel trace of bench.py.

    python tools/small_grid_kernels.py <kernel_trace.csv> [min_us]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
seg = rows[cuts[-2] + 1:cuts[-1] + 1]
agg = defaultdict(lambda: [0, 0.0, 0])
for r in seg:
    wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    n = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if n < 256 and us >= min_us:
        a = agg[(r["Kernel_Name"][:100], n, wg)]
        a[0] += 1
        a[1] += us
print("launches of the last step with < 256 workgroups and >= %.0f us:" % min_us)
tot = 0.0
for (name, n, wg), (c, us, _) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us  x%-3d %4d workgroups of %4d  %s" % (us, c, n, wg, name))
    tot += us
print("total %.1f us" % tot)
