"""Dev helper: where the device idles inside a step, from a rocprofv3 kernel trace of bench.py -- the largest gaps (no launch of any
stream running) of the last steps with the launches on either side, and the idle total.

    python tools/step_gaps.py <kernel_trace.csv> [steps] [top]"""
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
cuts = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r[2]]
for c0, c1 in list(zip(cuts[:-1], cuts[1:]))[-steps:]:
    seg = rows[c0:c1 + 1]
    t0 = seg[0][1]
    end, gaps = t0, []
    prev = seg[0][2]
    for s, e, n in seg[1:]:
        if s > end:
            gaps.append((s - end, end - t0, prev, n))
        if e > end:
            end, prev = e, n
    tot = sum(g[0] for g in gaps)
    print("step %.3f ms, idle %.3f ms in %d gaps; %d gaps > 10 us = %.3f ms" % ((seg[-1][1] - t0) / 1e6, tot / 1e6, len(gaps), sum(1 for g in gaps if g[0] > 10000),
                                                                                  sum(g[0] for g in gaps if g[0] > 10000) / 1e6))
    for g, at, a, b in sorted(gaps, reverse=True)[:top]:
        print("  %7.1f us at +%6.3f ms   %-52s -> %s" % (g / 1e3, at / 1e6, a[:52], b[:60]))
