"""Dev helper: the proposal decode inside a step, from a rocprofv3 kernel trace of bench.py -- every launch between cn_scores_kernel and
cn_finalize_kernel of the last steps with its start offset, duration and the gap to the launch before it, then the step length
(optimizer launch to optimizer launch) and the GPU's idle time per step.

    python tools/decode_window.py <kernel_trace.csv> [steps]"""
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
cuts = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r[2]]
for c0, c1 in list(zip(cuts[:-1], cuts[1:]))[-steps:]:
    seg = rows[c0 + 1:c1 + 1]
    t0 = rows[c0][1]
    busy, end = 0, t0
    for s, e, _ in seg:
        busy += max(0, e - max(s, end))
        end = max(end, e)
    print("step %.3f ms, GPU busy (union) %.3f ms, idle %.3f ms, %d launches" % ((rows[c1][1] - t0) / 1e6, busy / 1e6, (rows[c1][1] - t0 - busy) / 1e6, len(seg)))
    a = next((i for i, r in enumerate(seg) if "cn_scores_kernel" in r[2]), None)
    z = next((i for i, r in enumerate(seg) if "cn_finalize_kernel" in r[2]), None)
    if a is None or z is None:
        continue
    base = seg[a][0]
    print("  decode window %.1f us (cn_scores start -> cn_finalize end), starts %.3f ms into the step" % ((seg[z][1] - base) / 1e3, (base - t0) / 1e6))
    prev = seg[a - 1][1] if a else base
    for s, e, n in seg[a:z + 1]:
        print("    +%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - base) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n[:90]))
        prev = max(prev, e)
    # the 12 launches behind the window: what waits for the proposals
    for s, e, n in seg[z + 1:z + 9]:
        print("    +%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - base) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n[:90]))
        prev = max(prev, e)
