"""Dev helper: GPU idle time inside the training steps of a rocprofv3 kernel trace: the union of all kernel intervals per step
(steps are cut at the fused optimizer kernel), the idle remainder, its split by gap size, and the largest gaps with the kernels
on either side.   usage: trace_gaps.py <kernel_trace.csv> [skip_steps]"""
import csv
import sys
from collections import Counter

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]) for r in csv.DictReader(open(sys.argv[1]))]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort()
cuts = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r[2]]
print("kernels %d, optimizer steps %d" % (len(rows), len(cuts)))
tot_busy = tot_idle = tot_len = 0.0
hist = Counter()
big = []
nst = 0
for a, b in zip(cuts[skip:-1], cuts[skip + 1:]):
    seg = rows[a + 1:b + 1]
    t0, cur_end = rows[a][1], rows[a][1]
    busy = 0.0
    prev = rows[a][2]
    for s, e, n in seg:
        if s > cur_end:
            gap = (s - cur_end) / 1e3
            hist["<2us" if gap < 2 else "2-5us" if gap < 5 else "5-20us" if gap < 20 else "20-100us" if gap < 100 else ">100us"] += gap
            big.append((gap, prev, n))
            busy += (e - s)
            cur_end = e
        elif e > cur_end:
            busy += (e - cur_end)
            cur_end = e
        if e >= cur_end:
            prev = n
    tot_busy += busy / 1e6
    tot_len += (cur_end - t0) / 1e6
    nst += 1
print("per step: length %.2f ms, busy (union) %.2f ms, idle %.2f ms" % (tot_len / nst, tot_busy / nst, (tot_len - tot_busy) / nst))
for k in ("<2us", "2-5us", "5-20us", "20-100us", ">100us"):
    print("  idle in gaps %-9s %.3f ms/step" % (k, hist[k] / 1e3 / nst))
# idle per segment of the step, cut at marker kernels
marks = ["roi_label_kernel", "roi_gather_kernel", "mask_bce_partial_kernel", "wgrad256_partial_kernel", "win_attn_bwd_kernel", "adamw_ema_kernel"]
seg_idle = {}
seg_len = {}
for a, b in zip(cuts[skip:-1], cuts[skip + 1:]):
    seg = rows[a + 1:b + 1]
    cur_end = rows[a][1]
    label, seg_start = "start", rows[a][1]
    seen = set()
    for s, e, n in seg:
        for m in marks:
            if m in n and m not in seen:
                seen.add(m)
                seg_len[label] = seg_len.get(label, 0.0) + (s - seg_start) / 1e3
                label, seg_start = "from " + m, s
        if s > cur_end:
            seg_idle[label] = seg_idle.get(label, 0.0) + (s - cur_end) / 1e3
        cur_end = max(cur_end, e)
    seg_len[label] = seg_len.get(label, 0.0) + (cur_end - seg_start) / 1e3
print("idle per segment (first occurrence of each marker kernel in the step):")
for k in ["start"] + ["from " + m for m in marks]:
    if k in seg_len:
        print("  %-34s length %7.2f ms  idle %6.3f ms" % (k, seg_len[k] / 1e3 / nst, seg_idle.get(k, 0.0) / 1e3 / nst))
pair = {}
for g, p_, n_ in big:
    a_ = pair.setdefault((p_, n_), [0, 0.0])
    a_[0] += 1
    a_[1] += g
print("idle by (kernel before -> kernel after), us per step (count per step):")
for (p_, n_), (c_, g_) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %8.1f  (%4.1f)  %s -> %s" % (g_ / nst, c_ / nst, p_, n_))
print("largest gaps (us, kernel before -> kernel after):")
for g, p, n in sorted(big, reverse=True)[:12]:
    print("  %8.1f  %s -> %s" % (g, p, n))
