"""Per GEMM shape: duration, effective shader clock and L2 hit rate from ONE rocprofv3 counter pass, joined with the per-launch
log of dgx_gemm_bf16_nt (DGX_GEMM_LOG).  Run once over bench.py (in situ) and once over tools/gemm_shapes_probe.py (the same
shapes back to back on hot operands) to see what separates the two:

    DGX_GEMM_LOG=/tmp/g.log rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/p -o p -- <cmd>
    python tools/gemm_insitu_pmc.py <counter_collection.csv> <kernel_trace.csv> /tmp/g.log <divide calls by>
"""
import csv
import sys
from collections import defaultdict

cc, kt, log, steps = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
dur = {}
for r in csv.DictReader(open(kt)):
    if "gemm_nt_kernel" in r["Kernel_Name"] or "gemm256_kernel" in r["Kernel_Name"]:
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
cnt = defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if "gemm_nt_kernel" in r["Kernel_Name"] or "gemm256_kernel" in r["Kernel_Name"]:
        cnt[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(set(dur) & set(cnt))
lines = [l.split() for l in open(log) if l.strip()]
print("dispatches %d, log lines %d" % (len(ids), len(lines)))
n = min(len(ids), len(lines))
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for d, l in zip(ids[-n:], lines[-n:]):
    a = agg[tuple(l)]
    c = cnt[d]
    a[0] += 1
    a[1] += dur[d]
    a[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
    a[3] += c.get("TCC_HIT_sum", 0.0)
    a[4] += c.get("TCC_MISS_sum", 0.0)
print("%8s %6s %6s mode tile      calls/step   avg_us   ms/step   TF/s   GHz  L2hit  L2req/launch" % ("M", "N", "K"))
tot = 0.0
for key, (c, us, cyc, hit, miss) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, mode, bm, bn = map(int, key[:6])
    tot += us
    print("%8d %6d %6d %4d %3dx%-3d   %8.1f %8.1f %9.3f %6.0f %5.2f %6.3f %10.0f" % (
        M, N, K, mode, bm, bn, c / steps, us / c, us / 1e3 / steps, 2.0 * M * N * K / (us / c) / 1e6,
        cyc / us / 1e4 if us else 0.0, hit / (hit + miss) if hit + miss else 0.0, (hit + miss) / c))
print("total %.2f ms/step" % (tot / 1e3 / steps))
