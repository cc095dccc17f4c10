"""Join the per-launch log of dgx_gemm_bf16_nt (DGX_GEMM_LOG) with a rocprofv3 kernel trace: in-situ time per GEMM shape.
usage: gemm_insitu.py <kernel_trace.csv> <gemm.log> <steps>"""
import csv
import sys
from collections import defaultdict

trace, log, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
rows = [r for r in csv.DictReader(open(trace)) if "gemm_nt_kernel" in r["Kernel_Name"] or "gemm256_kernel" in r["Kernel_Name"] or "gemm_lw_kernel" in r["Kernel_Name"] or "gemm_k192_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lines = [l.split() for l in open(log) if l.strip()]
print("dispatches %d, log lines %d" % (len(rows), len(lines)))
n = min(len(rows), len(lines))
agg = defaultdict(lambda: [0, 0.0])
for r, l in zip(rows[-n:], lines[-n:]):
    kn = r["Kernel_Name"]
    tag = ""
    if "gemm256" in kn:
        tag = "g%s" % kn.split("<")[1].split(",")[0].split(">")[0]
    elif "gemm_k192" in kn:
        tag = "K192<%s>" % kn.split("<")[1].split(">")[0].replace(" ", "")
    elif "gemm_lw" in kn:
        a = kn.split("<")[1].split(">")[0].split(",")
        tag = "L%sx%s" % (a[0].strip(), a[1].strip())
    key = tuple(l) + (tag,)
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0.0
print("%8s %6s %6s mode tile      calls/step   avg_us   ms/step   TF/s" % ("M", "N", "K"))
for key, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, mode, bm, bn = map(int, key[:6])
    tot += us
    print("%8d %6d %6d %4d %7s   %8.1f %8.1f %9.3f %6.0f" % (M, N, K, mode, key[6] or "%dx%d" % (bm, bn), c / steps, us / c, us / 1e3 / steps, 2.0 * M * N * K / (us / c) / 1e6))
print("total %.2f ms/step" % (tot / 1e3 / steps))
