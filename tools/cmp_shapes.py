"""Dev helper: side-by-side of tools/gemm_shapes_probe.py outputs.  python tools/cmp_shapes.py base.txt a.txt [b.txt ...]"""
import sys


def load(f):
    d = {}
    for line in open(f, errors="replace"):
        if line.startswith("name="):
            kv = dict(x.split("=") for x in line.split())
            d[kv["name"]] = (float(kv["own_us"]), int(kv["count"]), kv["M"], kv["N"], kv["K"])
    return d


files = sys.argv[1:]
data = [load(f) for f in files]
tot = [0.0] * (len(files) + 1)
print("%-16s %7s %5s %5s " % ("name", "M", "N", "K") + " ".join("%9s" % f.split("/")[-1][:9] for f in files))
for k in data[0]:
    us = [d[k][0] if k in d else float("nan") for d in data]
    c = data[0][k][1]
    for i, u in enumerate(us):
        tot[i] += u * c
    tot[-1] += min(us) * c
    print("%-16s %7s %5s %5s " % ((k,) + data[0][k][2:]) + " ".join("%9.1f" % u for u in us) + "   best=%d" % us.index(min(us)))
print("ms/step: " + "  ".join("%.2f" % (t / 1e3) for t in tot[:-1]) + "   best-of %.2f" % (tot[-1] / 1e3))
