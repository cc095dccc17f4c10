"""Static scan of the gfx950 ISA of every libdgx kernel for memory loads whose latency is exposed at once: a vector-memory load followed
within a few instructions by `s_waitcnt vmcnt(0)` (round 5 found the attention backward and the LayerNorm kernels serialising their
loads that way: a select / conversion next to the load makes the compiler wait on the spot).
    python tools/isa_wait_scan.py [file.hip ...]     (cross-compiles; no GPU needed)"""
import os
import re
import subprocess
import sys

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "divergen_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(HERE) if f.endswith(".hip") and f != "abi.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wno-unused-result", "-S", "--cuda-device-only"]
for f in files:
    out = "/tmp/_scan_%s.s" % f
    if subprocess.call(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(HERE, f), "-o", out], stderr=subprocess.DEVNULL) != 0:
        print(f, "compile failed")
        continue
    name, body = None, []
    kernels = {}
    for ln in open(out):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            kernels[name] = body
        elif name is not None:
            t = ln.strip()
            if t and not t.startswith((";", ".")):
                body.append(t)
    for k, ins in kernels.items():
        loads = [i for i, t in enumerate(ins) if re.match(r"(global_load|buffer_load|flat_load)", t) and " lds" not in t]
        if not loads:
            continue
        hits = 0
        for i in loads:
            for t in ins[i + 1:i + 5]:
                if re.match(r"(global_load|buffer_load|flat_load)", t):
                    break
                if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                    hits += 1
                    break
        if hits >= 2:
            dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print("%-22s %3d of %3d loads waited for at once   %s" % (f, hits, len(loads), dem[:110]))
