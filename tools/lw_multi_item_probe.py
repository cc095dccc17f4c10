"""gemm_lw with more than one item per workgroup, per tail mode: each case in its own process (a memory fault aborts the process)."""
import os
import subprocess
import sys

CASE = r'''
import sys, torch
sys.path.insert(0, %r)
from divergen_amd import _lib as L
from divergen_amd.layers import gemm_ops as G
M, N, K, mode, reserved, lw = %d, %d, %d, %d, %d, %d
BF = torch.bfloat16
L.lib().dgx_set_reserved_cus(reserved)
L.lib().dgx_dev_set(b"gemm_lw", lw)
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(M, K, device="cuda", generator=g).to(BF); b = (torch.randn(N, K, device="cuda", generator=g) * 0.06).to(BF)
bias = torch.randn(N, device="cuda", generator=g).to(BF)
y = a.float() @ b.float().t() + bias.float()
if mode == 1:
    out = G.gemm_nt(a, b, bias); ref = y
elif mode == 2:
    out, act = G.gemm_bias_gelu(a, b, bias); ref = y
elif mode == 3 or mode == 6:
    rdt = BF if mode == 3 else torch.float32
    res = torch.randn(2, M // 2, N, device="cuda", generator=g).to(rdt)
    out = G.gemm_bias_residual(a, b, bias, res, None, 2, M // 2, 1, 0, 0); ref = (res.float() + y.to(BF).float().reshape(2, M // 2, N))
elif mode == 4:
    f1 = torch.randn(M, N, device="cuda", generator=g).to(BF)
    out = G.gemm_gelu_grad(a, b, f1); ref = None
torch.cuda.synchronize()
bm, bn, sp = L.c_i(), L.c_i(), L.c_i()
form = L.lib().dgx_gemm_last_form(bm, bn, sp)
err = float((out.float() - ref.reshape(out.shape)).abs().max()) if ref is not None else -1
print("ok form %%d %%dx%%d err %%.3g" %% (form, bm.value, bn.value, err))
'''
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for (M, N, K) in [(8192, 768, 3072), (16384, 768, 3072), (8192, 768, 768), (10368, 2304, 768)]:
    for mode in (1, 2, 3, 6, 4):
        for reserved in (0, 16):
            r = subprocess.run([sys.executable, "-c", CASE % (ROOT, M, N, K, mode, reserved, 1)], capture_output=True, text=True, timeout=300)
            print((M, N, K), "mode", mode, "reserved", reserved, "->", r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "rc %d %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:150] if r.stderr.strip() else ""), flush=True)
