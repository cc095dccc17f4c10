import torch, time
a = torch.randn(512, 4096, device="cuda", dtype=torch.bfloat16); b = torch.randn(4096, 256, device="cuda", dtype=torch.bfloat16)
acc = torch.zeros(512, 256, device="cuda")
try:
    c = torch.mm(a, b, out_dtype=torch.float32); print("mm out_dtype ok", c.dtype, (c - a.float() @ b.float()).abs().max().item())
except Exception as e: print("mm out_dtype FAIL", repr(e)[:200])
try:
    c = torch.addmm(acc, a, b, out_dtype=torch.float32); print("addmm out_dtype ok", c.dtype)
except Exception as e: print("addmm out_dtype FAIL", repr(e)[:200])
try:
    torch.addmm(acc, a, b, out_dtype=torch.float32, out=acc); print("addmm out= ok", acc.abs().max().item())
except Exception as e: print("addmm out= FAIL", repr(e)[:200])
# transposed operand (wgrad pattern)
dy = torch.randn(4096, 512, device="cuda", dtype=torch.bfloat16); x = torch.randn(4096, 256, device="cuda", dtype=torch.bfloat16)
try:
    g = torch.mm(dy.t(), x, out_dtype=torch.float32); print("wgrad out_dtype ok", (g - dy.float().t() @ x.float()).abs().max().item())
except Exception as e: print("wgrad FAIL", repr(e)[:200])
try:
    s = torch.sum(dy, 0, dtype=torch.float32); print("sum dtype ok", s.dtype)
except Exception as e: print("sum FAIL", e)
