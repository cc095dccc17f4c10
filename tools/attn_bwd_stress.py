"""Dev helper: stress of the attention backward's table-gradient protocol (slot stores, per-head counters reset by the last run, one
workspace per stream): many launches of varying shapes back to back on one stream WITHOUT host syncs in between, each checked afterwards
against an fp32 torch evaluation of the same table gradient and against a second pass over the same inputs (bit-identical)."""
import sys
import torch
sys.path.insert(0, ".")
from divergen_amd import layers as la
from oracle import swin as OSW

dev = "cuda"
torch.manual_seed(0)
shapes = [(12, 72, 24), (12, 74, 24), (12, 13, 48), (12, 242, 12), (7, 103, 24), (7, 50, 3), (12, 5, 6), (12, 968, 6), (12, 18, 48), (7, 8, 1)]


def ref_dtable(qkv, table, go, nH, ws, scale):
    B_, N, _ = qkv.shape
    q, k, v = qkv.float().view(B_, N, 3, nH, 32).permute(2, 0, 3, 1, 4)
    idx = OSW.relative_position_index(ws).to(dev).reshape(-1)
    s = (q * scale) @ k.transpose(-2, -1) + table.float()[idx].reshape(N, N, nH).permute(2, 0, 1)[None]
    p = torch.softmax(s, -1)
    do = go.float().view(B_, N, nH, 32).permute(0, 2, 1, 3)
    dp = do @ v.transpose(-2, -1)
    ds = p * (dp - (dp * p).sum(-1, keepdim=True))
    out = torch.zeros(table.shape[0], nH, device=dev)
    out.index_add_(0, idx, ds.sum(0).permute(1, 2, 0).reshape(N * N, nH))
    return out


cases = []
for rep in range(6):
    for ws, B_, nH in shapes:
        N = ws * ws
        qkv = (torch.randn(B_, N, 3 * nH * 32, device=dev) * 1.5).to(torch.bfloat16)
        table = torch.randn((2 * ws - 1) ** 2, nH, device=dev)
        go = torch.randn(B_, N, nH * 32, device=dev).to(torch.bfloat16)
        cases.append((ws, B_, nH, qkv, table, go))
results = []
for pas in range(2):                      # no synchronisation inside a pass
    res = []
    for ws, B_, nH, qkv, table, go in cases:
        qd = qkv.clone().requires_grad_(True)
        td = table.clone().requires_grad_(True)
        out = la.window_attention_core(qd, td, None, 1, nH, ws, 32 ** -0.5)
        out.backward(go)
        res.append(td.grad)
    torch.cuda.synchronize()
    results.append(res)
bad = 0
for (ws, B_, nH, qkv, table, go), a, b in zip(cases, results[0], results[1]):
    r = ref_dtable(qkv, table, go, nH, ws, 32 ** -0.5)
    err = float((a - r).abs().max() / r.abs().max())
    same = torch.equal(a, b)
    if err > 0.015 or not same:
        bad += 1
        print("MISMATCH ws %d B_ %d nH %d: rel err %.3e, bit-identical %s" % (ws, B_, nH, err, same))
print("%d launches x 2 passes, %d bad" % (len(cases), bad))
sys.exit(1 if bad else 0)
