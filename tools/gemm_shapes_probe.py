"""Per-shape timing of the forward / input-gradient GEMMs of the Swin-L CenterNet2 step (1024 px, 2 images):
the library path (torch.addmm / torch.mm -> hipBLASLt with the tuned table) next to libdgx's own MFMA GEMM
(dgx_gemm_bf16_nt) when the loaded library exports it.  Run on the GPU box:

    python tools/gemm_shapes_probe.py [--own-only] [--json gpurun_out/gemm_shapes.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _dev  # noqa: E402
_dev.apply_env()       # DGX_GEMM_LW / DGX_GEMM_TILE / DGX_WGRAD_LW ... of the calling script -> dgx_dev_set



def swin_shapes(size=1024, batch=2, embed=192, ws=12, depths=(2, 2, 18, 2)):
    """(name, M, N, K, count per step) of every Linear of the backbone: forward y = x W^T (M,K)x(N,K) and the
    input gradient dx = dy W (M,N)x(N,K) -> (M,K), which as an NT problem over the transposed weight is (M, K, N)."""
    out = []
    for s, d in enumerate(depths):
        C, H = embed * 2 ** s, size // 4 // 2 ** s
        T = batch * H * H
        Tw = batch * (-(-H // ws)) ** 2 * ws * ws
        for nm, M, N, K in (("qkv", Tw, 3 * C, C), ("proj", Tw, C, C), ("fc1", T, 4 * C, C), ("fc2", T, C, 4 * C)):
            out.append(("s%d.%s.fwd" % (s, nm), M, N, K, d))
            out.append(("s%d.%s.dgrad" % (s, nm), M, K, N, d))
        if s < 3:
            out.append(("s%d.merge.fwd" % s, T // 4, 2 * C, 4 * C, 1))
            out.append(("s%d.merge.dgrad" % s, T // 4, 4 * C, 2 * C, 1))
    return out


def time_fn(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--own-only", action="store_true")
    a = ap.parse_args()
    dev = "cuda"
    own = None
    try:
        from divergen_amd.layers import gemm_ops
        own = gemm_ops.gemm_nt
    except Exception as ex:  # library without the GEMM yet
        print("own GEMM not available:", ex)
    rows = []
    tot_lib = tot_own = tot_fl = 0.0
    g = torch.Generator(device=dev).manual_seed(0)
    for name, M, N, K, cnt in swin_shapes():
        x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        b = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
        fl = 2.0 * M * N * K
        r = {"name": name, "M": M, "N": N, "K": K, "count": cnt, "gflop": fl / 1e9}
        if not a.own_only:
            if name.endswith("dgrad"):
                wt = w.t().contiguous()          # (K, N): the forward weight; dx = dy @ W
                t = time_fn(lambda: torch.mm(x, wt))
            else:
                t = time_fn(lambda: torch.addmm(b, x, w.t()))
            r["lib_us"], r["lib_tflops"] = t * 1e6, fl / t / 1e12
            tot_lib += t * cnt
        if own is not None:
            y = own(x, w, None if name.endswith("dgrad") else b)
            ref = torch.mm(x.float(), w.float().t())
            if not name.endswith("dgrad"):
                ref = ref + b.float()
            err = float((y.float() - ref).abs().max() / ref.abs().max())
            t = gemm_ops.dev_time_us(x, w, None if name.endswith("dgrad") else b) * 1e-6
            r["own_us"], r["own_tflops"], r["own_relerr"] = t * 1e6, fl / t / 1e12, err
            tot_own += t * cnt
        tot_fl += fl * cnt
        rows.append(r)
        print(" ".join("%s=%s" % (k, ("%.4g" % v) if isinstance(v, float) else v) for k, v in r.items()), flush=True)
    summ = {"total_tflop_per_step": tot_fl / 1e12, "lib_ms_per_step": tot_lib * 1e3, "own_ms_per_step": tot_own * 1e3}
    print(summ)
    if a.json:
        os.makedirs(os.path.dirname(a.json), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump({"rows": rows, "summary": summ}, f, indent=1)


if __name__ == "__main__":
    main()
