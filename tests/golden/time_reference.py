"""Container-only (needs /root/reference; never imported by tests, bench.py or smoke()): CPU time of the REFERENCE's own
Swin backbone file against oracle/swin.py on the same weights and the same input, forward + backward, same thread count.

    python tests/golden/time_reference.py [--swin T|S|B|L-22k-384] [--size 256] [--threads 8] [--repeat 3]

This is the check behind `cpu_baseline.kind = "port"` in bench.py: the oracle is a restatement, so its CPU time is only
a fair stand-in for the reference's CPU path if the two agree.  Prints both times, their ratio and the max output
difference; exits non-zero when the restatement is more than 10 % away from the reference file's time.
Reference: DG/divergen/modeling/backbone/swintransformer.py (SwinTransformer.forward :560-590, loaded through
tests/golden/_refload.py's stub-import harness; drop_path_rate 0 so both sides are deterministic)."""
import argparse
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--swin", default="T")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 1, 32))
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    import _refload as R
    from oracle import swin as OSW
    from tests._recipes import fill_state, swin_param_shapes
    sw = R.ref("divergen.modeling.backbone.swintransformer")
    c = OSW.SIZE2CONFIG[a.swin]
    net = sw.SwinTransformer(embed_dim=c["embed_dim"], depths=c["depths"], num_heads=c["num_heads"], window_size=c["ws"],
                             drop_path_rate=0.0, out_indices=(1, 2, 3))
    p = fill_state(swin_param_shapes(c["embed_dim"], c["depths"], c["num_heads"], c["ws"]), 7, 0.02)
    with torch.no_grad():
        for k, v in net.named_parameters():
            v.copy_(p[k])
    for v in p.values():
        v.requires_grad_(True)
    img = torch.randn(1, 3, a.size, a.size)

    def run_ref():
        net.zero_grad()
        t0 = time.time()
        outs = net(img)
        sum(o.square().mean() for o in outs.values()).backward()
        return time.time() - t0, outs

    def run_oracle():
        for v in p.values():
            v.grad = None
        t0 = time.time()
        outs = OSW.swin_forward(img, p, c["embed_dim"], c["depths"], c["num_heads"], c["ws"])
        sum(o.square().mean() for o in outs.values()).backward()
        return time.time() - t0, outs

    run_ref(), run_oracle()          # warm-up (allocator, thread pool)
    tr = min(run_ref()[0] for _ in range(a.repeat))
    to = min(run_oracle()[0] for _ in range(a.repeat))
    _, o_ref = run_ref()
    _, o_orc = run_oracle()
    diff = max(float((o_ref[k] - o_orc[k]).abs().max()) for k in o_ref)
    gdiff = max(float((dict(net.named_parameters())[k].grad - v.grad).abs().max()) for k, v in p.items() if v.grad is not None)
    ratio = to / tr
    print("Swin-%s %dx%d, %d threads, best of %d:  reference file %.3f s   oracle %.3f s   oracle/reference = %.3f   "
          "max |out diff| = %.2e   max |param-grad diff| = %.2e" % (a.swin, a.size, a.size, a.threads, a.repeat, tr, to, ratio, diff, gdiff))
    return 0 if abs(ratio - 1.0) <= 0.10 else 1


if __name__ == "__main__":
    sys.exit(main())
