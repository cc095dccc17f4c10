"""bench.py -- images/sec of the DiverGen training step (Swin-L CenterNet2, 1024 px, LVIS-shaped
synthetic data + copy-paste) on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = for the rank's 2 images: GPU copy-paste compositor (19 pastes/image) -> forward ->
backward (gradient all-reduce over RCCL overlapped) -> fused clip+AdamW+EMA.  Inputs are resident in
HBM when the timed region starts.  Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from divergen_amd.tuning import enable as _enable_tuned_gemm  # noqa: E402  (no torch import inside)
_enable_tuned_gemm()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=12)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--size", type=int, default=1024)
    p.add_argument("--swin", default="L-22k-384")
    p.add_argument("--batch", type=int, default=2, help="images per GPU (IMS_PER_BATCH 16 / 8 GPUs)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-copy-paste", action="store_true")
    return p.parse_args()


def make_pastes(rng, size, k=19):
    out = []
    for i in range(k):
        s = int(rng.uniform(0.05, 0.3) * size)
        rgba = rng.integers(0, 256, (s, s, 4), dtype=np.uint8)
        yy, xx = np.mgrid[0:s, 0:s]
        rgba[..., 3] *= ((((xx - s / 2) / (s / 2)) ** 2 + ((yy - s / 2) / (s / 2)) ** 2) <= 1).astype(np.uint8)
        out.append((rgba, int(rng.integers(-s // 2, size - s // 2)), int(rng.integers(-s // 2, size - s // 2)),
                    int(rng.integers(1203, 1453))))
    return out


# HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc, corrected as the
# MI355X guide prescribes); None where no pass was taken.  Filled from profiles/r01_pmc.json when present.
PMC_TRAFFIC = {}
try:
    with open(os.path.join(ROOT, "profiles", "r01_pmc.json")) as _f:
        PMC_TRAFFIC = {k: v.get("hbm_bytes_per_launch") for k, v in json.load(_f).items() if isinstance(v, dict)}
except (OSError, ValueError):
    pass


class KernelTimer:
    """HIP events around one libdgx entry point, recorded on the stream the kernel is launched on."""

    def __init__(self, lib, name):
        self.lib, self.name, self.events, self.enabled = lib, name, [], False
        self.orig = getattr(lib, name)

        def wrapped(*a):
            if not self.enabled:
                return self.orig(*a)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self.orig(*a)
            e.record()
            self.events.append((s, e, a))
            return r
        wrapped.argtypes, wrapped.restype = self.orig.argtypes, self.orig.restype
        setattr(lib, name, wrapped)


def cpu_baseline(swin, budget_s=25.0):
    """Oracle (CPU restatement of the reference backbone, pinned on the reference's goldens) timed on
    this box's host cores: Swin fwd+bwd on ONE image, at the largest power-of-two size that fits the
    time budget; reported as 1024^2-equivalent images/s by token count."""
    from oracle import swin as OSW
    from tests._recipes import fill_state, swin_param_shapes
    c = OSW.SIZE2CONFIG[swin]
    cores = min(os.cpu_count() or 1, 32)   # more threads than this only adds contention on these GEMM sizes
    torch.set_num_threads(cores)
    p = fill_state(swin_param_shapes(c["embed_dim"], c["depths"], c["num_heads"], c["ws"]), 7, 0.02)
    for v in p.values():
        v.requires_grad_(True)
    size, t = 256, None
    t_start = time.time()
    while True:
        img = torch.randn(1, 3, size, size)
        t0 = time.time()
        outs = OSW.swin_forward(img, p, c["embed_dim"], c["depths"], c["num_heads"], c["ws"])
        sum(o.square().mean() for o in outs.values()).backward()
        t = time.time() - t0
        # next size costs ~4x (token count); stop when it would not fit the budget
        if t * 4 + (time.time() - t_start) > budget_s or size >= 1024:
            break
        size *= 2
    scale = (1024.0 / size) ** 2
    return {"value": 1.0 / (t * scale), "unit": "images/s (1024^2-equivalent, Swin backbone fwd+bwd only)",
            "cores": cores, "kind": "port",
            "sample": "oracle/swin.py Swin-%s fwd+bwd (fp32, %d threads), 1 image %dx%d in %.1f s, x%.0f pixel-count scaling to 1024^2" % (swin, cores, size, size, t, scale)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DGX_FORCE_DEVICE / DGX_DIST_BACKEND exist for ONE purpose: exercising the multi-rank code path (reducer hooks,
    # graph capture next to collectives, CenterNet normaliser all-reduces) on a single-GPU box with gloo
    if "DGX_FORCE_DEVICE" in os.environ:
        local = int(os.environ["DGX_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("DGX_FORCE_PG") == "1":      # DGX_FORCE_PG: a 1-rank RCCL group, to test RCCL next to graph capture
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("DGX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == a.gpus, "launch one process per GPU (WORLD_SIZE=%d, --gpus %d)" % (world, a.gpus)
    # DGX_GRAPH_BACKBONE=1 (opt-in) replays the static-shape backbone fwd+bwd as a hipGraph: -6 % step
    # time at N=1 (the step is CPU-launch-bound), but per-kernel HIP events (the roofline object) and the
    # per-layer gradient readiness the arena reducer overlaps on are only available on the eager path,
    # which is therefore the default and what `value` reports.
    os.environ.setdefault("DGX_GRAPH_BACKBONE", "0")

    from divergen_amd import _lib
    from divergen_amd import layers as la
    from divergen_amd.config import get_cfg
    from divergen_amd.data import synthetic_batch
    from divergen_amd.engine import ArenaReducer
    from divergen_amd.modeling import build_model
    from divergen_amd.solver import build_lr_scheduler, build_optimizer
    from divergen_amd.structures import BitMasks, Boxes, Instances
    from divergen_amd.utils.events import EventStorage

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "tests", "configs", "DiverGen_swinL.yaml"))
    cfg.merge_from_list(["MODEL.SWIN.SIZE", a.swin, "INPUT.TRAIN_SIZE", a.size, "MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH",
                         os.path.join(ROOT, "tests", "configs", "metadata",
                                      "ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json")])
    torch.manual_seed(cfg.SEED + rank)
    model = build_model(cfg).train()
    opt = build_optimizer(cfg, model)
    sched = build_lr_scheduler(cfg, opt)
    reducer = ArenaReducer(opt.arena)
    reducer.broadcast_parameters()
    if opt.ema is not None:
        opt.ema.copy_(opt.arena.p)
    nparams = sum(p.numel() for p in model.parameters())

    base = synthetic_batch(a.batch, a.size, cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=1234 + rank, device=dev)
    rng = np.random.default_rng(7 + rank)
    pastes = [make_pastes(rng, a.size) for _ in range(a.batch)]
    # pre-stage paste patches as device tensors so the timed region starts with inputs resident in HBM
    pastes = [[(torch.from_numpy(r).to(dev), x, y, l) for r, x, y, l in ps] for ps in pastes]

    # The copy-paste compositor is the data-loading side of the step (the reference runs it in loader workers,
    # DG/divergen/data/custom_build_copypaste_mapper.py): it depends on nothing the optimizer produces, so the batch of
    # step t+1 is composited on a side HIP stream while step t trains, and its one data-dependent shape (objects that
    # end up fully covered are dropped) is read back from THAT stream instead of draining the training stream.
    # Every step still composites exactly one batch inside the timed region.
    side = torch.cuda.Stream()

    def compose():
        batch = []
        with torch.cuda.stream(side):
            for d, ps in zip(base, pastes):
                inst = d["instances"]
                if a.no_copy_paste:
                    batch.append(d)
                    continue
                out = la.copy_paste(d["image"], inst.gt_masks.tensor.view(torch.uint8), inst.gt_boxes.tensor,
                                    inst.gt_classes, ps)
                ni = Instances(inst.image_size)
                ni.gt_boxes, ni.gt_classes = Boxes(out["boxes"]), out["labels"]
                ni.gt_masks, ni.instance_source = BitMasks(out["masks"]), out["source"]
                batch.append({"image": out["image"], "instances": ni, "height": d["height"], "width": d["width"],
                              "file_name": d["file_name"]})
            ev = torch.cuda.Event()
            ev.record(side)
        return batch, ev

    nxt = [compose()]

    def one_step():
        batch, ev = nxt[0]
        torch.cuda.current_stream().wait_event(ev)      # the training stream consumes the composited tensors
        for d in batch:                                  # ... and owns them from here on (allocator stream bookkeeping)
            d["image"].record_stream(torch.cuda.current_stream())
            if "instances" in d and d["instances"].has("gt_masks"):
                d["instances"].gt_masks.tensor.record_stream(torch.cuda.current_stream())
        opt.zero_grad()
        losses = model(batch)
        nxt[0] = compose()                               # next batch, concurrently with this step's backward
        total = sum(losses.values())
        total.backward()
        scale = reducer.finish()
        opt.step(grad_scale=scale)
        sched.step()
        return total

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    L = _lib.lib()
    timers = {n: KernelTimer(L, n) for n in ("dgx_linear_wgrad_grouped", "dgx_window_attention_fwd", "dgx_window_attention_bwd")}
    with EventStorage(0):
        for _ in range(a.warmup):
            one_step()
        sync()
        for t in timers.values():
            t.enabled = True
        t0 = time.perf_counter()
        for _ in range(a.steps):
            total = one_step()
        t_issue = time.perf_counter() - t0     # host time to ENQUEUE the steps (diagnostic: CPU- vs GPU-bound)
        sync()
        dt = time.perf_counter() - t0
    assert bool(torch.isfinite(total)), "non-finite loss"
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    imgs = a.batch * world * a.steps
    # After the timed region: every rank applied the same all-reduced gradients with the same deterministic optimizer
    # kernel, so the weights must still be IDENTICAL across ranks.  A bucket reduced before its last gradient write (or
    # any rank-local contribution that escaped the all-reduce) shows up here as diverged weights: fail loudly.
    in_sync = None
    if world > 1:
        chk = torch.stack([opt.arena.p.double().sum(), opt.arena.p.double().abs().sum()]).cpu()
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk.to(dev) if dist.get_backend() == "nccl" else chk)
        in_sync = all(torch.equal(g.cpu(), gathered[0].cpu()) for g in gathered)
        assert in_sync, "weights diverged across ranks: %s" % [g.tolist() for g in gathered]

    # roofline objects from HIP events recorded (on the launch stream) around every call of the three heaviest
    # libdgx entry points inside the timed region; "roofline" = the one with the largest total time
    def work(name, args):
        """-> (flops, algorithmic bytes) of one call."""
        if name == "dgx_linear_wgrad_grouped":
            fl = by = 0.0
            for i in range(args[1]):
                p = args[0][i]
                fl += 2.0 * p.M * p.Nn * p.Kk
                by += 2.0 * p.M * (p.Nn + p.Kk) + 8.0 * p.Nn * p.Kk      # dY, X read once (bf16); fp32 gradient read + written
            return fl, by
        if name.endswith("fwd"):
            B_, nH, ws, mm, io = args[7], args[9], args[10], 2, 4            # QK^T, PV;  q,k,v in + out
        else:
            B_, nH, ws, mm, io = args[10], args[12], args[13], 5, 8         # S, dP, dV, dK, dQ;  q,k,v,o,do in + dq,dk,dv out
        N = ws * ws
        return B_ * nH * mm * 2.0 * N * N * 32, B_ * nH * N * (32 * 2 * io + 4)

    objs = []
    for name, t in timers.items():
        if not t.events:
            continue
        tot_ms = sum(s.elapsed_time(e) for s, e, _ in t.events)
        fl = by = 0.0
        for _, _, args in t.events:
            f, b_ = work(name, args)
            fl += f
            by += b_
        tf, gbs = fl / (tot_ms * 1e-3) / 1e12, by / (tot_ms * 1e-3) / 1e9
        # window attention at head_dim 32 is 72 FLOP/B -- under the ~310 FLOP/B ridge, i.e. HBM-bound; the 256x256
        # weight-gradient tiles are MFMA-bound
        mfma = name == "dgx_linear_wgrad_grouped"
        objs.append({"kernel": name, "bound": "mfma" if mfma else "hbm", "achieved": tf if mfma else gbs,
                     "peak": 2500.0 if mfma else 8000.0, "unit": "TFLOP/s" if mfma else "GB/s",
                     "frac": (tf / 2500.0) if mfma else (gbs / 8000.0), "traffic": PMC_TRAFFIC.get(name),
                     "avg_launch_us": tot_ms * 1e3 / len(t.events), "launches": len(t.events), "total_ms_per_step": tot_ms / a.steps,
                     "tflops": tf, "algorithmic_gbytes_per_s": gbs})
    objs.sort(key=lambda o: -o["total_ms_per_step"])
    roof = objs[0] if objs else None

    if rank == 0:
        line = {"metric": "images/sec (node) Swin-L CenterNet2 LVIS 1024px", "value": imgs / dt, "unit": "images/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                "data": "synthetic (random-init weights, LVIS-shaped boxes/masks, 19 RGBA pastes per image)",
                "config": {"workload": "CenterNet2 Swin-%s, %dx%d, %d images/GPU, 1453 classes, GPU copy-paste + fwd + bwd + "
                                       "fused clip/AdamW/EMA; configs/DiverGen_swinL.yaml" % (a.swin, a.size, a.size, a.batch),
                           "global_batch": a.batch * world, "parallelism": "dp%d" % world, "params_M": nparams / 1e6},
                "roofline": roof, "roofline_other": objs[1:],
                "host_issue_ms_per_step": t_issue / a.steps * 1e3,
                "peak_hbm_gb_rank0": torch.cuda.max_memory_allocated(dev) / 1e9}
        if in_sync is not None:
            line["weights_identical_across_ranks"] = in_sync
            line["buckets_reduced_during_backward"] = "%d/%d" % (reducer.last_early, len(reducer.buckets))
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.swin)
    # RCCL writes its version banner to the C-level stdout, which is block-buffered when redirected and would otherwise be
    # flushed at exit, i.e. AFTER the JSON: every rank drains it (and they meet) before rank 0 prints, so that the JSON
    # line is the last line on the job's stdout.
    def drain():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
    drain()
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    drain()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
