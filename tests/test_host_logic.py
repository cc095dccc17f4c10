"""CPU tests of the host-side mirror: config surface, registries / state-dict key names, LR schedule,
arena bucketing, and the world_size-2 gloo path of the arena reducer."""
import os
import sys

import pytest
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs")
FREQ = os.path.join(CFG, "metadata", "ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json")


def _cfg(name="DiverGen_swinL.yaml", opts=()):
    from divergen_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(CFG, name))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH", FREQ] + list(opts))
    return cfg


def test_reference_yamls_load_unchanged():
    cfg = _cfg()
    assert cfg.MODEL.META_ARCHITECTURE == "CustomRCNN" and cfg.MODEL.PROPOSAL_GENERATOR.NAME == "CenterNet"
    assert cfg.MODEL.ROI_HEADS.NAME == "DeticCascadeROIHeads" and cfg.MODEL.ROI_HEADS.NUM_CLASSES == 1453
    assert cfg.MODEL.BACKBONE.NAME == "build_swintransformer_fpn_backbone" and cfg.MODEL.SWIN.SIZE == "L-22k-384"
    assert cfg.SOLVER.BASE_LR == 1e-4 and cfg.SOLVER.MODEL_EMA == 0.999 and cfg.SOLVER.IMS_PER_BATCH == 16
    assert cfg.INPUT.TRAIN_SIZE == 896 and cfg.INPUT.INST_POOL_SAMPLE_TYPE == "cas_random" and cfg.FP16 is True
    assert cfg.MODEL.CENTERNET.POST_NMS_TOPK_TRAIN == 2000 and cfg.MODEL.ROI_BOX_CASCADE_HEAD.IOUS == [0.6, 0.7, 0.8]
    base = _cfg("baseline_swinL.yaml")
    assert base.MODEL.ROI_HEADS.NUM_CLASSES == 1203 and base.INPUT.INST_POOL is False
    with pytest.raises(KeyError):
        cfg.merge_from_list(["MODEL.NO_SUCH_KEY", 1])
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.SEED = 1


def test_registry_build_and_state_dict_keys():
    from divergen_amd.modeling import build_model
    model = build_model(_cfg(opts=["MODEL.SWIN.SIZE", "T"]))
    keys = set(model.state_dict().keys())
    for k in ["backbone.bottom_up.layers.2.blocks.5.attn.qkv.weight",
              "backbone.bottom_up.layers.0.blocks.1.attn.relative_position_bias_table",
              "backbone.bottom_up.layers.0.blocks.1.attn.relative_position_index",
              "backbone.bottom_up.layers.1.downsample.reduction.weight", "backbone.bottom_up.norm3.bias",
              "backbone.bottom_up.patch_embed.proj.weight", "backbone.fpn_lateral3.weight", "backbone.fpn_output5.bias",
              "backbone.top_block.p6.weight", "backbone.top_block.p7.bias",
              "proposal_generator.centernet_head.bbox_tower.0.weight", "proposal_generator.centernet_head.bbox_tower.10.bias",
              "proposal_generator.centernet_head.agn_hm.weight", "proposal_generator.centernet_head.scales.4.scale",
              "roi_heads.box_head.2.fc1.weight", "roi_heads.box_predictor.0.cls_score.bias",
              "roi_heads.box_predictor.1.bbox_pred.weight", "roi_heads.box_predictor.2.freq_weight",
              "roi_heads.mask_head.mask_fcn4.weight", "roi_heads.mask_head.deconv.weight", "roi_heads.mask_head.predictor.bias"]:
        assert k in keys, k
    sd = model.state_dict()
    assert sd["roi_heads.box_predictor.0.cls_score.weight"].shape == (1454, 1024)
    assert sd["roi_heads.box_head.0.fc1.weight"].shape == (1024, 256 * 7 * 7)
    assert sd["backbone.bottom_up.layers.0.blocks.0.attn.relative_position_bias_table"].shape == (169, 3)
    # the HIP path is the only path: a forward on CPU tensors must fail loudly
    from divergen_amd._lib import DgxError
    from divergen_amd.data import synthetic_batch
    from divergen_amd.utils.events import EventStorage
    with EventStorage(0), pytest.raises((DgxError, RuntimeError)):
        model.train()(synthetic_batch(1, 64, 1453))


def test_state_dict_matches_the_reference_manifest():
    """tests/golden/swinL_state_manifest.json = key -> shape of the state dict of the REFERENCE's own module classes at the
    configuration of configs/DiverGen_swinL.yaml (tests/golden/make_manifest.py).  The registry-built model must expose exactly
    these keys and shapes -- the released Swin-L checkpoint (DiverGen/README.md:57) is such a state dict -- and must take one
    with strict loading (so `train_net.py --eval-only MODEL.WEIGHTS <checkpoint>`, the run that produces the AP number, cannot
    fail on loading)."""
    import json
    from divergen_amd.modeling import build_model
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "swinL_state_manifest.json")))["entries"]
    model = build_model(_cfg())
    sd = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert sorted(sd) == sorted(man), (sorted(set(sd) - set(man))[:5], sorted(set(man) - set(sd))[:5])
    bad = [(k, sd[k], man[k]) for k in sd if sd[k] != man[k]]
    assert not bad, bad[:5]
    ref_like = {k: torch.empty(v, dtype=model.state_dict()[k].dtype) for k, v in man.items()}
    res = model.load_state_dict(ref_like, strict=True)
    assert not res.missing_keys and not res.unexpected_keys


def test_lr_schedule_and_arena_buckets(golden):
    from divergen_amd.solver import warmup_cosine_lr
    g = golden("solver")
    for it in range(12):
        assert abs(warmup_cosine_lr(1e-2, it, 100, 10, 1e-4) - float(g["lrs"][it])) < 1e-12
    from divergen_amd.engine.ddp import ArenaReducer
    from divergen_amd.solver import FlatArena
    net = torch.nn.Sequential(torch.nn.Linear(10, 7), torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ar = FlatArena(net)
    assert ar.numel % 4 == 0 and all(o % 4 == 0 for o in ar.offsets)
    assert all(p.data_ptr() == ar.p[o:].data_ptr() for p, o in zip(ar.params, ar.offsets))
    red = ArenaReducer(ar, bucket_bytes=64 * 4)
    # buckets tile the arena back to front, contiguously
    assert red.buckets[0][1] == ar.numel and red.buckets[-1][0] == 0
    for a, b in zip(red.buckets[:-1], red.buckets[1:]):
        assert b[1] == a[0]
    (net(torch.randn(4, 10)) ** 2).sum().backward()
    assert ar.g.abs().sum() > 0 and ar.params[0].grad.data_ptr() == ar.g.data_ptr()


def _ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from divergen_amd.engine.ddp import ArenaReducer
    from divergen_amd.solver import FlatArena
    torch.manual_seed(100 + rank)  # different init per rank -> broadcast must equalise
    net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 4), torch.nn.Linear(4, 2))
    unused = torch.nn.Linear(3, 3)  # never used in forward: the "unused parameter" bucket
    model = torch.nn.ModuleList([net, unused])
    ar = FlatArena(model)
    red = ArenaReducer(ar, bucket_bytes=40 * 4)
    red.broadcast_parameters()
    p0 = ar.p.clone()
    x = torch.full((5, 6), float(rank + 1))
    for _ in range(2):  # two iterations: bucket bookkeeping must reset
        ar.zero_grad()
        net(x).sum().backward()
        scale = red.finish()
    # by value (numpy), not as shared-memory tensor handles: the handles die with this process and the parent may
    # not have mapped them yet
    q.put((rank, p0.numpy().copy(), (ar.g.clone() * scale).numpy().copy()))
    dist.destroy_process_group()


class _DirectLinearFn(torch.autograd.Function):
    """The protocol of layers/linear_ops on the CPU: the weight gradient is ADDED straight into the arena view and the
    reducer is signalled once per use; autograd never sees the weight."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x)
        ctx.w = w
        return x @ w.detach().t()

    @staticmethod
    def backward(ctx, g):
        from divergen_amd.layers.linear_ops import notify_ready
        (x,) = ctx.saved_tensors
        ctx.w.grad.add_(g.t() @ x)
        notify_ready(ctx.w)
        return g @ ctx.w.detach(), None          # None: the gradient is already in the arena (no AccumulateGrad)


def _direct_linear(x, w):
    return _DirectLinearFn.apply(x, w)


def _ddp_shared_worker(rank, world, port, q):
    """A weight used TWICE per step through the layers that write gradients straight into the arena (one 'ready' signal
    per use): the bucket must leave after the last use, not the first."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from divergen_amd.engine.ddp import ArenaReducer
    from divergen_amd.solver import FlatArena
    torch.manual_seed(7)
    shared, head = torch.nn.Linear(8, 8, bias=False), torch.nn.Linear(8, 3)
    model = torch.nn.ModuleList([shared, head])
    ar = FlatArena(model)
    red = ArenaReducer(ar, bucket_bytes=16)          # tiny buckets: every parameter its own bucket
    red.broadcast_parameters()
    x = torch.full((4, 8), 0.25 * (rank + 1))
    out = []
    for it in range(3):                              # step 0 calibrates, steps 1-2 launch from the ready signals
        ar.zero_grad()
        y = head(_direct_linear(torch.relu(_direct_linear(x, shared.weight)), shared.weight))
        y.sum().backward()
        scale = red.finish()
        out.append((ar.g.clone() * scale).numpy().copy())
    q.put((rank, ar.p.numpy().copy(), out, list(red._learned[()])))      # no hipGraph segments on the CPU: one mode combination
    dist.destroy_process_group()


def _ddp_uneven_worker(rank, world, port, q):
    """Rank 1 never uses the LAST-registered layer (first bucket in launch order), rank 0 does: which buckets complete
    early differs between the ranks, the order of the collectives must not."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from divergen_amd.engine.ddp import ArenaReducer
    from divergen_amd.solver import FlatArena
    torch.manual_seed(3)
    a, b, c = torch.nn.Linear(5, 5), torch.nn.Linear(5, 5), torch.nn.Linear(5, 2)
    ar = FlatArena(torch.nn.ModuleList([a, b, c]))
    red = ArenaReducer(ar, bucket_bytes=16)
    red.broadcast_parameters()
    x = torch.full((3, 5), 0.5 + rank)
    out = []
    for it in range(3):
        ar.zero_grad()
        h = b(a(x))
        y = c(h).sum() if rank == 0 else h.sum()      # rank 1: `c` unused -> its buckets only leave at finish()
        y.backward()
        scale = red.finish()
        out.append((ar.g.clone() * scale).numpy().copy())
    q.put((rank, ar.p.numpy().copy(), out))
    dist.destroy_process_group()


def test_arena_reducer_rank_dependent_unused_branch_world2_gloo():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, p_a, g_a), (_, _, g_b) = res
    from divergen_amd.solver import FlatArena
    torch.manual_seed(3)
    a, b, c = torch.nn.Linear(5, 5), torch.nn.Linear(5, 5), torch.nn.Linear(5, 2)
    ar = FlatArena(torch.nn.ModuleList([a, b, c]))
    ar.p.copy_(torch.from_numpy(p_a))
    tot = torch.zeros_like(ar.g)
    for r in range(2):
        ar.zero_grad()
        h = b(a(torch.full((3, 5), 0.5 + r)))
        (c(h).sum() if r == 0 else h.sum()).backward()
        tot += ar.g
    for it in range(3):
        assert np.allclose(g_a[it], g_b[it], atol=1e-6), it
        assert np.allclose(g_a[it], (tot / 2).numpy(), atol=1e-5), it


def _ddp_bf16_worker(rank, world, port, q):
    """Three optimizer-free steps with the gradient buckets on the wire as bf16 and as fp32, same weights, same inputs."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from divergen_amd.engine.ddp import ArenaReducer
    from divergen_amd.solver import FlatArena
    out = {}
    for wire in ("fp32", "bf16"):
        torch.manual_seed(5)
        net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8))
        ar = FlatArena(net)
        red = ArenaReducer(ar, bucket_bytes=1024, wire_dtype=wire)
        red.broadcast_parameters()
        g = torch.Generator().manual_seed(100 + rank)
        res = []
        for it in range(3):
            ar.zero_grad()
            net(torch.randn(6, 16, generator=g)).square().sum().backward()
            scale = red.finish()
            res.append((ar.g.clone() * scale).numpy().copy())
        out[wire] = res
    q.put((rank, out))
    dist.destroy_process_group()


def test_arena_reducer_bf16_wire_world2_gloo():
    """SOLVER.ALLREDUCE_DTYPE 'bf16' (SURVEY 5.8): the averaged gradients equal the fp32 all-reduce's within bf16
    rounding of the per-rank terms (2 ranks: each term rounded once, the sum once: <= 3 x 2^-9 relative to the largest term), are
    identical on both ranks, and the alignment gaps of the arena stay zero."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, a), (_, b) = res
    for it in range(3):
        assert np.array_equal(a["bf16"][it], b["bf16"][it]) and np.array_equal(a["fp32"][it], b["fp32"][it])
        f, h = a["fp32"][it], a["bf16"][it]
        assert not np.array_equal(f, h)                                   # it really went over the wire in bf16
        assert np.abs(f - h).max() <= 3 * 2.0 ** -8 * np.abs(f).max()
        assert np.linalg.norm(f - h) <= 2.0 ** -7 * np.linalg.norm(f)
        assert np.array_equal(f == 0, h == 0) or np.count_nonzero(h[f == 0]) == 0


def test_arena_reducer_shared_weight_world2_gloo():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_shared_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, p_a, g_a, exp_a), (_, p_b, g_b, _) = res
    # the shared weight signals once per use (+ once more if autograd runs its AccumulateGrad node on the undefined
    # gradient), the head's two parameters once: the reducer has learned that from step 0
    assert exp_a[0] >= 2 and exp_a[1] == 1 and exp_a[2] == 1
    from divergen_amd.solver import FlatArena
    torch.manual_seed(7)
    shared, head = torch.nn.Linear(8, 8, bias=False), torch.nn.Linear(8, 3)
    ar = FlatArena(torch.nn.ModuleList([shared, head]))
    ar.p.copy_(torch.from_numpy(p_a))
    tot = torch.zeros_like(ar.g)
    for r in range(2):          # plain autograd reference with the broadcast weights
        ar.zero_grad()
        head(shared(torch.relu(shared(torch.full((4, 8), 0.25 * (r + 1)))))).sum().backward()
        tot += ar.g
    for it in range(3):
        assert np.allclose(g_a[it], g_b[it], atol=1e-6), it
        assert np.allclose(g_a[it], (tot / 2).numpy(), atol=1e-5), it


def test_arena_reducer_world2_gloo():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, p_a, g_a), (_, p_b, g_b) = [(r, torch.from_numpy(a), torch.from_numpy(b)) for r, a, b in res]
    assert torch.equal(p_a, p_b)            # parameters broadcast from rank 0
    assert torch.allclose(g_a, g_b)         # averaged gradients identical on both ranks
    # reference value: mean of the two single-rank gradients, computed locally with the same weights
    from divergen_amd.solver import FlatArena
    net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 4), torch.nn.Linear(4, 2))
    model = torch.nn.ModuleList([net, torch.nn.Linear(3, 3)])
    ar = FlatArena(model)
    ar.p.copy_(p_a)
    tot = torch.zeros_like(ar.g)
    for r in range(2):
        ar.zero_grad()
        net(torch.full((5, 6), float(r + 1))).sum().backward()
        tot += ar.g
    assert torch.allclose(g_a, tot / 2, atol=1e-6)
    assert (g_a[-12:] == 0).all()           # the unused layer's bucket was flushed zero-filled


def _ddp_skip_worker(rank, world, port, q):
    """One rank's loss goes non-finite in step 1: the skip flag is MAX-reduced, so BOTH ranks leave their weights alone in that
    step (the optimizer kernel's `found_inf` contract, emulated on the CPU here) and the weights stay identical afterwards."""
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from divergen_amd.engine import ArenaReducer
    from divergen_amd.solver import FlatArena
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 2))
    ar = FlatArena(net)
    red = ArenaReducer(ar, bucket_bytes=64)
    red.broadcast_parameters()
    hist, flags = [], []
    for it in range(3):
        ar.zero_grad()
        loss = net(torch.full((4, 5), 0.25 * (rank + 1))).sum()
        if it == 1 and rank == 0:
            loss = loss * float("inf")               # this rank's loss overflows
        bad = red.agree_on_skip((~torch.isfinite(loss.detach())).to(torch.int32).reshape(1))
        loss.backward()
        scale = red.finish()
        if not int(bad):                             # dgx_adamw_ema_step(found_inf): the whole update is skipped when the flag is set
            with torch.no_grad():
                ar.p.add_(torch.nan_to_num(ar.g) * scale, alpha=-0.1)
        flags.append(int(bad))
        hist.append(ar.p.numpy().copy())
    q.put((rank, flags, hist))
    dist.destroy_process_group()


def test_non_finite_loss_on_one_rank_skips_the_step_on_every_rank_world2_gloo():
    import socket
    import numpy as np
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_skip_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, f0, h0), (_, f1, h1) = res
    assert f0 == [0, 1, 0] and f1 == [0, 1, 0]                       # rank 1 learned of rank 0's overflow
    for a, b in zip(h0, h1):
        assert np.array_equal(a, b)                                  # identical weights after every step
    assert np.array_equal(h0[0], h0[1]) and not np.array_equal(h0[1], h0[2])      # step 1 skipped, step 2 applied


def _ddp_trial_worker(rank, world, port, q):
    """BSGAL's selection runs extra backward passes INSIDE a training step (trial passes, rank-local decisions): under
    linear_ops.suspend_ready() they must neither count as gradient-ready signals nor launch a collective; the step's real
    backward afterwards reduces exactly as without them."""
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from divergen_amd.engine import ArenaReducer
    from divergen_amd.layers.linear_ops import suspend_ready
    from divergen_amd.solver import FlatArena
    torch.manual_seed(11)
    net = torch.nn.Sequential(torch.nn.Linear(6, 6), torch.nn.ReLU(), torch.nn.Linear(6, 2))
    ar = FlatArena(net)
    red = ArenaReducer(ar, bucket_bytes=64)
    red.broadcast_parameters()
    x = torch.full((4, 6), 0.5 + rank)
    out = []
    for it in range(3):
        ar.zero_grad()
        if it >= 1 and rank == 0:                 # only ONE rank runs trial passes: a collective launched from them would hang
            with suspend_ready():
                for _ in range(2):
                    net(x * 3.0).sum().backward()
            ar.zero_grad()
        net(x).sum().backward()
        scale = red.finish()
        out.append((ar.g.clone() * scale).numpy().copy())
    q.put((rank, ar.p.numpy().copy(), out))
    dist.destroy_process_group()


def test_arena_reducer_ignores_trial_backward_passes_world2_gloo():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_trial_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, p_a, g_a), (_, _, g_b) = res
    from divergen_amd.solver import FlatArena
    torch.manual_seed(11)
    net = torch.nn.Sequential(torch.nn.Linear(6, 6), torch.nn.ReLU(), torch.nn.Linear(6, 2))
    ar = FlatArena(net)
    ar.p.copy_(torch.from_numpy(p_a))
    tot = torch.zeros_like(ar.g)
    for r in range(2):
        ar.zero_grad()
        net(torch.full((4, 6), 0.5 + r)).sum().backward()
        tot += ar.g
    for it in range(3):
        assert np.allclose(g_a[it], g_b[it], atol=1e-6), it
        assert np.allclose(g_a[it], (tot / 2).numpy(), atol=1e-5), it


def test_pending_weight_gradients_never_leak_into_the_next_step():
    """layers/swin_block.py defers a block's weight gradients to pair them with the next block's; if a backward pass dies in
    between, clearing the gradients for the next step must drop the stale block (with a warning) instead of pairing it."""
    import warnings
    from divergen_amd.layers import swin_block as SB
    from divergen_amd.solver import FlatArena
    SB._PENDING.append(([], ()))
    arena = FlatArena(torch.nn.Linear(4, 4))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        arena.zero_grad()
    assert not SB._PENDING and not SB._PENDING_LN and any("pending weight gradients dropped" in str(x.message) for x in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        arena.zero_grad()
    assert not w


def test_bench_launches_its_own_ranks_gloo():
    """`python bench.py --gpus 2` without a launcher around it must start one process per rank itself (re-exec under
    torch.distributed.run, as DG/train_net.py:357-362 does through detectron2's launch()) and reach the collective self-check:
    run here over gloo without a GPU (--launch-check: rendezvous + all-reduce probe + the JSON line, no model)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--launch-check", "--backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ranks_seen_by_collective"] == 2 and line["n_gpus"] == 2 and line["launched_by"] == "torchrun"


def test_build_optimizer_branches_follow_build_custom_optimizer():
    """custom_solver.py:19-77: OPTIMIZER 'ADAMW' | 'SGD' (momentum / nesterov from the config, lr multipliers for backbone and
    CUSTOM_MULTIPLIER_NAME, one weight decay), CLIP_TYPE 'value' or 'full_model'; anything else raises like :74-75."""
    import pytest
    import torch
    from divergen_amd.config import get_cfg
    from divergen_amd.solver import FusedAdamWEMA, FusedSGDEMA, build_optimizer

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Linear(4, 4)
            self.head = torch.nn.Linear(4, 2)
    cfg = get_cfg()
    cfg.SOLVER.USE_CUSTOM_SOLVER = True
    cfg.SOLVER.OPTIMIZER = "ADAMW"
    assert type(build_optimizer(cfg, Net())) is FusedAdamWEMA
    cfg.SOLVER.OPTIMIZER = "SGD"
    cfg.SOLVER.MOMENTUM, cfg.SOLVER.NESTEROV, cfg.SOLVER.BACKBONE_MULTIPLIER = 0.8, True, 0.1
    cfg.SOLVER.CLIP_GRADIENTS.ENABLED, cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE, cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE = True, "full_model", 0.5
    opt = build_optimizer(cfg, Net())
    assert type(opt) is FusedSGDEMA and (opt.momentum, opt.nesterov, opt.clip_norm, opt.clip_value) == (0.8, True, 0.5, 0.0)
    assert opt.buf is not None and opt.buf.shape == opt.arena.p.shape
    by_name = dict(zip(opt.arena.names, opt.lr_scale.tolist()))
    assert by_name["backbone.weight"] == pytest.approx(0.1) and by_name["head.weight"] == 1.0
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE = "value"
    opt = build_optimizer(cfg, Net())
    assert (opt.clip_norm, opt.clip_value) == (0.0, 0.5)
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE = "norm"
    with pytest.raises(NotImplementedError):
        build_optimizer(cfg, Net())
    # the reference wraps AdamW in the full-model clipper as well (custom_solver.py:46-60,69-72)
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE, cfg.SOLVER.OPTIMIZER = "full_model", "ADAMW"
    opt = build_optimizer(cfg, Net())
    assert type(opt) is FusedAdamWEMA and (opt.clip_norm, opt.clip_value) == (0.5, 0.0)
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE, cfg.SOLVER.OPTIMIZER = "value", "LAMB"
    with pytest.raises(NotImplementedError):
        build_optimizer(cfg, Net())


def test_optimizer_state_is_remapped_by_name_or_refused():
    """--resume: flat moments written under another arena layout are mapped per parameter (same storage order) or refused with a
    clear message (another storage-order version / no sizes) -- never copied raw onto the wrong parameters."""
    from divergen_amd.solver import ARENA_LAYOUT_VERSION, FlatArena, FusedAdamWEMA

    def net(order):
        m = torch.nn.Module()
        for nm in order:
            setattr(m, nm, torch.nn.Linear(8, 8, bias=False))
        return m
    a = FusedAdamWEMA(FlatArena(net(["one", "two"])), 1e-3)
    a.m.copy_(torch.arange(a.m.numel(), dtype=torch.float32))
    a.v.copy_(torch.arange(a.v.numel(), dtype=torch.float32) * 2)
    a.step_count = 7
    sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in a.state_dict().items()}
    assert sd["layout_version"] == ARENA_LAYOUT_VERSION and sd["sizes"] == list(a.arena.sizes)
    b = FusedAdamWEMA(FlatArena(net(["two", "one"])), 1e-3)          # same parameters, other registration order
    b.load_state_dict(sd)
    assert b.step_count == 7
    for nm in ("one.weight", "two.weight"):
        ia, ib = a.arena.names.index(nm), b.arena.names.index(nm)
        sa = slice(a.arena.offsets[ia], a.arena.offsets[ia] + a.arena.sizes[ia])
        sb = slice(b.arena.offsets[ib], b.arena.offsets[ib] + b.arena.sizes[ib])
        assert torch.equal(a.m[sa], b.m[sb]) and torch.equal(a.v[sa], b.v[sb])
    c = FusedAdamWEMA(FlatArena(net(["one", "two"])), 1e-3)          # identical layout: plain copy
    c.load_state_dict(sd)
    assert torch.equal(c.m, a.m)
    old = dict(sd)
    old["layout_version"] = ARENA_LAYOUT_VERSION - 1
    old["offsets"] = [o + 4 for o in sd["offsets"]]
    with pytest.raises(RuntimeError, match="storage layout"):
        b.load_state_dict(old)
    legacy = {k: v for k, v in sd.items() if k not in ("sizes", "layout_version")}
    legacy["offsets"] = [o + 4 for o in sd["offsets"]]
    with pytest.raises(RuntimeError, match="storage layout"):
        b.load_state_dict(legacy)


def test_lazy_zero_grad_mixed_launch_does_not_accumulate_onto_stale_gradients():
    """A grouped weight-gradient launch that has to ACCUMULATE (one of its members was already written in this pass) may contain
    members the lazy zero_grad left un-zeroed: claim_first_write zeroes those before the launch adds into them.  The zero table of
    the lazy path is built on 4-element boundaries."""
    from divergen_amd.solver import FlatArena
    m = torch.nn.Module()
    m.a, m.b = torch.nn.Linear(8, 8, bias=False), torch.nn.Linear(8, 6, bias=False)
    ar = FlatArena(m)
    ar._direct_state()
    pa, pb = m.a.weight, m.b.weight
    ia, ib = pa._dgx_arena_slot[1], pb._dgx_arena_slot[1]
    seg = lambda i: ar.g[ar.offsets[i]:ar.offsets[i] + ar.sizes[i]]
    # step 1: both written first -> direct; the "launch" overwrites
    ar.zero_grad()
    assert ar.claim_first_write([pa, pb])
    seg(ia).fill_(1.0), seg(ib).fill_(2.0)
    # step 2 (what zero_grad(lazy=True) does on the device is emulated: the kept segments are left alone)
    ar.gen += 1
    ar.direct = {ia, ib}
    ar._lazy_pending = {ia, ib}
    assert ar.claim_first_write([pa])                    # first forward's backward writes a
    seg(ia).fill_(5.0)
    assert not ar.claim_first_write([pa, pb])            # mixed launch: a already written, b still holds last step's 2.0
    assert float(seg(ib).abs().max()) == 0.0 and float(seg(ia).max()) == 5.0
    assert ib not in ar._lazy_pending and ib not in ar.direct


def test_box_head_first_fc_is_stored_hwc_and_state_dicts_keep_the_reference_order():
    """FastRCNNConvFCHead keeps fc1's columns in (h, w, c) order (the pooled features are channels-last in memory: flattening is a
    view), while state_dict() / load_state_dict() speak the reference's (c, h, w) order (box_head.py:26-98: nn.Flatten of
    (R, C, S, S)): weights copied from a reference-layout module give the same outputs, and a round trip is the identity."""
    import torch
    from divergen_amd.modeling import ShapeSpec
    from divergen_amd.modeling.roi_heads.box_head import FastRCNNConvFCHead
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(8 * 3 * 3, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU())
    sd = {"fc1.weight": ref[1].weight.detach().clone(), "fc1.bias": ref[1].bias.detach().clone(),
          "fc2.weight": ref[3].weight.detach().clone(), "fc2.bias": ref[3].bias.detach().clone()}
    h = FastRCNNConvFCHead(ShapeSpec(channels=8, height=3, width=3), conv_dims=[], fc_dims=[16, 16])
    h.load_state_dict(sd)
    out = h.state_dict()
    for k in sd:
        assert torch.equal(out[k], sd[k]), k
    w = h.fcs[0].weight.detach()
    assert not torch.equal(w, sd["fc1.weight"])                      # stored permuted ...
    assert torch.equal(w.view(16, 3, 3, 8).permute(0, 3, 1, 2).reshape(16, -1), sd["fc1.weight"])       # ... as (h, w, c)
    x = torch.randn(5, 8, 3, 3)
    rows = h.flatten_rows(x.contiguous(memory_format=torch.channels_last))
    assert rows.data_ptr() == x.contiguous(memory_format=torch.channels_last).data_ptr() or rows.is_contiguous()
    got = torch.relu(torch.relu(rows @ w.t() + sd["fc1.bias"]) @ sd["fc2.weight"].t() + sd["fc2.bias"])
    torch.testing.assert_close(got, ref(x), atol=1e-5, rtol=1e-5)
