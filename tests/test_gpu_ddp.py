"""Multi-GPU readiness that a one-GPU box can prove (DG/train_net.py:357-362 wraps the model in DistributedDataParallel; here
engine/ddp.ArenaReducer): a ONE-rank RCCL group runs three real training steps of the benchmarked model (Swin-L CenterNet2, 1024^2, 2
images) with everything the N > 1 path switches on --

  * the reducer's hooks, bucket launches during backward and RCCL collectives next to the hipGraph-replayed segments,
  * dgx_set_reserved_cus(16): the persistent GEMM / weight-gradient kernels on 30 instead of 32 workgroups per XCD,
  * the deferred weight-gradient queue (7 stage-2 blocks = 28 problems per loader-wave launch),

and must produce the gradients of the run WITHOUT a reducer bit for bit in every step (a sum over one rank is the identity; each step
starts from the other run's weights; the relative-position tables, whose gradient is scattered with fp32 atomics, to 1e-5), and every bucket must be launched AFTER the last weight-gradient launch
that writes into its slice of the gradient arena -- checked on the launches themselves (host order and HIP events), not by weight
equality."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(swin, size):
    from divergen_amd.config import get_cfg
    from divergen_amd.modeling import build_model
    from divergen_amd.solver import build_optimizer
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "DiverGen_swinL.yaml"))
    cfg.merge_from_list(["MODEL.SWIN.SIZE", swin, "INPUT.TRAIN_SIZE", size, "MODEL.ROI_BOX_HEAD.CAT_FREQ_PATH",
                         os.path.join(ROOT, "configs", "metadata", "ImageNet2012_filtered04_lvis_v1_train_cat_info_250.json")])
    torch.manual_seed(cfg.SEED)
    model = build_model(cfg).train()
    return cfg, model, build_optimizer(cfg, model)


def _run(with_reducer, steps=4, swin="L-22k-384", size=1024, weights_in=None, early=True):
    import torch.distributed as dist
    from divergen_amd import _lib as L
    from divergen_amd.data import synthetic_batch
    from divergen_amd.engine import ArenaReducer, total_loss
    from divergen_amd.engine import ddp as DDP
    from divergen_amd.layers import swin_block as SB
    from divergen_amd.utils.events import EventStorage
    cfg, model, opt = _build(swin, size)
    model.early_proposal_backward = early          # as bench.py / train_net.py run it: CenterNet's gradients are written DURING the forward
    reducer = ArenaReducer(opt.arena, single_rank_group=with_reducer)
    assert reducer.active == with_reducer
    if with_reducer:
        from divergen_amd.utils import graphs
        reducer.reserved_cus = 16                 # what ArenaReducer takes from NCCL_MAX_NCHANNELS when the group has more than one rank:
        graphs.CAPTURE_RESERVED_CUS = 16          # 16 CUs left alone from begin_backward() to finish(), graphs captured at that width
        reducer.broadcast_parameters()
    log, seq = [], [0]
    orig_w, orig_b = SB.wgrad_grouped, reducer._launch
    g0, g_end = opt.arena.g.data_ptr(), opt.arena.g.data_ptr() + opt.arena.g.numel() * 4

    def wgrad_spy(problems, beta=1.0):
        orig_w(problems, beta)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        seq[0] += 1
        offs = [(p[0].data_ptr() - g0) // 4 for p in problems if g0 <= p[0].data_ptr() < g_end]
        offs += [(p[3].data_ptr() - g0) // 4 for p in problems if len(p) > 3 and p[3] is not None and g0 <= p[3].data_ptr() < g_end]
        log.append(("w", seq[0], ev, offs, len(problems)))

    def bucket_spy(b):
        if not reducer._launched[b]:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            seq[0] += 1
            log.append(("b", seq[0], ev, reducer.buckets[b][:2], b))
        orig_b(b)
    SB.wgrad_grouped = wgrad_spy
    reducer._launch = bucket_spy
    batch = synthetic_batch(2, size, cfg.MODEL.ROI_HEADS.NUM_CLASSES, device="cuda")
    grads, per_step, weights = [], [], []
    try:
        with EventStorage(0):
            for it in range(steps):
                torch.manual_seed(1000 + it)
                if weights_in is not None:        # every step starts from the reference run's weights: the tables' atomically scattered
                    opt.arena.p.copy_(weights_in[it])      # gradients differ in the last bits between any two runs, and one optimizer step
                    opt.arena.sync_shadow()                # later so does every activation
                weights.append(opt.arena.p.clone())
                del log[:]
                opt.zero_grad()
                losses = model(batch)
                reducer.begin_backward()
                assert L.reserved_cus() == (16 if with_reducer else 0)
                total_loss(losses).backward()
                scale = reducer.finish()
                assert L.reserved_cus() == 0
                torch.cuda.synchronize()
                grads.append(opt.arena.g.clone())
                per_step.append((list(log), reducer.last_early if with_reducer else 0))
                opt.step(grad_scale=scale)
    finally:
        SB.wgrad_grouped = orig_w
        L.set_reserved_cus(0)
        if with_reducer:
            from divergen_amd.utils import graphs
            graphs.CAPTURE_RESERVED_CUS = 0
    return opt, grads, per_step, reducer, weights


@pytest.mark.parametrize("early,block_graphs", [(True, False), (False, False), (True, True)])
def test_one_rank_rccl_group_trains_like_no_reducer(early, block_graphs, monkeypatch):
    """block_graphs False: the Swin blocks issued eagerly (rounds 1-5; what a batch size beyond graphs.MAX_GRAPHS still runs as), so the
    bucket-after-last-write order can be checked on the launches themselves.  True: the block groups replayed as hipGraphs (round 6):
    no Python runs between the weight-gradient launches of a group, its parameters are signalled behind the replay -- gradients bit for
    bit as without a reducer, buckets still leaving during backward."""
    import torch.distributed as dist
    from divergen_amd.modeling.backbone import swintransformer as S
    from divergen_amd.utils import graphs
    assert graphs.ENABLED, "the hipGraph segments are part of what is being proven"
    monkeypatch.setattr(S, "GRAPH_BLOCKS", block_graphs)
    monkeypatch.setattr(S, "GRAPH_BLOCKS_MAX_GFLOP", 1e9)      # (the policy leaves Swin-L's blocks eager at this size: replay them here all the same)
    opt0, grads0, log0, _, weights0 = _run(False, early=early)
    if not block_graphs:
        assert max(n for kind, _, _, _, n in log0[-1][0] if kind == "w") >= 28, "the deferred 28-problem loader-wave group must be active"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        opt1, grads1, log1, reducer, _ = _run(True, weights_in=weights0, early=early)
    finally:
        dist.destroy_process_group()
    # ---- gradients: bit for bit outside the atomically scattered relative-position tables
    table = torch.zeros(opt0.arena.g.numel(), dtype=torch.bool, device="cuda")
    for n, o, z in zip(opt0.arena.names, opt0.arena.offsets, opt0.arena.sizes):
        if "relative_position_bias_table" in n:
            table[o:o + z] = True
    for it, (a, b) in enumerate(zip(grads0, grads1)):
        assert bool(torch.isfinite(b).all())
        diff = (a != b) & ~table
        assert not bool(diff.any()), (it, int(diff.sum()), float((a - b).abs().max()))
        if float((a - b)[table].abs().max()) > 1e-5 * float(a[table].abs().max()) + 1e-9:
            bad = []
            for n, o, z in zip(opt0.arena.names, opt0.arena.offsets, opt0.arena.sizes):
                if "relative_position_bias_table" in n and float((a - b)[o:o + z].abs().max()) > 1e-5 * float(a[table].abs().max()) + 1e-9:
                    bad.append((n, float(a[o:o + z].abs().max()), float(b[o:o + z].abs().max()), float((a - b)[o:o + z].abs().max())))
            raise AssertionError("relative-position table gradients differ at step %d: %s" % (it, bad))
    # ---- every bucket behind the last weight-gradient launch into its slice: steps 3 and 4 (step 1 runs the hipGraph segments eagerly and
    # learns the signal counts of that mode, step 2 captures them and learns the counts of the replayed mode: neither launches early)
    for it in (2, 3):
        log, early = log1[it]
        buckets = [e for e in log if e[0] == "b"]
        writes = [e for e in log if e[0] == "w"]
        if block_graphs:          # the weight-gradient launches are inside the replayed graphs: nothing to spy on; the order is by construction
            assert len(buckets) == len(reducer.buckets) and early >= len(reducer.buckets) - 4, (early, len(reducer.buckets))
            continue
        assert len(buckets) == len(reducer.buckets) and len(writes) >= 4
        assert early >= len(reducer.buckets) - 2, "buckets are expected to leave DURING backward (%d of %d did)" % (early, len(reducer.buckets))
        for _, bseq, bev, (lo, hi), b in buckets:
            for _, wseq, wev, offs, _n in writes:
                if any(lo <= o < hi for o in offs):
                    assert wseq < bseq, "bucket %d was launched before a weight-gradient launch that writes into it" % b
                    assert wev.elapsed_time(bev) >= 0.0, "bucket %d's launch point precedes its last gradient write on the stream" % b
