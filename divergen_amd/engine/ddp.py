"""Data-parallel gradient reduction over the flat gradient arena (one process per GPU, RCCL).

Replaces torch DistributedDataParallel as used at DG/train_net.py:357-362
(broadcast_buffers=False, find_unused_parameters=True): gradients already live contiguously in
FlatArena.g, so a bucket is a SLICE of that arena -- no flatten/unflatten copies, no reducer graph
walk.  Buckets are cut in reverse registration order (mask head -> box heads -> CenterNet head ->
FPN -> Swin stage 3..0), each bucket's all-reduce is launched from the post-accumulate hook of its
last gradient (RCCL runs it on its own stream, ordered after the producing kernels) and overlaps
the rest of backward; buckets whose parameters received no gradient this step (the reference's
'unused parameter' case, e.g. the mask head on a batch without masks) are flushed zero-filled at
the end.  Averaging (1/world) is folded into the optimizer kernel's grad_scale.
xGMI is point-to-point, so a ring all-reduce is bound by one ~153 GB/s link: buckets default to
64 MiB (fewer, larger collectives) rather than DDP's 25 MiB.

"Ready" means the LAST write of a step into a parameter's gradient.  Autograd's AccumulateGrad hook fires once per
parameter, but the layers that write their weight gradient straight into the arena signal once per USE, and a weight
shared by several call sites (the CenterNet tower over five FPN levels) is used several times per step.  The reducer
therefore learns the number of signals per parameter from the first step (during which nothing is launched early) and
afterwards counts a parameter as ready at its last expected signal.  The counts depend on HOW the hipGraph segments ran in a step
(a replayed segment signals each parameter once behind the replay, an eagerly issued one once per use; a segment runs eagerly on
the first sight of a batch size and on sizes beyond utils.graphs.MAX_GRAPHS), so one count vector is learned per combination of
segment modes (layers.linear_ops.SEGMENT_MODES), each in a step of its own without early launches.  Fewer signals than learned (a hipGraph replay
produces none, a branch not taken) only defer the bucket to `finish()`; MORE signals than learned would mean a bucket
could have left before its gradients were complete, and `finish()` raises."""
import os

import torch
import torch.distributed as dist

from ..layers import linear_ops


class ArenaReducer:
    def __init__(self, arena, bucket_bytes=64 << 20, process_group=None, single_rank_group=False, wire_dtype="fp32"):
        """wire_dtype 'bf16' (SOLVER.ALLREDUCE_DTYPE): a bucket is converted to bf16 into a staging arena of the same layout, the
        staging slice is all-reduced, and `finish()` converts the whole staging arena back into the fp32 gradient arena in one pass:
        half the bytes on the xGMI links (0.5 instead of 1.0 GB per rank and step), sums rounded to bf16 at every ring hop -- the
        optimizer's moments and the weights stay fp32.  'fp32' (default) all-reduces the gradient arena in place, bit-compatible
        with the reference's DDP."""
        self.arena, self.group = arena, process_group
        assert wire_dtype in ("fp32", "bf16"), wire_dtype
        self.wire = torch.zeros(arena.g.numel(), dtype=torch.bfloat16, device=arena.g.device) if wire_dtype == "bf16" else None
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        n = len(arena.params)
        ends = arena.segment_ends()
        self.buckets = []           # (start, end) element ranges, in launch (reverse) order
        self.bucket_of = [0] * n
        cap = bucket_bytes // 4
        hi = n - 1
        while hi >= 0:
            lo = hi
            while lo - 1 >= 0 and ends[hi] - arena.offsets[lo - 1] <= cap:
                lo -= 1
            for i in range(lo, hi + 1):
                self.bucket_of[i] = len(self.buckets)
            self.buckets.append((arena.offsets[lo], ends[hi], hi - lo + 1))
            hi = lo - 1
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._ready = [False] * len(self.buckets)
        self._next = 0                 # buckets are launched strictly in index order (same order on every rank)
        self.last_early = 0
        self._got = [0] * n            # ready signals of this step, per parameter
        self._learned = {}             # segment-mode combination -> learned signal counts
        self._expected = None          # this step's counts; None = calibrating (no early launches)
        self._mode_key = None          # taken at the step's first signal (every segment has run its forward by then)
        # single_rank_group: reduce over a one-rank group as well (bench.py --force-pg: RCCL next to hipGraph capture on one GPU)
        self.active = self.world > 1 or (dist.is_initialized() and single_rank_group)
        self.reserved_cus = 0
        if self.active and self.world > 1 and arena.g.is_cuda and dist.get_backend(self.group) == "nccl":
            # RCCL runs one workgroup per channel, each pinned to a CU for the collective's duration, beside the backward kernels.
            # libdgx's persistent kernels (one workgroup per CU, whole LDS + register file) cannot share a CU with a channel, so they
            # are told to leave that many CUs alone (csrc/gemm_lw.hip): NCCL_MAX_NCHANNELS bounds the channel count
            if "NCCL_MAX_NCHANNELS" not in os.environ:
                # torch creates the RCCL communicator at the first collective, so a default set here still bounds the channels when
                # no collective has run yet; a launcher that already ran one keeps RCCL's own (larger) count -- say so
                import warnings
                os.environ["NCCL_MAX_NCHANNELS"] = "16"
                warnings.warn("divergen_amd: NCCL_MAX_NCHANNELS was not set by the launcher; set to 16 now (engine/launch.py, train_net.py "
                              "and bench.py export it before init_process_group) -- if a collective has already run, RCCL keeps its own "
                              "channel count and its channels will contend with the persistent GEMM workgroups")
            self.reserved_cus = int(os.environ["NCCL_MAX_NCHANNELS"])
        # The CUs are left to RCCL only while collectives can be in flight -- from the start of backward (`begin_backward`, or the first
        # bucket of a loop that does not announce it) to `finish()`: forward and optimizer run on all 256.  hipGraph segments fix their
        # grids at capture, and their backward graphs replay beside the collectives: they are captured at the reduced width.
        self._cus_on = False
        if arena.g.is_cuda:      # process-global in libdgx: an inactive / single-rank reducer gives the CUs back
            from ..utils import graphs
            graphs.CAPTURE_RESERVED_CUS = self.reserved_cus
            self._reserve(False)
        if self.active:
            for i, p in enumerate(arena.params):
                hook = self._make_hook(i)
                p.register_post_accumulate_grad_hook(hook)
                # ops that write their gradient straight into the arena (layers/linear_ops.py) bypass
                # AccumulateGrad and call this instead
                p._dgx_ready = (lambda h=hook: h(None))

    def _reserve(self, on):
        if self.arena.g.is_cuda:
            from .. import _lib as L
            L.set_reserved_cus(self.reserved_cus if on else 0)
        self._cus_on = bool(on)

    def begin_backward(self):
        """Call right before `loss.backward()`: from here to `finish()` the persistent kernels leave `reserved_cus` CUs to RCCL."""
        if self.active and self.reserved_cus and not self._cus_on:
            self._reserve(True)

    def _make_hook(self, i):
        b = self.bucket_of[i]

        def hook(_param):
            if linear_ops._READY_SUSPENDED[0]:     # trial backward passes (hipGraph capture warm-ups, BSGAL's selection) are not
                return                             # part of the training step: nothing to count, nothing to reduce
            if self._mode_key is None:
                self._mode_key = tuple(sorted(linear_ops.SEGMENT_MODES.items()))
                self._expected = self._learned.get(self._mode_key)
            self._got[i] += 1
            if self._expected is not None and self._got[i] == self._expected[i]:
                self._pending[b] -= 1
                if self._pending[b] == 0:
                    self._ready[b] = True
                    self._launch_in_order()
        return hook

    def _launch_in_order(self):
        """Collectives must be issued in the SAME order on every rank, and which buckets complete early is data dependent
        (a rank whose batch has no mask targets never completes the mask head's bucket before `finish()`): a bucket is
        only launched once every bucket before it has been (torch DDP's rule)."""
        while self._next < len(self.buckets) and self._ready[self._next]:
            self._launch(self._next)
            self._next += 1

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        if self.reserved_cus and not self._cus_on:      # a loop that did not call begin_backward: from the first collective on
            self._reserve(True)
        s, e, _ = self.buckets[b]
        buf = self.arena.g[s:e]
        if self.wire is not None:
            buf = self.wire[s:e]
            buf.copy_(self.arena.g[s:e])          # fp32 -> bf16 (round to nearest even), on the stream the gradients were written on
        self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def broadcast_parameters(self, src=0):
        """Same initial weights on every rank (DDP's constructor broadcast)."""
        if self.active:
            dist.broadcast(self.arena.p, src=src, group=self.group)
            self.arena.sync_shadow()

    def agree_on_skip(self, flag):
        """flag: int32 tensor, non-zero on a rank whose loss is not finite (DG/train_net.py:266 asserts there; here the flag stays
        on the device and makes the optimizer kernel skip the step: `found_inf`).  With several ranks EVERY rank must skip, or the
        weights diverge: the flag is MAX-reduced over the group, in place.  Returns the flag."""
        if self.active and self.world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        return flag

    def finish(self):
        """Call after backward: flush never-ready buckets, wait for every collective.  Returns the
        factor the optimizer must apply to the summed gradients."""
        self.arena.finish_grads() if hasattr(self.arena, "finish_grads") else None     # directly-written segments nobody wrote: zeroed
        if self.active:
            self.last_early = self._next          # buckets that left while backward was still running (diagnostic)
            for b in range(self._next, len(self.buckets)):
                self._launch(b)
            for w in self._works:
                w.wait()
            if self.wire is not None:
                self.arena.g.copy_(self.wire)     # bf16 -> fp32, every bucket at once (the staging arena's alignment gaps stay zero)
        self._works = []
        if self._cus_on:
            self._reserve(False)
        if self.active:
            if self._mode_key is None:
                self._mode_key = tuple(sorted(linear_ops.SEGMENT_MODES.items()))
                self._expected = self._learned.get(self._mode_key)
            if self._expected is None:
                self._learned[self._mode_key] = list(self._got)
            else:
                late = [self.arena.names[i] for i, (g, e) in enumerate(zip(self._got, self._expected)) if e > 0 and g > e]
                if late:
                    raise RuntimeError("ArenaReducer: parameters signalled 'gradient ready' more often than in the first step "
                                       "(%s ...): their bucket may have been reduced before the last write" % ", ".join(late[:4]))
        self._got = [0] * len(self._got)
        self._mode_key = self._expected = None
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        return 1.0 / self.world
