"""hipGraph capture of static-shape model segments (forward AND backward).

The dense heads (FPN top-down path, CenterNet tower) are a few hundred tiny kernels whose shapes depend only
on the padded image size; eagerly the host needs longer to issue them than the GPU needs to run them.  A
segment is captured once per input signature with torch.cuda.make_graphed_callables and replayed afterwards:
two graph launches (forward, backward) instead of ~10^3 kernel launches.  The reference has no counterpart (it
runs these modules eagerly: D2/modeling/backbone/fpn.py:113-154, CN/modeling/dense_heads/centernet_head.py:141-162).

Segments must be free of host syncs, data-dependent shapes and Python-side effects.  Parameter gradients are
written in place into the gradient arena by the captured kernels.  The data-parallel reducer cannot be signalled from
inside a replay (its per-parameter callbacks are Python), so the segment's inputs pass through an identity autograd node
whose backward runs right after the replayed backward graph has been enqueued and signals every parameter of the segment
(without it the reducer, which launches buckets in index order, would hold the whole backbone behind these buckets)."""
import os

import torch

from ..layers import linear_ops
from . import prof

ENABLED = True
# Which input signatures get a graph.  A training run on real data sees MANY padded batch sizes (EfficientDetResizeCrop with
# SCALE_RANGE (0.1, 2.0): every multiple of the size divisibility up to TRAIN_SIZE in both dimensions; about half of the batches
# have the full TRAIN_SIZE x TRAIN_SIZE), and a captured pair of graphs pins its activations (~1-2 GB at 1024^2).  A signature is
# captured when it is seen for the CAPTURE_AFTER-th time and while fewer than MAX_GRAPHS signatures hold graphs; every other
# call runs the segment eagerly (the callers fall back to the module's own forward when usable() says no).
CAPTURE_AFTER = 2
MAX_GRAPHS = 4
CAPTURE_RESERVED_CUS = 0     # engine/ddp.ArenaReducer: the width (256 - this many CUs) the captured persistent kernels are launched at
ALIAS_STATIC = True     # chained segments share their hand-over buffers


class _SignalAfterBackward(torch.autograd.Function):
    """Identity on the first input of a graphed segment; its backward = 'the segment's parameter gradients are written'."""

    @staticmethod
    def forward(ctx, x, params):
        ctx.params = params
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        for p in ctx.params:
            linear_ops.notify_ready(p)
        return g, None


class _thread_local_capture:
    """torch.cuda.make_graphed_callables captures with the stream-capture mode 'global': a prohibited call by ANY thread of the
    process invalidates the capture.  In the training process the DataLoader's pin thread allocates pinned host memory
    (hipHostMalloc) whenever a worker result arrives -- observed: `hipErrorStreamCaptureInvalidated` on the first step fed by the
    real loader.  The captures here run entirely on the calling thread, so the mode that matches them is 'thread_local'."""

    def __enter__(self):
        self.orig = torch.cuda.graph
        orig = self.orig

        class graph(orig):
            def __init__(self, cuda_graph, pool=None, stream=None, capture_error_mode="thread_local"):
                super().__init__(cuda_graph, pool=pool, stream=stream, capture_error_mode="thread_local")
        torch.cuda.graph = graph

    def __exit__(self, *exc):
        torch.cuda.graph = self.orig


class GraphedSegment:
    def __init__(self, module):
        self.module = module           # nn.Module: forward(*tensors) -> tuple of tensors
        self._fns = {}
        self._work = {}
        self._seen = {}

    @staticmethod
    def _key(inputs, tag=None):
        return tuple((tuple(t.shape), t.dtype, t.requires_grad, tuple(t.stride())) for t in inputs) + (torch.is_autocast_enabled(), tag)

    def usable(self, inputs, tag=None):
        """True when this call replays (or now captures) a graph; False = run the segment eagerly.  `tag`: whatever else the
        segment's launch shapes depend on (the (H, W) of a token grid that arrives flattened)."""
        if not (ENABLED and torch.is_grad_enabled() and all(t.is_cuda for t in inputs)
                and not torch.cuda.is_current_stream_capturing()):
            linear_ops.SEGMENT_MODES[id(self)] = "e"
            return False
        key = self._key(inputs, tag)
        ok = key in self._fns
        if not ok:
            n = self._seen[key] = self._seen.get(key, 0) + 1
            ok = n >= CAPTURE_AFTER and len(self._fns) < MAX_GRAPHS
        linear_ops.SEGMENT_MODES[id(self)] = "g" if ok else "e"
        return ok

    def __call__(self, *inputs, tag=None):
        key = self._key(inputs, tag)
        fn = self._fns.get(key)
        if fn is None:
            self.module.amp = torch.is_autocast_enabled()
            # an input that IS another segment's static output (the FPN's levels handed to the tower) becomes this graph's static
            # input as it stands: same address on every replay, so the replay wrapper's address check skips its input copy
            sample = tuple((t.detach() if ALIAS_STATIC and getattr(t, "_dgx_static_output", False) else t.detach().clone())
                           .requires_grad_(t.requires_grad) for t in inputs)
            # capture outside the caller's autocast region (its weight-cast cache cannot be captured); the segment
            # module re-enters autocast itself with the cache off
            before = prof.captured_snapshot() if prof.ON else None
            # make_graphed_callables warms up on one side stream and captures on another: the sample inputs' AccumulateGrad nodes are
            # born on the first and meet gradients produced on the second, which autograd reports ("AccumulateGrad node's stream does
            # not match ...").  That is the capture itself; afterwards NO parameter gradient of the model passes through an
            # AccumulateGrad node (all are written in place into the arena: tools/accum_grad_params.py prints 0 of 408 per step), so
            # the report is silenced for the capture only and stays armed for the training steps.
            warn = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if warn is not None:
                warn(False)
            from .. import _lib as L
            width = L.reserved_cus()
            if CAPTURE_RESERVED_CUS != width:
                L.set_reserved_cus(CAPTURE_RESERVED_CUS)
            try:
                # Given an nn.Module, make_graphed_callables returns the module with its `forward` REPLACED (an instance attribute)
                # by the graphed one.  A second signature must not run its warm-up through the first one's graph (rounds 1-5 never saw
                # a second signature: synthetic batches have one size; the real loader's second batch size died here with a shape
                # mismatch in the replay's input copy): the graphed forward is taken OFF the module and kept per signature.
                with linear_ops.suspend_ready(), torch.autocast("cuda", enabled=False), _thread_local_capture():
                    torch.cuda.make_graphed_callables(self.module, sample, allow_unused_input=True)
                fn = self.module.__dict__.pop("forward")
            finally:
                if CAPTURE_RESERVED_CUS != width:
                    L.set_reserved_cus(width)
                if warn is not None:
                    warn(True)
            if before is not None:      # heavy launches recorded into the two graphs (forward + backward): credited per replay
                after = prof.captured_snapshot()
                self._work[key] = {k: tuple(x - y for x, y in zip(after[k], before[k])) for k in after}
            # the capture warm-up ran real backward passes whose in-place gradient writes landed in the arena;
            # this step's backward has not started yet, so clearing them is exact
            for p in self.module.parameters():
                if p.grad is not None:
                    p.grad.zero_()
            self._fns[key] = fn
        k = next((i for i, t in enumerate(inputs) if t.requires_grad), None)
        if k is not None and torch.is_grad_enabled():
            params = self.__dict__.get("_params")
            if params is None:
                params = self.__dict__["_params"] = tuple(p for p in self.module.parameters() if p.requires_grad)
            inputs = inputs[:k] + (_SignalAfterBackward.apply(inputs[k], params),) + inputs[k + 1:]
        if prof.ON and key in self._work:
            prof.add_replay(self._work[key])
        outs = fn(*inputs)
        for o in outs:
            if isinstance(o, torch.Tensor):
                o._dgx_static_output = True
        return outs
