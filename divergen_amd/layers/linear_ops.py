"""Linear layers over the flat parameter arenas.

A parameter that lives in a FlatArena carries `_dgx16` (its bf16 shadow, refreshed by the optimizer
kernel) and a persistent fp32 `.grad` view.  The GEMMs then run bf16 x bf16 -> bf16 with no per-step
weight casts, and the weight gradient is ONE library GEMM that accumulates in fp32 straight into the
gradient arena (beta = 1) -- no bf16 gradient tensor, no cast, no AccumulateGrad add.  That removes
~6 tiny launches per Linear per step (~1000 launches for Swin-L CenterNet2).
Reference: nn.Linear call sites in swintransformer.py (qkv/proj/fc1/fc2/reduction), box_head.py, fast_rcnn.py."""
import torch
import torch.nn as nn

from .. import _lib as L

BF16 = torch.bfloat16


def pad8(n):
    return (n + 7) // 8 * 8


def wgrad_into(g2, dy2, x2, beta=1.0, gb=None):
    """g2 fp32 (Nn,Kk) = beta*g2 + dy2^T x2 on the 256x256 split-M MFMA kernel (dgx_linear_wgrad_grouped, a group of
    one).  There is no library route: shapes the kernel does not take raise."""
    M, Nn = dy2.shape
    Kk = x2.shape[1]
    if not (dy2.is_cuda and Nn % 8 == 0 and Kk % 8 == 0 and g2.is_contiguous() and dy2.dtype == BF16 and x2.dtype == BF16
            and M * max(Nn, Kk) * 2 < (1 << 31)):
        raise L.DgxError("wgrad_into: dy %s %s x %s %s -- libdgx takes bf16 GPU operands with widths that are multiples of 8"
                         % (tuple(dy2.shape), dy2.dtype, tuple(x2.shape), x2.dtype))
    wgrad_grouped([(g2, dy2.contiguous(), x2.contiguous(), gb)], beta)


def _wgrad_array(problems):
    n = len(problems)
    arr = (L.WgradProblem * n)()
    for i, pr in enumerate(problems):
        g2, dy2, x2 = pr[:3]
        gb = pr[3] if len(pr) > 3 else None
        assert g2.is_contiguous() and dy2.is_contiguous() and x2.is_contiguous()
        assert dy2.dtype == BF16 and x2.dtype == BF16 and g2.dtype == torch.float32
        assert gb is None or (gb.dtype == torch.float32 and gb.is_contiguous() and gb.numel() == dy2.shape[1])
        arr[i].dy, arr[i].x, arr[i].gw = dy2.data_ptr(), x2.data_ptr(), g2.data_ptr()
        arr[i].gb = gb.data_ptr() if gb is not None else None
        arr[i].M, arr[i].Nn, arr[i].Kk = dy2.shape[0], dy2.shape[1], x2.shape[1]
    return arr


def wgrad_group_form(problems):
    """1 if dgx_linear_wgrad_grouped would take this group (<= 32 problems) on the persistent loader-wave kernel, 0 if on the
    split-M kernel (which takes <= 12): dgx_wgrad_grouped_form."""
    n = len(problems)
    return 0 if n == 0 or n > 32 else int(L.lib().dgx_wgrad_grouped_form(_wgrad_array(problems), n))


def wgrad_grouped(problems, beta=1.0):
    """[(g2 fp32 (Nn,Kk), dy2 bf16 (M,Nn), x2 bf16 (M,Kk)[, gb fp32 (Nn) | None]), ...]: g2 = beta*g2 + dy2^T x2 and, with gb, the
    bias gradient gb = beta*gb + dy2.sum(0) from the same pass -- ONE launch (dgx_linear_wgrad_grouped: 256x256 MFMA tiles with an
    M-split sized for the whole group, <= 12 problems; or, for groups that fill the chip with 256x192 tiles for whole rounds, the
    persistent loader-wave kernel, <= 32 problems -- wgrad_group_form tells which)."""
    n = len(problems)
    arr = _wgrad_array(problems)
    lib = L.lib()
    nbytes = int(lib.dgx_wgrad_grouped_workspace_bytes(arr, n))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=problems[0][1].device)
    L.check(lib.dgx_linear_wgrad_grouped(arr, n, float(beta), L.ptr(ws), L.stream()), "dgx_linear_wgrad_grouped")


_READY_SUSPENDED = [False]
# How each hipGraph segment ran in the current step ('g' replayed / captured, 'e' issued eagerly), by segment: a replayed segment signals
# its parameters once each behind the replay, an eager one once per use from its own autograd nodes -- the data-parallel reducer counts
# signals per parameter and keeps one learned count vector PER COMBINATION of modes (engine/ddp.py)
SEGMENT_MODES = {}


class suspend_ready:
    """Context: gradient-ready callbacks (data-parallel reducer) are not fired -- used while a segment is being
    captured into a hipGraph, whose warm-up backward passes are not part of any training step."""

    def __enter__(self):
        self.prev = _READY_SUSPENDED[0]
        _READY_SUSPENDED[0] = True

    def __exit__(self, *a):
        _READY_SUSPENDED[0] = self.prev


def notify_ready(p):
    r = getattr(p, "_dgx_ready", None)
    if r is not None and not _READY_SUSPENDED[0] and not (p.is_cuda and torch.cuda.is_current_stream_capturing()):
        r()


def shadow(p):
    """bf16 view of a parameter: the arena shadow if present, else a cast (CPU tests, pre-arena)."""
    s = getattr(p, "_dgx16", None)
    return s if s is not None else p.detach().to(BF16)


def shadow_t(p):
    """bf16 TRANSPOSE of a matrix parameter, (numel / shape[0], shape[0]): the arena's transposed twin if present (refreshed
    after every optimizer step, solver.FlatArena.refresh_transposes), else a transposed copy (pre-arena use)."""
    s = getattr(p, "_dgx16t", None)
    if s is not None:
        return s
    w = shadow(p)
    return w.reshape(w.shape[0], -1).t().contiguous()


def shadow_padded(p):
    """bf16 operand of a Linear weight (N, K) / bias (N,) whose width N is not a multiple of 8 (cls_score 1454, bbox_pred 4,
    the 1- and 4-channel CenterNet predictors as Linears): N rounded up to 8 with zero rows.  Inside a FlatArena the padding
    lives in the arena itself (`_dgx16p`: the rows behind the parameter's own are part of its segment and stay zero under
    AdamW / EMA because their gradient is zero), so the GEMMs read it without a per-step copy; before that a padded copy."""
    s = getattr(p, "_dgx16g", None)            # member of an arena parameter group: the whole group's rows
    if s is None:
        s = getattr(p, "_dgx16p", None)
    if s is not None:
        return s
    w = shadow(p)
    w = w.reshape(w.shape[0], -1) if w.dim() > 1 else w
    n = w.shape[0]
    if n % 8 == 0:
        return w
    pad = pad8(n) - n
    return torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], 0)


def shadow_t_padded(p):
    """(K, pad8(N)) twin of `shadow_padded(weight)`: the B operand of the input-gradient GEMM."""
    s = getattr(p, "_dgx16tg", None)
    if s is None:
        s = getattr(p, "_dgx16t", None)
    if s is not None and s.shape[1] % 8 == 0 and not getattr(p, "_dgx16t_flipped", False):
        return s
    return shadow_padded(p).t().contiguous()


def grad_padded(p):
    """fp32 gradient view with the padded leading dimension (arena), else None."""
    g = getattr(p, "_dgxgg", None)
    return g if g is not None else getattr(p, "_dgxgp", None)


def group_parameters(first_weight_or_bias, *others, pad_to=8):
    """Tag parameters that always run together on the same input (rows of ONE GEMM operand) as an arena group: see
    solver.FlatArena.  The first argument is the group's handle (its `_dgx16g` / `_dgx16tg` / `_dgxgg` views span the group)."""
    members = (first_weight_or_bias,) + others
    key = ("group", id(first_weight_or_bias))
    for i, q in enumerate(members):
        if hasattr(q, "_dgx_pad_rows"):
            del q._dgx_pad_rows
        q._dgx_group = (key, i, pad_to)
        q._dgx_group_members = members


def ensure_zeroed(p):
    """A gradient segment that the lazy zero_grad (solver.FlatArena.zero_grad(lazy=True)) left to its usual first writer, reached by an
    ACCUMULATING path instead: zero it now (and stop treating it as directly written).  A parameter-group handle stands for the
    whole group: the accumulating GEMM writes every member's rows of the group view."""
    if p is None:
        return
    for q in getattr(p, "_dgx_group_members", None) or (p,):
        slot = getattr(q, "_dgx_arena_slot", None)
        if slot is not None and slot[1] in slot[0]._lazy_pending:
            a, i = slot
            a.g[a.offsets[i]:a.offsets[i] + a.sizes[i]].zero_()
            a._lazy_pending.discard(i)
            a.direct.discard(i)


def accumulate_grad(p, make_grad_fp32, gemm_into=None):
    """Write a gradient into p's arena view.  Returns None if done in place (and signals the
    data-parallel reducer), else the gradient tensor for autograd to accumulate."""
    g = p.grad if p.is_leaf else None
    if g is not None and g.dtype == torch.float32 and getattr(p, "_dgx16", None) is not None:
        ensure_zeroed(p)
        if gemm_into is not None:
            gemm_into(g)
        else:
            g.add_(make_grad_fp32())
        notify_ready(p)
        if gemm_into is not None:             # a group's gradient rows are written by its handle's call: signal every member
            for q in getattr(p, "_dgx_group_members", ()):
                if q is not p:
                    notify_ready(q)
        return None
    return make_grad_fp32().to(p.dtype)


class _LinearFn(torch.autograd.Function):
    """y_pad (M, pad8(N)) = x W^T + b on libdgx's MFMA GEMM; callers slice [..., :N].  The slice's backward hands this
    node a zero-padded gradient, which is exactly the K-padded operand the input- and weight-gradient GEMMs need."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from .gemm_ops import gemm_nt
        w16 = shadow_padded(weight)                  # conv 1x1 weights (Cout,Cin,1,1) are Linear weights
        w16 = w16.reshape(w16.shape[0], -1)
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != BF16:
            x2 = x2.to(BF16)
        x2 = x2.contiguous()
        if w16.shape[1] % 8:
            raise L.DgxError("Linear: in_features %d is not a multiple of 8 (no such layer on the path)" % w16.shape[1])
        y = gemm_nt(x2, w16, shadow_padded(bias) if bias is not None else None)
        ctx.save_for_backward(x2)
        ctx.weight, ctx.bias, ctx.xshape, ctx.xdtype = weight, bias, x.shape, x.dtype
        return y.view(*x.shape[:-1], w16.shape[0])

    @staticmethod
    def backward(ctx, dy):
        from .gemm_ops import gemm_nt
        x2, = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        ensure_zeroed(weight)                  # (this path accumulates: see solver.FlatArena.zero_grad(lazy=True))
        ensure_zeroed(bias)
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        dy2 = dy2.contiguous()
        n = weight.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = gemm_nt(dy2, shadow_t_padded(weight)).view(ctx.xshape)
            if ctx.xdtype != BF16:
                dx = dx.to(ctx.xdtype)
        gw = gb = None
        bp = grad_padded(bias) if bias is not None else None
        bias_in_arena = (bias is not None and ctx.needs_input_grad[2] and bias.is_leaf and bias.grad is not None
                         and bias.grad.dtype == torch.float32 and getattr(bias, "_dgx16", None) is not None and (n % 8 == 0 or bp is not None))
        bias_done = False
        if ctx.needs_input_grad[1]:
            gp = grad_padded(weight)
            fuse_bias = bias_in_arena and weight.is_leaf and weight.grad is not None and getattr(weight, "_dgx16", None) is not None \
                and (n % 8 == 0 or gp is not None)

            def into(g):       # the arena's (padded) gradient rows: accumulated in place, fp32; the bias gradient rides along
                wgrad_into(gp if gp is not None else g.view(g.shape[0], -1), dy2, x2,
                           gb=(bp if bp is not None else bias.grad) if fuse_bias else None)

            def fresh():       # not arena resident: a gradient tensor for autograd to accumulate
                g = torch.empty(dy2.shape[1], x2.shape[1], dtype=torch.float32, device=x2.device)
                wgrad_into(g, dy2, x2, beta=0.0)
                return g[:n].view(weight.shape)
            gw = accumulate_grad(weight, fresh, gemm_into=into if (n % 8 == 0 or gp is not None) else None)
            if fuse_bias:
                bias_done = True
                for q in getattr(bias, "_dgx_group_members", (bias,)):
                    notify_ready(q)
        if bias is not None and ctx.needs_input_grad[2] and not bias_done:
            if bias_in_arena:
                from .swin_block import colsum_into
                colsum_into(bp if bp is not None else bias.grad, dy2)      # bias gradient summed straight into the arena
                for q in getattr(bias, "_dgx_group_members", (bias,)):
                    notify_ready(q)
            else:
                gb = accumulate_grad(bias, lambda: torch.sum(dy2[:, :n], 0, dtype=torch.float32))
        return dx, gw, gb


def linear_padded(x, weight, bias=None):
    """(.., pad8(N)): the GEMM's own output; columns >= N come from zero weight rows and zero bias padding."""
    if not x.is_cuda:
        raise L.DgxError("libdgx ops need GPU (ROCm) tensors; got a %s tensor -- no CPU fallback exists" % x.device)
    with torch.autocast("cuda", enabled=False):
        return _LinearFn.apply(x, weight, bias)


def linear(x, weight, bias=None):
    y = linear_padded(x, weight, bias)
    n = weight.shape[0]
    return y if y.shape[-1] == n else y[..., :n]


class Linear(nn.Linear):
    """nn.Linear parameters (checkpoint-compatible); every forward is libdgx's bf16 MFMA GEMM over the arena shadow (fp32
    accumulation, bf16 result) -- there is no second precision mode and no library route."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if self.out_features % 8:
            self.weight._dgx_pad_rows = pad8(self.out_features)
            if self.bias is not None:
                self.bias._dgx_pad_rows = pad8(self.out_features)

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class _GeluFn(torch.autograd.Function):
    """Exact (erf) GELU on bf16 activations: dgx_gelu_fwd / dgx_gelu_bwd_colsum (nn.GELU of the Swin MLP on the composed path;
    the one-node Swin block has it inside the fc1 GEMM's read-out)."""

    @staticmethod
    def forward(ctx, x):
        x = x.to(BF16).contiguous()
        if x.numel() % 8:
            raise L.DgxError("gelu: element count must be a multiple of 8")
        y = torch.empty_like(x)
        L.check(L.lib().dgx_gelu_fwd(L.ptr(x), L.ptr(y), x.numel(), L.stream()), "dgx_gelu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        N = x.shape[-1]
        M = x.numel() // N
        dy = dy.to(BF16).contiguous()
        dx = torch.empty_like(x)
        lib = L.lib()
        ws = torch.empty(max(int(lib.dgx_gelu_bwd_workspace_bytes(M, N)), 16), dtype=torch.uint8, device=x.device)
        L.check(lib.dgx_gelu_bwd_colsum(L.ptr(dy), L.ptr(x), L.ptr(dx), None, M, N, 0.0, L.ptr(ws), L.stream()), "dgx_gelu_bwd_colsum")
        return dx


def gelu(x):
    return _GeluFn.apply(x)


class GELU(nn.Module):
    def forward(self, x):
        return gelu(x)
