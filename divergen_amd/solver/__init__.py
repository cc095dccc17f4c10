"""Solver: flat-arena fused AdamW(+value clip)+EMA, WarmupCosineLR, ModelEma surface.

Reference: DG/divergen/custom_solver.py:19-77 (AdamW, one param group per tensor, per-param
clip_grad_value_), D2/solver/build.py:24-112, D2/solver/lr_scheduler.py:171-238, DG/divergen/ema.py.

MI355X-first: all trainable parameters live in ONE fp32 arena (parameters are views into it), so
do gradients, Adam moments and the EMA copy.  A step is one HBM sweep of one kernel
(dgx_adamw_ema_step: 36 B/param) instead of ~400 per-tensor clip + foreach-AdamW launches and a
Python loop of ~400 EMA lerps; and the gradient arena is what the data-parallel reducer all-reduces
in place, bucket by bucket, without flatten/unflatten copies."""
import math
from collections import OrderedDict

import torch

from ..layers import adamw_ema_step


def warmup_cosine_lr(base_lr, it, max_iters, warmup_iters, warmup_factor, warmup_method="linear"):
    """D2/solver/lr_scheduler.py:190-238."""
    if it >= warmup_iters:
        wf = 1.0
    elif warmup_method == "constant":
        wf = warmup_factor
    elif warmup_method == "linear":
        a = it / warmup_iters
        wf = warmup_factor * (1 - a) + a
    else:
        raise ValueError("Unknown warmup method: {}".format(warmup_method))
    return base_lr * wf * 0.5 * (1.0 + math.cos(math.pi * it / max_iters))


_LAZY_ZERO = True      # first-writer weight gradients (FlatArena.zero_grad(lazy=True)); the eager form stays for the parity test of the two


# The transposed weight images refreshed on a side stream (FlatArena.refresh_transposes(overlap=True)).  OFF by default: a loop may
# switch it on only if everything that reads the images runs behind a join_transposes() on its stream -- CustomRCNN.training_losses
# joins behind the backbone + FPN forward (ahead of the early backward and of the mask head), engine.total_loss() joins ahead of the
# backward of the sum; train_net.py and bench.py switch it on.  A model that back-propagates without passing either must leave it off.
OVERLAP_TRANSPOSES = False
_pending_transposes = []


def join_transposes():
    """The current stream waits for every overlapped refresh queued so far (a few ns when there is none)."""
    while _pending_transposes:
        torch.cuda.current_stream().wait_event(_pending_transposes.pop())


class FlatArena:
    """Re-homes every trainable parameter of `model` into one contiguous fp32 buffer and gives each a persistent .grad view
    into a second buffer.  Segments start on multiples of ALIGN = 64 elements: 128 bytes in the bf16 shadow / transposed twin
    (the GEMMs' B operand: a 128-byte row chunk of an LDS-direct load is then ONE cache line; with the 4-element alignment of
    rounds 1-2 every weight behind a 529 x nH bias table sat 48 bytes off and each chunk cost two L2 requests -- measured in
    situ, profiles/r03_gemm_insitu_pmc.txt: +27..46 % requests per launch against the same shapes on aligned operands) and
    256 bytes in the fp32 weight / gradient arenas (the weight-gradient read-outs)."""
    ALIGN = 64

    def __init__(self, model):
        params, seen = [], set()
        for name, p in model.named_parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append((name, p))
        # Parameter GROUPS (`p._dgx_group = (key, position, pad_to)`): layers that always run together on the same input (cls_score
        # + bbox_pred of a box predictor; the 1- and 4-channel CenterNet predictors) are laid out back to back, the group's row
        # count rounded up to `pad_to` with zero rows, so that ONE GEMM (forward, input gradient, weight gradient) serves the
        # group through the views `_dgx16g` / `_dgx16tg` / `_dgxgg`.  The parameters themselves stay separate (state-dict keys,
        # shapes and checkpoints are the reference's); only the allocation order changes.
        groups = {}
        for name, p in params:
            g = getattr(p, "_dgx_group", None)
            if g is not None:
                groups.setdefault(g[0], []).append((g[1], name, p))
        units, placed = [], set()
        for name, p in params:
            if id(p) in placed:
                continue
            g = getattr(p, "_dgx_group", None)
            members = [(name, p)] if g is None else [(n_, q) for _, n_, q in sorted(groups[g[0]], key=lambda t: t[0])]
            units.append(members)
            placed.update(id(q) for _, q in members)
        self.names = [n_ for u in units for n_, _ in u]
        self.params = [q for u in units for _, q in u]
        dev = self.params[0].device
        offs, sizes, n = [], [], 0
        self._groups = []                      # (offset, padded rows, columns, members)
        for u in units:
            if len(u) == 1:
                offs.append(n)
                sizes.append((self.padded_numel(u[0][1]) + self.ALIGN - 1) // self.ALIGN * self.ALIGN)
                n += sizes[-1]
                continue
            cols = u[0][1].numel() // u[0][1].shape[0]
            assert all(q.numel() // q.shape[0] == cols and getattr(q, "_dgx_pad_rows", 0) == 0 for _, q in u), \
                "grouped parameters must share their column count"
            rows = sum(q.shape[0] for _, q in u)
            pad_to = max(q._dgx_group[2] for _, q in u)
            rows_pad = (rows + pad_to - 1) // pad_to * pad_to
            g0 = n
            for _, q in u:
                offs.append(n)
                sizes.append(q.numel())
                n += q.numel()
            tail = (g0 + rows_pad * cols + self.ALIGN - 1) // self.ALIGN * self.ALIGN - n          # zero rows behind the last member belong to its segment
            sizes[-1] += tail
            n += tail
            self._groups.append((g0, rows_pad, cols, [q for _, q in u]))
        self.offsets, self.sizes, self.numel = offs, sizes, n
        self.p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.p16 = torch.zeros(n, dtype=torch.bfloat16, device=dev)   # bf16 shadow, refreshed by the optimizer kernel
        # transposed twin of the shadow: every matrix parameter (rows = shape[0] of its STORED layout) transposed in place
        # of itself -- the B operand of the input-gradient GEMMs (layers/gemm_ops.py); rebuilt by refresh_transposes()
        self.p16t = torch.zeros(n, dtype=torch.bfloat16, device=dev) if dev.type == "cuda" else None
        jobs, tiles = [], 0
        for p, o in zip(self.params, offs):
            v = self.view(self.p, p, o)
            v.copy_(p.data)
            p.data = v
            p.grad = self.view(self.g, p, o)
            p._dgx16 = self.view(self.p16, p, o)
            pr = getattr(p, "_dgx_pad_rows", 0)
            if pr:
                # a Linear whose width is not a multiple of 8 (cls_score 1454, bbox_pred 4): the segment holds pad8(rows) rows,
                # the extra ones zero for ever (zero gradient -> AdamW / EMA leave them at zero); the GEMMs take these views
                cols = p.numel() // p.shape[0]
                shape = (pr, cols) if p.dim() >= 2 else (pr,)
                p._dgx16p = self.p16[o:o + pr * cols].view(shape)
                p._dgxgp = self.g[o:o + pr * cols].view(shape)
            if self.p16t is not None and p.dim() >= 2 and getattr(p, "_dgx_group", None) is None:
                rows = pr or self.stored_rows(p)
                cols = p.numel() // self.stored_rows(p)
                cin = 0
                if getattr(p, "_dgx_flip", False) and getattr(p, "_dgx_ohwi", False) and p.dim() == 4 and p.shape[1] % 64 == 0 \
                        and p.shape[0] % 8 == 0:
                    cin = p.shape[1]              # stride-1 3x3 convolution: tap-flipped twin (Cin, 3, 3, Cout) for its input gradient
                    p._dgx16t = self.p16t[o:o + p.numel()].view(cin, 9 * rows)
                    p._dgx16t_flipped = True
                else:
                    p._dgx16t = self.p16t[o:o + rows * cols].view(cols, rows)
                jobs.append((o, rows | (cols << 32), tiles, cin))
                tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
        for g0, rows_pad, cols, members in self._groups:
            one_d = members[0].dim() == 1
            shape = (rows_pad,) if one_d else (rows_pad, cols)
            v16, vg = self.p16[g0:g0 + rows_pad * cols].view(shape), self.g[g0:g0 + rows_pad * cols].view(shape)
            vt = None
            # a group of stride-1 3x3 convolutions stored (Cout, kh, kw, Cin) (the 1- and 4-channel CenterNet predictors): the
            # twin is the tap-flipped (Cin, 3, 3, rows_pad) image their input-gradient convolution reads, as for single weights
            conv = (not one_d and all(q.dim() == 4 and getattr(q, "_dgx_ohwi", False) and getattr(q, "_dgx_flip", False) for q in members)
                    and members[0].shape[1] % 64 == 0 and rows_pad % 8 == 0)
            if self.p16t is not None and not one_d:
                cin = members[0].shape[1] if conv else 0
                vt = self.p16t[g0:g0 + rows_pad * cols].view((cin, 9 * rows_pad) if conv else (cols, rows_pad))
                jobs.append((g0, rows_pad | (cols << 32), tiles, cin))
                tiles += ((rows_pad + 63) // 64) * ((cols + 63) // 64)
            r0 = 0
            for q in members:
                q._dgx16g, q._dgxgg, q._dgx16tg, q._dgx_group_row0 = v16, vg, vt, r0
                q._dgx16tg_flipped = bool(conv and vt is not None)
                r0 += q.shape[0]
        self._tjobs = torch.tensor(jobs, dtype=torch.int64, device=dev) if jobs else None
        self._ttiles = tiles
        self.sync_shadow()

    @staticmethod
    def view(buf, p, o):
        """The parameter-shaped view of `buf` at offset o.  3x3 convolution weights (tagged `_dgx_ohwi` by
        layers.conv_ops.Conv2d) are STORED (Cout, kh, kw, Cin) -- the K-order of the im2col GEMM -- and exposed in the
        reference's (Cout, Cin, kh, kw) shape as a permuted view: checkpoints see the reference layout, the GEMMs read
        and write the arena directly (no per-step weight permutation, weight gradient accumulated in place)."""
        flat = buf[o:o + p.numel()]
        if getattr(p, "_dgx_ohwi", False) and p.dim() == 4:
            co, ci, kh, kw = p.shape
            return flat.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat.view(p.shape)

    @staticmethod
    def padded_numel(p):
        """Elements of the parameter's arena segment: pad8(rows) x columns for a Linear tagged `_dgx_pad_rows`."""
        pr = getattr(p, "_dgx_pad_rows", 0)
        return pr * (p.numel() // p.shape[0]) if pr else p.numel()

    @staticmethod
    def stored_rows(p):
        """Leading dimension of the parameter's STORED matrix: shape[0] (also for the (Cout, kh, kw, Cin) conv storage)."""
        return p.shape[0]

    def sync_shadow(self):
        """Re-derive the bf16 shadow from the fp32 weights (after init / broadcast / checkpoint load)."""
        self.p16.copy_(self.p)
        self.refresh_transposes()

    def refresh_transposes(self, overlap=False):
        """p16t <- transposes of the matrix parameters of p16: one grouped launch (csrc/transpose.hip).
        overlap (the optimizers' step() asks for it; honoured only under OVERLAP_TRANSPOSES): the launch goes to a side stream behind the
        optimizer kernel and the event behind it is queued for join_transposes() -- the images are the B operands of INPUT-GRADIENT
        GEMMs (and of the mask head's deconvolution), i.e. first read ~8 ms into the next step; 0.8 GB of pure streaming then runs beside
        the next forward's first blocks instead of in front of them."""
        if self._tjobs is None:
            return
        from .. import _lib as L
        if overlap and OVERLAP_TRANSPOSES and self.p16.is_cuda:
            side = self.__dict__.get("_tstream")
            if side is None:
                side = self.__dict__["_tstream"] = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                L.check(L.lib().dgx_transpose_bf16_grouped(self.p16.data_ptr(), self.p16t.data_ptr(), self._tjobs.data_ptr(),
                                                           self._tjobs.shape[0], self._ttiles, L.stream()), "dgx_transpose_bf16_grouped")
                ev = torch.cuda.Event()
                ev.record()
            _pending_transposes.append(ev)
            return
        join_transposes()          # an overlapped refresh still in flight writes the same images
        L.check(L.lib().dgx_transpose_bf16_grouped(self.p16.data_ptr(), self.p16t.data_ptr(), self._tjobs.data_ptr(),
                                                   self._tjobs.shape[0], self._ttiles, L.stream()), "dgx_transpose_bf16_grouped")

    # ---- gradients written by their FIRST writer instead of zero-filled and accumulated (round 3).  The weight / bias gradients of
    # the backbone's Linear layers are produced exactly once per backward pass by the grouped weight-gradient launches
    # (layers/swin_block.py): with beta = 0 those launches neither read the old value nor need it zeroed, so the optimizer's
    # zero_grad(lazy=True) leaves those segments alone -- 0.8 GB less written and 0.8 GB less read per Swin-L step.  `gen` counts
    # zero_grad calls; `written[i]` is the generation in which parameter i was last written directly; `direct` the parameters
    # learned (from the previous step) to be written that way; whatever of them has NOT been written when the gradients are
    # consumed (finish_grads: optimizer step, reducer flush) is zeroed then, so a skipped layer can never leave a stale gradient.
    gen = 0

    def _direct_state(self):
        if "written" not in self.__dict__:
            self.written = [-1] * len(self.params)
            self.direct, self._lazy_pending, self._zero_tables = set(), set(), {}
            for i, q in enumerate(self.params):
                q._dgx_arena_slot = (self, i)
        return self

    def claim_first_write(self, params):
        """True when none of `params` (arena residents) has been written in this generation: the caller may overwrite their
        gradient segments (beta = 0) instead of accumulating.  Marks them written either way.  When the launch as a whole has to
        accumulate (some member was already written: two forward passes before one backward, ACTIVE_COMPARE 'all', gradient
        accumulation), the members the lazy zero_grad left un-zeroed and nobody has written yet are zeroed HERE -- otherwise the
        accumulating launch would add onto last step's gradient."""
        self._direct_state()
        slots = [q._dgx_arena_slot[1] for q in params]
        first = all(self.written[i] != self.gen for i in slots)
        for i in slots:
            if first:
                self.direct.add(i)
            elif i in self._lazy_pending and self.written[i] != self.gen:
                self.g[self.offsets[i]:self.offsets[i] + self.sizes[i]].zero_()
                self.direct.discard(i)
            self.written[i] = self.gen
            self._lazy_pending.discard(i)
        return first

    def zero_grad(self, lazy=False):
        """lazy: skip the segments of the parameters that were written directly in the previous step (the training loop's
        zero_grad); the plain call zeroes everything (trial passes, gradient banks, tests)."""
        from ..layers.swin_block import reset_pending
        reset_pending()
        self._direct_state()
        self.gen += 1
        keep = sorted(i for i in self.direct if self.written[i] == self.gen - 1) if lazy and self.g.is_cuda and _LAZY_ZERO else []
        self.direct = set(keep)
        if not keep:
            self._lazy_pending = set()
            self.g.zero_()
            return
        key = tuple(keep)
        tab = self._zero_tables.get(key)
        if tab is None:                        # complement of the kept segments, cut into work items of <= 64 K floats
            # dgx_zero_ranges_f32 moves 16 bytes per lane: every range starts and ends on a 4-element boundary (the arena's
            # length is a multiple of 4).  A kept segment therefore shrinks to its aligned interior [ceil4(start), floor4(end)):
            # the up-to-3 elements on either side are zeroed with the neighbouring range, and the segment's first writer
            # overwrites them anyway.
            items, pos = [], 0
            for i in keep + [None]:
                end = self.numel if i is None else (self.offsets[i] + 3) // 4 * 4
                end = max(end, pos)
                while pos < end:
                    n = min(end - pos, 65536)
                    items += [pos, n]
                    pos += n
                if i is not None:
                    pos = max(pos, (self.offsets[i] + self.params[i].numel()) // 4 * 4)
            assert all(v % 4 == 0 for v in items), "zero ranges must be 16-byte aligned"
            self._zero_tables.clear()
            tab = self._zero_tables[key] = torch.tensor(items, dtype=torch.int64, device=self.g.device).view(-1, 2)
        from .. import _lib as L
        L.check(L.lib().dgx_zero_ranges_f32(L.ptr(self.g), L.ptr(tab), tab.shape[0], L.stream()), "dgx_zero_ranges_f32")
        self._lazy_pending = set(keep)

    def finish_grads(self):
        """Before the gradients are consumed: zero the directly-written segments that this pass did not write."""
        if self.__dict__.get("_lazy_pending"):
            for i in sorted(self._lazy_pending):
                self.g[self.offsets[i]:self.offsets[i] + self.sizes[i]].zero_()
                self.direct.discard(i)
            self._lazy_pending = set()

    def segment_ends(self):
        return [o + n for o, n in zip(self.offsets, self.sizes)]


# Bumped whenever the STORAGE ORDER of a parameter inside its arena segment changes (round 3: segments on 64-element boundaries,
# grouped predictors, (h, w, c) columns of the box heads' first FC): flat optimizer state of another version is not remapped.
ARENA_LAYOUT_VERSION = 3


class FusedAdamWEMA:
    """Optimizer + EMA over a FlatArena.  API mirrors torch.optim (step/zero_grad/state_dict) as far
    as the training loop and the checkpointer need."""

    def __init__(self, arena, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, clip_value=1.0,
                 ema_decay=0.0, lr_multipliers=None, clip_norm=0.0):
        self.arena, self.lr, self.betas, self.eps = arena, lr, betas, eps
        self.weight_decay, self.clip_value, self.ema_decay = weight_decay, clip_value, ema_decay
        self.clip_norm, self.last_clip = clip_norm, None      # clip_norm > 0: FullModelGradientClippingOptimizer (custom_solver.py:46-60)
        self.m = torch.zeros_like(arena.p)
        self.v = torch.zeros_like(arena.p)
        self.ema = arena.p.clone() if ema_decay > 0 else None
        self.step_count = 0
        self.lr_scale = self.seg_end = None
        if lr_multipliers is not None and any(abs(x - 1.0) > 0 for x in lr_multipliers):
            self.lr_scale = torch.tensor(lr_multipliers, dtype=torch.float32, device=arena.p.device)
            self.seg_end = torch.tensor(arena.segment_ends(), dtype=torch.int64, device=arena.p.device)
        self.param_groups = [{"lr": lr}]

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad(lazy=True)

    def step(self, grad_scale=1.0, found_inf=None):
        self.step_count += 1
        self.arena.finish_grads()
        if self.clip_norm > 0:                 # coefficient stays on the device: [min(1, max_norm / (norm + 1e-6)), norm]
            from ..layers.optim_ops import clip_coef
            self.last_clip = clip_coef(self.arena.g, self.clip_norm, grad_scale)
        adamw_ema_step(self.arena.p, self.arena.g, self.m, self.v, self.ema, self.step_count, self.param_groups[0]["lr"],
                       self.betas, self.eps, self.weight_decay, self.clip_value, grad_scale, self.ema_decay,
                       p_bf16=self.arena.p16 if self.arena.p16.is_cuda else None,
                       lr_scale=self.lr_scale, seg_end=self.seg_end, found_inf=found_inf,
                       grad_scale_dev=self.last_clip if self.clip_norm > 0 else None)
        self.arena.refresh_transposes(overlap=True)

    def _layout(self):
        return {"names": list(self.arena.names), "offsets": list(self.arena.offsets), "sizes": list(self.arena.sizes),
                "layout_version": ARENA_LAYOUT_VERSION}

    def _load_flat(self, dst, src, sd, what):
        """Copy a saved flat state vector into `dst`.  Same layout (names, offsets, sizes, storage-order version): one copy.
        Another arena layout of the SAME storage order: per parameter by name.  Anything else is refused -- a raw copy would put
        the moments on the wrong parameters."""
        cur = self._layout()
        saved = {k: sd.get(k) for k in cur}
        if all(saved[k] is not None and (list(saved[k]) if k != "layout_version" else saved[k]) == cur[k] for k in cur) \
                and src.numel() == dst.numel():
            dst.copy_(src)
            return
        # a checkpoint from before the version field whose names and offsets ARE this arena's.  The version exists because the storage
        # order INSIDE a parameter changed (the (h, w, c) columns of the box heads' first FC, the grouped predictors): names and offsets
        # cannot see that, so the raw copy is allowed only for arenas without such a parameter
        reordered = any(hasattr(q, "_dgx_sd_perm") or getattr(q, "_dgx16g", None) is not None for q in self.arena.params)
        if saved["layout_version"] is None and not reordered and saved["names"] is not None and saved["offsets"] is not None \
                and list(saved["names"]) == cur["names"] and list(saved["offsets"]) == cur["offsets"] and src.numel() == dst.numel():
            dst.copy_(src)
            return
        if saved["layout_version"] != ARENA_LAYOUT_VERSION or saved["names"] is None or saved["sizes"] is None:
            raise RuntimeError("optimizer state '%s' was written by a build with another parameter storage layout (saved version %s, "
                               "this build %d): it cannot be mapped onto this arena -- resume with the model weights only, or from a "
                               "checkpoint of this build" % (what, saved["layout_version"], ARENA_LAYOUT_VERSION))
        where = {n: (o, z) for n, o, z in zip(saved["names"], saved["offsets"], saved["sizes"])}
        dst.zero_()
        for n, o, z in zip(cur["names"], cur["offsets"], cur["sizes"]):
            if n in where:
                so, sz = where[n]
                if sz != z:
                    raise RuntimeError("optimizer state '%s': parameter %s has %d stored elements, this arena holds %d" % (what, n, sz, z))
                dst[o:o + z].copy_(src[so:so + sz])

    def state_dict(self):
        return dict({"step": self.step_count, "exp_avg": self.m, "exp_avg_sq": self.v, "lr": self.param_groups[0]["lr"]}, **self._layout())

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self._load_flat(self.m, sd["exp_avg"], sd, "exp_avg")
        self._load_flat(self.v, sd["exp_avg_sq"], sd, "exp_avg_sq")

    # ---- EMA surface (DG/divergen/ema.py: state_dict / load_state_dict, keys = model state-dict keys)
    def ema_state_dict(self, model):
        out = OrderedDict()
        # (parameters stored in another column order than their state-dict form -- the box heads' first FC -- go out permuted)
        view = {n: (p._dgx_sd_perm[0](self.arena.view(self.ema, p, o)) if hasattr(p, "_dgx_sd_perm") else self.arena.view(self.ema, p, o))
                for n, o, p in zip(self.arena.names, self.arena.offsets, self.arena.params)}
        for k, v in model.state_dict().items():
            out[k] = view[k] if k in view else v
        return out

    def load_ema_state_dict(self, sd):
        for n, o, p in zip(self.arena.names, self.arena.offsets, self.arena.params):
            key = n if n in sd else ("module." + n if "module." + n in sd else None)
            if key is not None:
                src = sd[key]
                if hasattr(p, "_dgx_sd_perm") and tuple(src.shape) == tuple(p.shape):
                    src = p._dgx_sd_perm[1](src.to(self.ema.device))
                self.arena.view(self.ema, p, o).copy_(src)


class FusedSGDEMA(FusedAdamWEMA):
    """The 'SGD' branch of build_custom_optimizer (custom_solver.py:64-68): torch.optim.SGD(momentum, nesterov, one weight decay for
    every parameter) + EMA as ONE pass over the arena (csrc/optim.hip: dgx_sgd_ema_step).  clip_norm > 0 = the reference's
    FullModelGradientClippingOptimizer (:46-60): the whole-arena gradient norm and its clip coefficient stay on the device."""

    def __init__(self, arena, lr, momentum=0.9, nesterov=False, weight_decay=1e-4, clip_value=0.0, clip_norm=0.0, ema_decay=0.0,
                 lr_multipliers=None):
        self.arena, self.lr, self.momentum, self.nesterov = arena, lr, momentum, nesterov
        self.weight_decay, self.clip_value, self.clip_norm, self.ema_decay = weight_decay, clip_value, clip_norm, ema_decay
        self.buf = torch.zeros_like(arena.p) if momentum != 0 else None
        self.ema = arena.p.clone() if ema_decay > 0 else None
        self.step_count = 0
        self.lr_scale = self.seg_end = None
        if lr_multipliers is not None and any(abs(x - 1.0) > 0 for x in lr_multipliers):
            self.lr_scale = torch.tensor(lr_multipliers, dtype=torch.float32, device=arena.p.device)
            self.seg_end = torch.tensor(arena.segment_ends(), dtype=torch.int64, device=arena.p.device)
        self.param_groups = [{"lr": lr}]
        self.last_clip = None                 # device (2,): [coefficient, gradient norm] of the last step with clip_norm

    def step(self, grad_scale=1.0, found_inf=None):
        from ..layers.optim_ops import clip_coef, sgd_ema_step
        self.step_count += 1
        self.arena.finish_grads()
        self.last_clip = clip_coef(self.arena.g, self.clip_norm, grad_scale) if self.clip_norm > 0 else None
        sgd_ema_step(self.arena.p, self.arena.g, self.buf, self.ema, self.step_count, self.param_groups[0]["lr"], self.momentum,
                     self.nesterov, self.weight_decay, self.clip_value, grad_scale, self.last_clip, self.ema_decay,
                     p_bf16=self.arena.p16 if self.arena.p16.is_cuda else None, lr_scale=self.lr_scale, seg_end=self.seg_end,
                     found_inf=found_inf)
        self.arena.refresh_transposes(overlap=True)

    def state_dict(self):
        return dict({"step": self.step_count, "momentum_buffer": self.buf, "lr": self.param_groups[0]["lr"]}, **self._layout())

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        if self.buf is not None and sd.get("momentum_buffer") is not None:
            self._load_flat(self.buf, sd["momentum_buffer"], sd, "momentum_buffer")


def build_optimizer(cfg, model):
    """build_custom_optimizer (custom_solver.py:19-77): OPTIMIZER 'ADAMW' (the shipped configs; value clipping) or 'SGD'
    (value clipping or the full-model norm clipping of :46-60)."""
    s = cfg.SOLVER
    if not (s.USE_CUSTOM_SOLVER and s.OPTIMIZER in ("ADAMW", "SGD")):
        raise NotImplementedError("no optimizer type %s (custom_solver.py:74-75); SOLVER.USE_CUSTOM_SOLVER is what the shipped configs set"
                                  % s.OPTIMIZER)
    ctype = s.CLIP_GRADIENTS.CLIP_TYPE
    if s.CLIP_GRADIENTS.ENABLED and ctype not in ("value", "full_model"):
        raise NotImplementedError("CLIP_GRADIENTS.CLIP_TYPE '%s': built are 'value' and 'full_model' (custom_solver.py:28-60)" % ctype)
    arena = FlatArena(model)
    mult = []
    for name in arena.names:
        m = 1.0
        if "backbone" in name:
            m *= s.BACKBONE_MULTIPLIER
        if any(k in name for k in s.CUSTOM_MULTIPLIER_NAME):
            m *= s.CUSTOM_MULTIPLIER
        mult.append(m)
    clip = s.CLIP_GRADIENTS.CLIP_VALUE if s.CLIP_GRADIENTS.ENABLED else 0.0
    if s.OPTIMIZER == "SGD":
        return FusedSGDEMA(arena, s.BASE_LR, momentum=s.MOMENTUM, nesterov=s.NESTEROV, weight_decay=s.WEIGHT_DECAY,
                           clip_value=clip if ctype == "value" else 0.0, clip_norm=clip if ctype == "full_model" else 0.0,
                           ema_decay=s.MODEL_EMA, lr_multipliers=mult)
    return FusedAdamWEMA(arena, s.BASE_LR, weight_decay=s.WEIGHT_DECAY, clip_value=clip if ctype == "value" else 0.0,
                         clip_norm=clip if ctype == "full_model" else 0.0, ema_decay=s.MODEL_EMA, lr_multipliers=mult)


class WarmupCosineLR:
    def __init__(self, optimizer, max_iters, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear", last_epoch=-1):
        self.opt, self.max_iters = optimizer, max_iters
        self.wf, self.wi, self.wm = warmup_factor, warmup_iters, warmup_method
        self.base_lr = optimizer.lr
        self.last_epoch = last_epoch
        self.step()

    def step(self):
        self.last_epoch += 1
        self.opt.param_groups[0]["lr"] = warmup_cosine_lr(self.base_lr, self.last_epoch, self.max_iters, self.wi, self.wf, self.wm)

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"] - 1
        self.step()


def build_lr_scheduler(cfg, optimizer):
    s = cfg.SOLVER
    if s.LR_SCHEDULER_NAME != "WarmupCosineLR":
        raise NotImplementedError(s.LR_SCHEDULER_NAME)
    return WarmupCosineLR(optimizer, s.MAX_ITER, s.WARMUP_FACTOR, s.WARMUP_ITERS, s.WARMUP_METHOD)
