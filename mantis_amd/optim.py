"""Fused AdamW + global-norm clipping over the flat parameter / gradient arenas.

Reference behaviour being replaced: `clip_grad_norm_(max_grad_norm=1.0)` + `optimizer.step()` + `lr_scheduler.step()` +
`model.zero_grad()` of the HF loop (transformers/trainer.py:1785-1796, :2535-2545) with the hyper-parameters of
/root/reference/mantis/train/scripts/train_mllava.sh:162-165 (AdamW, lr 1e-5, cosine schedule with 3 % warm-up, wd 0), resumed
from the newest `checkpoint-*` (/root/reference/mantis/train/train_mllava.py:281-294), which the reference executes through
DeepSpeed's fused Adam with fp32 master weights.  Here: one sum-of-squares launch, one scalar kernel, one AdamW launch
over the whole trainable arena; the clip coefficient stays on the device (no host sync).

`FusedAdamW` IS a `torch.optim.Optimizer` (round 4), so the reference's own loop can drive it:
  * one `param_group` over the trainable parameters whose `lr` (and betas / eps / weight_decay) is read on EVERY step -- an HF / torch LR
    scheduler (`get_cosine_schedule_with_warmup`, `LambdaLR`) steers it like any optimizer;
  * `state_dict()` / `load_state_dict()` carry the step count and the flat fp32 master (as its two halves, below) / exp_avg / exp_avg_sq
    arenas (the 96 GB of a Mantis-8B run) plus the layout they belong to: `Trainer._save_optimizer_and_scheduler` / `_load_optimizer_and_scheduler`
    (torch.save / torch.load(weights_only=True)) work unchanged;
  * `clip_grad_norm(max_norm)` is the fused `clip_grad_norm_`: norm and clip coefficient on the device, applied inside the next `step()`;
    `trainer.as_hf_trainer()` builds this optimizer in `create_optimizer` and routes the loop's clipping call here.

The fp32 master weights are stored SPLIT (round 5): the upper half of a master, rounded to nearest even, IS the bf16 parameter in the model's
arena (the GEMM operand that exists anyway); `master_lo` holds the low 16 bits, and the sign bit of `exp_avg_sq` (never negative) the one
case those 32 bits leave open (an exact tie that rounded up).  The step then moves 26 instead of 28 bytes per parameter and the optimizer
state shrinks by 2 bytes per parameter (16 GB for Mantis-8B), on the bit-identical fp32 trajectory (csrc/optim.hip: adamw_split_kernel;
GPU check `adamw_split_bitwise`).  `opt.master` joins the halves on demand."""
import torch

from . import hip_ops as K
from .arena import ARENA_ALIGN

STATE_FORMAT = "mantis_fused_adamw/3"       # /3: fp32 master split into master_hi (bf16) + master_lo (int16) + the sign of exp_avg_sq
FP32_MASTER_FORMAT = "mantis_fused_adamw/2"  # /2 (round 4): one flat fp32 `master`; still loadable


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, no_decay=None):
        """no_decay: names (or a predicate name -> bool) of the parameters that take no weight decay -- HF's default optimizer exempts
        biases and norm weights (trainer.py get_decay_parameter_names); `as_hf_trainer()` passes "every 1-D parameter".  None: the decay
        applies to every parameter (the reference's run has weight_decay 0, where it makes no difference)."""
        self.model = model
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        model._ensure_grad_arena()
        names = list(model._grad_key)
        self._names = names
        if no_decay is None:
            nd = lambda n: False
        elif callable(no_decay):
            nd = no_decay
        else:
            nd_set = set(no_decay)
            nd = lambda n: n in nd_set
        self._segments = self._plan(names, nd)
        super().__init__([{"params": [model._param(n) for n in names]}],
                         dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        n = model.grad_arena.numel()
        dev = model.device
        self.master_lo = torch.zeros(n, dtype=torch.int16, device=dev)     # zeros: alignment pads between segments are never written
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.resync_master()
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.last_grad_norm = None
        self._pending_scale = None           # clip coefficient of a clip_grad_norm() call the next step() has to apply
        self._side, self._norm_ws, self._norm_ready = None, None, False

    # hyper-parameters live in the param_group (what a scheduler writes); these are read-only conveniences
    lr = property(lambda self: self.param_groups[0]["lr"])
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])
    wd = property(lambda self: self.param_groups[0]["weight_decay"])

    def add_param_group(self, param_group):
        if getattr(self, "param_groups", None):
            raise NotImplementedError("FusedAdamW runs ONE launch over the model's flat trainable arena: a single param_group "
                                      "(per-parameter weight-decay exemptions: the `no_decay` argument)")
        super().add_param_group(param_group)

    def _plan(self, names, no_decay):
        """Maximal runs where the parameter arena and the gradient arena advance together (alignment pads included when both arenas
        have the same one: pads are zeros in every array, AdamW leaves them zero) and the weight-decay exemption does not change
        -> (param_off, grad_off, numel, decays)."""
        m = self.model
        segs = []
        for n in names:
            cnt = (m._param(n).numel() + 7) // 8 * 8
            po, go, dec = m._offs[n], m._grad_offs[n], not no_decay(n)
            if segs and segs[-1][3] == dec:
                gap_p, gap_g = po - (segs[-1][0] + segs[-1][2]), go - (segs[-1][1] + segs[-1][2])
                if gap_p == gap_g and 0 <= gap_p < ARENA_ALIGN:
                    segs[-1] = (segs[-1][0], segs[-1][1], segs[-1][2] + gap_p + cnt, dec)
                    continue
            segs.append((po, go, cnt, dec))
        return segs

    def resync_master(self):
        """Make the fp32 master weights equal to the bf16 parameters (low halves and tie bits cleared).  Needed whenever parameter values
        are replaced behind the optimizer's back (`load_reference_state_dict`, `reset_parameters`): the low half of the OLD master under a
        NEW parameter would be noise.  `step()` calls this itself when the model's `_param_version` moved; Adam moments are kept."""
        m = self.model
        for p_off, g_off, cnt, _ in self._segments:
            self.master_lo[g_off:g_off + cnt].zero_()
            self.exp_avg_sq[g_off:g_off + cnt].abs_()
        self._seen_version = getattr(m, "_param_version", 0)

    @property
    def master(self):
        """The fp32 master weights in the gradient arena's layout (zeros in the alignment pads): a NEW tensor, joined from the bf16
        parameters, `master_lo` and the tie bits (tests, exports; the step never materialises it)."""
        m = self.model
        out = torch.zeros(self.master_lo.numel(), dtype=torch.float32, device=self.master_lo.device)
        for p_off, g_off, cnt, _ in self._segments:
            K.master_join(m.arena[p_off:p_off + cnt], self.master_lo[g_off:g_off + cnt], self.exp_avg_sq[g_off:g_off + cnt],
                          out=out[g_off:g_off + cnt])
        return out

    def second_moment(self):
        """Adam's exp_avg_sq proper (|stored|: the stored array carries the masters' tie bits in its sign)."""
        return self.exp_avg_sq.abs()

    # ---- checkpoint / resume (train_mllava.py:281-294 resumes from the newest checkpoint-*; HF saves optimizer.state_dict() with torch.save)
    def _layout(self):
        """[name, elements, offset in the flat state] per trainable parameter, in arena order.  The offset (round 6, advisor finding) pins the
        placement rule the state was written under: names and sizes alone cannot tell two alignments apart."""
        m = self.model
        base = m.grad_arena.data_ptr() if getattr(m, "grad_arena", None) is not None else None
        out = []
        for n in self._names:
            p = m._param(n)
            g = getattr(p, "grad", None)
            off = None
            if base is not None and g is not None:
                off = int((g.data_ptr() - base) // g.element_size())
            out.append([n, int(p.numel())] if off is None else [n, int(p.numel()), off])
        return out

    def state_dict(self):
        """The flat state + the step count + the hyper-parameters + the layout the arenas belong to.  `master_lo`, `exp_avg`, `exp_avg_sq` are
        REFERENCES to the live device arrays (no copies); `master_hi` -- the bf16 parameters in the gradient arena's layout, which makes the
        checkpoint self-contained -- is a copy, assembled on the HOST segment by segment (round 6: it used to be a second 16-GB device
        array at every save).  exp_avg_sq is stored AS KEPT: its sign bit is a master's tie bit, its magnitude Adam's second moment --
        `exp_avg_sq_is_signed` says so; consumers that want Adam's v call `second_moment()` or `export_fp32_state()`."""
        g = self.param_groups[0]
        group = {k: (list(v) if isinstance(v, tuple) else v) for k, v in g.items() if k != "params"}
        group["params"] = list(range(len(g["params"])))
        m = self.model
        hi = torch.zeros(self.master_lo.numel(), dtype=m.arena.dtype, device="cpu")
        for p_off, g_off, cnt, _ in self._segments:
            hi[g_off:g_off + cnt].copy_(m.arena[p_off:p_off + cnt])
        return {"format": STATE_FORMAT, "step": int(self.step_count), "layout": self._layout(), "arena_align": ARENA_ALIGN,
                "exp_avg_sq_is_signed": True, "master_hi": hi, "master_lo": self.master_lo, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [group]}

    def export_fp32_state(self):
        """The state as plain fp32 arrays -- `master` (joined), `exp_avg`, `exp_avg_sq` (magnitudes) -- in the round-4 format
        `mantis_fused_adamw/2`: for consumers that want DeepSpeed-style fp32 masters (16 B of new tensors per parameter; `state_dict()`
        itself stays copy-free).  `load_state_dict` accepts it and continues the same trajectory bit for bit."""
        sd = self.state_dict()
        return {"format": FP32_MASTER_FORMAT, "step": sd["step"], "layout": sd["layout"], "arena_align": sd["arena_align"],
                "master": self.master, "exp_avg": self.exp_avg, "exp_avg_sq": self.second_moment(), "param_groups": sd["param_groups"]}

    def load_state_dict(self, state_dict):
        """Resume: step count, fp32 master weights and Adam moments, hyper-parameters (as torch: the checkpoint's lr / betas / eps /
        weight_decay replace the constructor's; an LR scheduler restores its own state separately).  The masters of the checkpoint are
        restored bit for bit -- including their upper halves, i.e. the trainable bf16 parameters are written too (= bf16(master), what the
        model checkpoint of the same step holds) -- so a resumed run continues the fp32 trajectory exactly.  Load the model weights BEFORE
        the optimizer state (the HF loop does).  Also accepts a round-4 state (`mantis_fused_adamw/2`, one fp32 `master` array)."""
        sd = state_dict
        if sd.get("format") not in (STATE_FORMAT, FP32_MASTER_FORMAT):
            raise ValueError(f"not a FusedAdamW state (format {sd.get('format')!r}, expected {STATE_FORMAT!r}); a torch.optim.AdamW state "
                             "holds per-parameter tensors and cannot be mapped onto the flat arenas")
        split = sd["format"] == STATE_FORMAT
        if "arena_align" not in sd and ARENA_ALIGN != 128:
            # a round-4 state carries no alignment: it was written under the then-default 128, which this process does not use
            raise ValueError(f"FusedAdamW.load_state_dict: the checkpoint does not record its arena alignment (a round-4 state, written "
                             f"with 128-element placement) and this process places parameters on {ARENA_ALIGN}-element boundaries "
                             "(MANTIS_ARENA_ALIGN): refusing to load shifted offsets")
        if int(sd.get("arena_align", 128)) != ARENA_ALIGN:
            raise ValueError(f"FusedAdamW.load_state_dict: the checkpoint's flat state was laid out with parameters aligned to "
                             f"{sd.get('arena_align', 128)} elements, this process places them on {ARENA_ALIGN}-element boundaries "
                             "(MANTIS_ARENA_ALIGN): the offsets inside master / exp_avg / exp_avg_sq differ")
        mine, theirs = self._layout(), [list(x) for x in sd["layout"]]
        if [x[:2] for x in theirs] != [x[:2] for x in mine]:
            raise ValueError("FusedAdamW.load_state_dict: the checkpoint's parameter layout (names / sizes of the trainable parameters, in "
                             "arena order) differs from this model's")
        bad = [(a[0], a[2], b[2]) for a, b in zip(theirs, mine) if len(a) > 2 and len(b) > 2 and a[2] != b[2]]
        if bad:
            raise ValueError(f"FusedAdamW.load_state_dict: same parameters, different offsets inside the flat state (first: {bad[0][0]} at "
                             f"{bad[0][1]} in the checkpoint, {bad[0][2]} here): the arenas were laid out under another placement rule")
        keys = ("master_hi", "master_lo", "exp_avg", "exp_avg_sq") if split else ("master", "exp_avg", "exp_avg_sq")
        for k in keys:
            if tuple(sd[k].shape) != tuple(self.exp_avg.shape):
                raise ValueError(f"FusedAdamW.load_state_dict: {k} has {tuple(sd[k].shape)}, expected {tuple(self.exp_avg.shape)}")
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        m = self.model
        if split:
            self.master_lo.copy_(sd["master_lo"])
            differ = 0
            for p_off, g_off, cnt, _ in self._segments:      # the upper halves ARE the parameters
                src = sd["master_hi"][g_off:g_off + cnt].to(device=m.arena.device, dtype=m.arena.dtype)
                differ += int((m.arena[p_off:p_off + cnt] != src).sum())
                m.arena[p_off:p_off + cnt].copy_(src)
            if differ:
                # the intended order is model weights first, optimizer state second, both of the same step: then nothing differs.  Warm-starting
                # OTHER weights with an old optimizer state silently lost them until round 6 (advisor finding)
                import warnings
                warnings.warn(f"FusedAdamW.load_state_dict: {differ} trainable parameter elements differed from the checkpoint's masters and were "
                              "overwritten by them (the masters' upper halves ARE the bf16 parameters).  To keep the model's current weights with "
                              "old moments, load the state and then copy the weights in again (the optimizer re-snapshots its masters).")
        else:                                                # a round-4 checkpoint: split its fp32 masters
            self.exp_avg_sq.abs_()
            for p_off, g_off, cnt, _ in self._segments:
                src = sd["master"][g_off:g_off + cnt].to(device=self.exp_avg.device, dtype=torch.float32).contiguous()
                K.master_split(src, m.arena[p_off:p_off + cnt], self.master_lo[g_off:g_off + cnt], self.exp_avg_sq[g_off:g_off + cnt])
        self.step_count = int(sd["step"])
        g = self.param_groups[0]
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                g[k] = tuple(v) if k == "betas" else v
        self._seen_version = getattr(self.model, "_param_version", 0)
        self._pending_scale, self._norm_ready = None, False

    # ---- gradient norm overlapped with the backward (the reference's clip_grad_norm_ is a separate pass over all gradients,
    # HF:trainer.py:2535-2545).  The engine reports every gradient bucket the moment its last kernel is enqueued (the same hook the
    # data-parallel reducer uses); each bucket's sum of squares is then taken on a SIDE stream behind that point -- and behind the
    # bucket's all-reduce when there is one -- while the compute stream carries on with the backward.  The buckets tile the arena
    # and the side stream runs them in hook order, so the accumulation order is fixed: deterministic.
    def begin_norm(self):
        if self.max_grad_norm is None or self.max_grad_norm <= 0 or not torch.cuda.is_available():
            return False
        if self._side is None:
            self._side = torch.cuda.Stream()
            big = max(b.numel() for b in self.model.grad_buckets().values())
            with torch.cuda.stream(self._side):
                self._norm_ws = torch.empty((K._L.mantis_sumsq_partials(big),), dtype=torch.float32, device=self.model.device)
        self._norm_buckets = self.model.grad_buckets()
        self._norm_first = True
        self._norm_seen = 0
        return True

    def bucket_ready(self, key, after=()):
        """Called from inside the backward right after the last kernel that writes bucket `key` was enqueued (`after`: async
        collective handles that must complete first)."""
        b = self._norm_buckets.get(key)
        if b is None:
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            for h in after:
                h.wait()
            K.grad_sumsq(b, self._sumsq, accumulate=not self._norm_first, ws=self._norm_ws)
        self._norm_first = False
        self._norm_seen += b.numel()

    def end_norm(self):
        if self._norm_seen != self.model.grad_arena.numel():
            raise RuntimeError("gradient-norm overlap: the backward reported buckets covering "
                               f"{self._norm_seen} of {self.model.grad_arena.numel()} gradient elements")
        torch.cuda.current_stream().wait_stream(self._side)
        self._norm_ready = True

    # ---- gradient norm folded into the weight-gradient GEMMs (single rank; the reference's clip_grad_norm_ is a separate pass over all
    # gradients, HF:trainer.py:2535-2545).  On the backward of an accumulation boundary every dW GEMM also leaves the sum of squares of the
    # tiles it stored (hip_ops.linear_dw -> mantis_gemm_bf16_nt_sumsq); end_fold() sums those ~1e5 floats and runs the ordinary
    # sum-of-squares kernel only over what no such GEMM wrote (norm weights, biases, embedding, shapes outside the fused kernel's
    # conditions).  Deterministic (fixed orders everywhere); differs from the separate pass only in summation order.  NOT for data
    # parallel runs: there the norm is that of the REDUCED gradient, which exists only after the exchange.
    class _Fold:
        def __init__(self, opt):
            self.opt, self.used, self.covered = opt, 0, []
            self.stale = []              # ranges whose tile partials must NOT be counted (written again after a fused launch): see take()
            self.slots = {}              # covered range start -> (first tile slot, tiles)

        def take(self, grad_w, tiles):
            o = self.opt
            base, n = o.model.grad_arena.data_ptr(), o.model.grad_arena.numel()
            off = (grad_w.data_ptr() - base) // 2
            if off < 0 or off + grad_w.numel() > n or self.used + tiles > o._fold_ws.numel():
                return None
            # a gradient written twice in one backward (a shared / tied weight, a per-chunk loop with accumulate=True): the second
            # launch's tile sums cover the accumulated values and the first launch's partials are still in the workspace -- the norm
            # would count that range twice.  Decline: the caller runs the plain GEMM, and end_fold() takes the range from the separate
            # pass only if no fused launch covered it; the already-covered range keeps the FIRST launch's partials, which are then stale,
            # so the whole range is handed back to the separate pass (round-4 advisor finding)
            for i, (o2, n2) in enumerate(self.covered):
                if off < o2 + n2 and o2 < off + grad_w.numel():
                    self.stale.append(self.covered.pop(i))          # end_fold() zeroes its tile partials
                    return None
            if any(off < o2 + n2 and o2 < off + grad_w.numel() for o2, n2 in self.stale):
                return None
            self.covered.append((off, grad_w.numel()))
            self.slots[off] = (self.used, tiles)
            ptr = o._fold_ws.data_ptr() + 4 * self.used
            self.used += tiles
            return ptr

        def give_back(self, grad_w, tiles):
            off, _ = self.covered.pop()
            self.slots.pop(off, None)
            self.used -= tiles

    def begin_fold(self):
        """-> collector for the step's `LaunchContext.dw_sumsq` (None when clipping is off).  Call right before the backward of an accumulation boundary."""
        if self.max_grad_norm is None or self.max_grad_norm <= 0:
            return None
        if getattr(self, "_fold_ws", None) is None:
            self._fold_ws = torch.empty(1 << 18, dtype=torch.float32, device=self.model.device)
        self._fold = FusedAdamW._Fold(self)
        return self._fold

    def end_fold(self):
        """The global sum of squares from the tile partials + a sum-of-squares pass over the gradient ranges no fused GEMM covered."""
        f, self._fold = self._fold, None
        m = self.model
        n = m.grad_arena.numel()
        for off, _ in f.stale:                               # ranges written again after their fused launch: their partials do not count
            a, t = f.slots[off]
            self._fold_ws[a:a + t].zero_()
        # uncovered ranges: the many small ones (norm weights, biases: one per layer) go through ONE multi-range launch whose partials land
        # behind the tile partials; the few large ones (embedding, shapes the fused GEMM declined) through the ordinary kernel
        gaps, pos = [], 0
        for off, cnt in sorted(f.covered) + [(n, 0)]:
            lo, hi = pos, (off // 8) * 8                      # parameters start at multiples of 8 elements; pads are zero
            if hi > lo:
                gaps.append((lo, hi - lo))
            pos = max(pos, -(-(off + cnt) // 8) * 8)
        small = [g for g in gaps if g[1] <= (1 << 20)]
        used = f.used
        if small and used + len(small) <= self._fold_ws.numel():
            key = tuple(small)
            if getattr(self, "_fold_small_key", None) != key:        # the layout is static: the table is uploaded once
                self._fold_small = torch.tensor(small, dtype=torch.int64).to(m.device)
                self._fold_small_key = key
            K.sumsq_ranges(m.grad_arena, self._fold_small, self._fold_ws[used:used + len(small)])
            used += len(small)
            gaps = [g for g in gaps if g[1] > (1 << 20)]
        first = True
        if used:
            K.sum_f32(self._fold_ws[:used], self._sumsq, accumulate=False)
            first = False
        for lo, cnt in gaps:
            K.grad_sumsq(m.grad_arena[lo:lo + cnt], self._sumsq, accumulate=not first)
            first = False
        self._norm_ready = True
        self.folded_tiles = f.used

    def clip_grad_norm(self, max_norm=None):
        """The fused `clip_grad_norm_`: global L2 norm of the (reduced) gradient arena and the clip coefficient min(1, max_norm / (norm
        + 1e-6)), both on the device; the coefficient is applied to the gradients INSIDE the next `step()` (one pass over the arena less
        than scaling them in place).  Returns the norm as a 0-d device tensor (what `accelerator.clip_grad_norm_` returns to the HF
        loop); `max_norm = inf` measures without clipping."""
        max_norm = self.max_grad_norm if max_norm is None else max_norm
        if max_norm is None or max_norm <= 0:
            max_norm = float("inf")
        if not self._norm_ready:
            K.grad_sumsq(self.model.grad_arena, self._sumsq, accumulate=False)
        self._norm_ready = False
        self._pending_scale, self.last_grad_norm = K.clip_scale(self._sumsq, min(float(max_norm), 3.0e38))
        return self.last_grad_norm.reshape(())

    def step(self, closure=None):
        """`self.stream` (optional, e.g. hip_ops.cu_masked_stream): run the pass there -- behind everything queued on the current stream,
        and the current stream resumes behind it -- so that work queued on OTHER streams (the next batch's frozen vision tower on the
        complementary compute units) runs beside it."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        st = getattr(self, "stream", None)
        if st is None:
            self._step()
            return loss
        cur = torch.cuda.current_stream()
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            self._step()
        cur.wait_stream(st)
        return loss

    def _step(self):
        m = self.model
        if getattr(m, "_param_version", 0) != self._seen_version:
            self.resync_master()
        self.step_count += 1
        scale, self._pending_scale = self._pending_scale, None
        if scale is None and self.max_grad_norm is not None and self.max_grad_norm > 0:
            # the loop did not clip through clip_grad_norm() (MantisHipTrainer / bench.py): clip here, to the constructor's max_grad_norm
            if not self._norm_ready:                       # no overlap this step (GA window not driven through the hooks, CPU tests)
                K.grad_sumsq(m.grad_arena, self._sumsq, accumulate=False)
            self._norm_ready = False
            scale, self.last_grad_norm = K.clip_scale(self._sumsq, self.max_grad_norm)
        g = self.param_groups[0]                            # read every step: an LR scheduler writes g["lr"]
        lr, (b1, b2), eps, wd = float(g["lr"]), g["betas"], float(g["eps"]), float(g["weight_decay"])
        for p_off, g_off, cnt, decays in self._segments:
            K.adamw_split_flat(m.arena[p_off:p_off + cnt], m.grad_arena[g_off:g_off + cnt], self.master_lo[g_off:g_off + cnt],
                               self.exp_avg[g_off:g_off + cnt], self.exp_avg_sq[g_off:g_off + cnt], lr, float(b1), float(b2), eps,
                               wd if decays else 0.0, self.step_count, grad_scale=scale)

    def zero_grad(self, set_to_none=True):
        """model.zero_grad() of the HF loop: dropping the .grad views lets the next backward overwrite instead of accumulate.  A gradient
        norm that was taken for a step which then never happened (an exception, a skipped optimizer step) dies with the gradients."""
        self._norm_ready = False
        self._pending_scale = None
        for p in self.model.parameters():
            if p.requires_grad:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()
