"""Fused AdamW + global-norm clipping over the flat parameter / gradient arenas.

Reference behaviour being replaced: `clip_grad_norm_(max_grad_norm=1.0)` + `optimizer.step()` + `model.zero_grad()` of the
HF loop (transformers/trainer.py:1785-1796, :2535-2545) with the hyper-parameters of
/root/reference/mantis/train/scripts/train_mllava.sh:162-165 (AdamW, lr 1e-5, wd 0), which the reference executes through
DeepSpeed's fused Adam with fp32 master weights.  Here: one sum-of-squares launch, one scalar kernel, one AdamW launch
over the whole trainable arena; the clip coefficient stays on the device (no host sync)."""
import torch

from . import hip_ops as K


class FusedAdamW:
    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0):
        self.model = model
        self.lr, self.betas, self.eps, self.wd, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        model._ensure_grad_arena()
        names = list(model._grad_key)
        self._segments = self._plan(names)
        n = model.grad_arena.numel()
        dev = model.device
        self.master = torch.empty(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.resync_master()
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.last_grad_norm = None
        self._side, self._norm_ws, self._norm_ready = None, None, False

    def _plan(self, names):
        """Maximal runs where the parameter arena and the gradient arena advance together -> (param_off, grad_off, numel)."""
        m = self.model
        segs = []
        for n in names:
            cnt = (m._param(n).numel() + 7) // 8 * 8
            po, go = m._offs[n], m._grad_offs[n]
            if segs and segs[-1][0] + segs[-1][2] == po and segs[-1][1] + segs[-1][2] == go:
                segs[-1] = (segs[-1][0], segs[-1][1], segs[-1][2] + cnt)
            else:
                segs.append((po, go, cnt))
        return segs

    def resync_master(self):
        """Re-snapshot the fp32 master weights from the bf16 parameter arena.  Needed whenever parameter values are replaced
        behind the optimizer's back (`load_reference_state_dict`, `reset_parameters`): `step()` writes bf16(master) over
        the parameters, so a stale master would silently undo the load.  `step()` calls this itself when the model's
        `_param_version` moved; Adam moments are kept."""
        m = self.model
        for p_off, g_off, cnt in self._segments:
            self.master[g_off:g_off + cnt].copy_(m.arena[p_off:p_off + cnt])
        self._seen_version = getattr(m, "_param_version", 0)

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

    def step(self):
        """`self.stream` (optional, e.g. hip_ops.cu_masked_stream): run the pass there -- behind everything queued on the current stream,
        and the current stream resumes behind it -- so that work queued on OTHER streams (the next batch's frozen vision tower on the
        complementary compute units) runs beside it."""
        st = getattr(self, "stream", None)
        if st is None:
            return self._step()
        cur = torch.cuda.current_stream()
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            self._step()
        cur.wait_stream(st)

    def _step(self):
        m = self.model
        if getattr(m, "_param_version", 0) != self._seen_version:
            self.resync_master()
        self.step_count += 1
        scale = None
        if self.max_grad_norm is not None and self.max_grad_norm > 0:
            if not self._norm_ready:                       # no overlap this step (GA window not driven through the hooks, CPU tests)
                K.grad_sumsq(m.grad_arena, self._sumsq, accumulate=False)
            self._norm_ready = False
            scale, self.last_grad_norm = K.clip_scale(self._sumsq, self.max_grad_norm)
        for p_off, g_off, cnt in self._segments:
            K.adamw_flat(m.arena[p_off:p_off + cnt], m.grad_arena[g_off:g_off + cnt], self.master[g_off:g_off + cnt],
                         self.exp_avg[g_off:g_off + cnt], self.exp_avg_sq[g_off:g_off + cnt], self.lr, self.betas[0],
                         self.betas[1], self.eps, self.wd, self.step_count, grad_scale=scale)

    def zero_grad(self, set_to_none=True):
        """model.zero_grad() of the HF loop: dropping the .grad views lets the next backward overwrite instead of accumulate."""
        for p in self.model.parameters():
            if p.requires_grad:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()
