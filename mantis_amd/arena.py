"""Flat-arena module base: every parameter of a model is a VIEW into one flat bf16 tensor in HBM, every gradient a view into one flat
gradient arena.  Fused projections (q|k|v, gate|up) are zero-copy views over adjacent parameters, data-parallel buckets are contiguous
slices, the optimizer is one launch over the arena (mantis_amd/optim.py).  Parameter names are the reference's state_dict keys.
Shared by the LLaVA path (modeling_llava.py) and the Idefics2 path (modeling_idefics2.py)."""
import os

import torch
from torch import nn


def numel(shape):
    n = 1
    for x in shape:
        n *= x
    return n


#: every parameter (and gradient) starts on a 256-byte boundary of its arena.  The GEMMs fetch their operands in 128-B row segments
#: (row-major) or 256-B pieces (K-major); with 16-byte alignment only, the SigLIP biases (4304 elements) left every decoder weight of
#: Mantis-8B 96 bytes into a cache line, each row segment straddled two lines, and every forward GEMM of the step ran 7 - 13 % slower
#: than the same launch on separately allocated operands (profiles/r04_experiments.md 12).
def _arena_align():
    """MANTIS_ARENA_ALIGN in bf16 elements (the variable exists for A/B measurements; 8 = round 3's layout).  The GEMM, AdamW and
    sum-of-squares kernels assume 16-byte aligned starts: a value below 8 or not a multiple of 8 is refused here, not as a
    ZeroDivisionError in _place() or a misaligned access on the device."""
    raw = os.environ.get("MANTIS_ARENA_ALIGN", "128")
    try:
        v = int(raw)
    except ValueError:
        raise ValueError(f"MANTIS_ARENA_ALIGN={raw!r}: expected a number of bf16 elements (a multiple of 8, >= 8)") from None
    if v < 8 or v % 8:
        raise ValueError(f"MANTIS_ARENA_ALIGN={v}: parameters must start on 16-byte boundaries -- a multiple of 8 elements, at least 8")
    return v


ARENA_ALIGN = _arena_align()


class ArenaModule(nn.Module):
    #: name prefixes of parameters that are frozen on this path (no backward kernels exist for them)
    frozen_prefixes = ()
    #: parameters that must directly follow their predecessor: the later members of a fused projection (`_flat` views: q|k|v, gate|up)
    adjacent_suffixes = ("k_proj.weight", "v_proj.weight", "k_proj.bias", "v_proj.bias", "up_proj.weight")

    # ---- activation checkpointing, with HF's names (PreTrainedModel.gradient_checkpointing_enable / _disable / is_gradient_checkpointing:
    # transformers.Trainer calls the first when TrainingArguments.gradient_checkpointing is set, as the reference's launch script does,
    # /root/reference/mantis/train/scripts/train_mllava.sh:168).  The engines read the flag on every step.
    gradient_checkpointing = False

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        """Keep only every decoder layer's input in the forward and re-run the layer in the backward (same kernels, same bits; ~0.75 GB
        less per Llama-3-8B layer at 5624 rows, one more layer forward of time).  `gradient_checkpointing_kwargs` (use_reentrant ...)
        concern torch.utils.checkpoint and are accepted and ignored: nothing here goes through autograd."""
        self.gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.gradient_checkpointing = False

    @property
    def is_gradient_checkpointing(self):
        return bool(self.gradient_checkpointing)

    def _place(self, items):
        """[(name, numel)] in arena order -> ({name: offset}, total): sizes padded to 8 elements (zeros), starts aligned to ARENA_ALIGN
        except inside a fused projection."""
        offs, off = {}, 0
        for name, n in items:
            if off % ARENA_ALIGN and not name.endswith(self.adjacent_suffixes):
                off = -(-off // ARENA_ALIGN) * ARENA_ALIGN
            offs[name] = off
            off += (n + 7) // 8 * 8
        return offs, off

    def _init_arena(self, specs, device, dtype=torch.bfloat16):
        if dtype != torch.bfloat16:
            raise NotImplementedError("the gfx950 path computes in bf16 (fp32 accumulate); construct with dtype=torch.bfloat16")
        dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        offs, off = self._place([(name, numel(shape)) for name, shape in specs])
        self._specs, self._offs, self._arena_numel = specs, offs, off
        self.arena = torch.zeros(off, dtype=dtype, device=dev)
        for name, shape in specs:
            view = self.arena[offs[name]: offs[name] + numel(shape)].view(shape)
            self._attach(name, nn.Parameter(view, requires_grad=not name.startswith(tuple(self.frozen_prefixes))))
        self.grad_arena = None
        self._grad_offs = None
        self._param_version = 0      # bumped whenever parameter VALUES are replaced wholesale (init / checkpoint load)

    # ------------------------------------------------------------------ module tree with the reference's parameter names
    def _attach(self, dotted, param):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        mod.register_parameter(parts[-1], param)

    def _param(self, name):
        mod = self
        parts = name.split(".")
        for p in parts[:-1]:
            mod = mod._modules[p]
        return mod._parameters[parts[-1]]

    def _flat(self, first, last_incl, rows, cols):
        a = self._offs[first]
        b = self._offs[last_incl] + numel(dict(self._specs)[last_incl])
        assert b - a == rows * cols, (first, last_incl, b - a, rows, cols)
        return self.arena[a:b].view(rows, cols)

    @property
    def device(self):
        return self.arena.device

    @property
    def dtype(self):
        return self.arena.dtype

    def _apply(self, fn, recurse=True):
        # parameters are views of one arena; moving / casting them individually would break the fused layouts
        probe = fn(torch.zeros(1, dtype=self.arena.dtype, device=self.arena.device))
        if probe.dtype != self.arena.dtype or probe.device != self.arena.device:
            raise RuntimeError(f"{type(self).__name__} lives in a flat bf16 arena: construct it with device=... instead "
                               "of calling .to()/.half()/.float()")
        return self

    # ------------------------------------------------------------------ gradients
    def _ensure_grad_arena(self):
        """(Re)attach `.grad` views.  Returns True if the gradients are known to be zero-initialised garbage that the
        next backward may OVERWRITE (i.e. every trainable .grad was None, the state after Trainer's model.zero_grad())."""
        # arena order (not module-tree order): fused projections keep their weights -- and their biases -- adjacent in both arenas
        trainable = sorted(((n, p) for n, p in self.named_parameters() if p.requires_grad), key=lambda np_: self._offs[np_[0]])
        bad = [n for n, _ in trainable if n.startswith(tuple(self.frozen_prefixes))] if self.frozen_prefixes else []
        if bad:
            raise NotImplementedError(f"{bad[0].split('.')[0]}...: frozen on this path (no backward kernels for it); "
                                      f"requires_grad was switched on for {len(bad)} of its parameters")
        key = tuple(n for n, _ in trainable)
        if self.grad_arena is None or self._grad_key != key:
            offs, off = self._place([(n, p.numel()) for n, p in trainable])
            self.grad_arena = torch.zeros(off, dtype=self.arena.dtype, device=self.device)
            self._grad_offs, self._grad_key = offs, key
            self._grad_views = {n: self.grad_arena[offs[n]: offs[n] + p.numel()].view(p.shape) for n, p in trainable}
            self._build_grad_views()
            for n, p in trainable:
                p.grad = None
        all_none = all(p.grad is None for _, p in trainable)
        for n, p in trainable:
            if p.grad is None:
                if not all_none:
                    self._grad_views[n].zero_()
                p.grad = self._grad_views[n]
            elif p.grad.data_ptr() != self._grad_views[n].data_ptr():
                # a foreign gradient tensor was installed: fold it into the arena view
                self._grad_views[n].copy_(p.grad)
                p.grad = self._grad_views[n]
        return all_none

    def set_precision(self, precision):
        """"bf16": every linear on the bf16 MFMA GEMM (the reference's arithmetic; the default).  "fp8": the decoder layers' linears
        (forward, dX, dW) on the fp8 MFMA GEMM with per-tensor e4m3 / e5m2 scaling (decoder_fp8.py; BASELINE configs[4] names it for
        the Qwen2-VL path); towers, projector / connector / merger, lm_head, norms, attention and the loss stay bf16 / fp32.
        "fp8_rowwise": the same linears with finer scales -- one per token / per feature instead of one per tensor (every "NT" GEMM
        operand carries one scale per row, the epilogue multiplies by their outer product); closer to bf16 where a few rows dominate
        a tensor's range, at the price of a two-pass quantiser with no producer-side amax and an unfused SwiGLU backward."""
        from .decoder_fp8 import Fp8Weights
        if precision not in ("bf16", "fp8", "fp8_rowwise"):
            raise ValueError(f"precision {precision!r}")
        tc = self.config.text_config
        if precision != "bf16":
            dims = (tc.hidden_size, tc.intermediate_size, tc.num_attention_heads * tc.head_dim,
                    (tc.num_attention_heads + 2 * tc.num_key_value_heads) * tc.head_dim)
            if any(v % 16 for v in dims):
                raise NotImplementedError(f"fp8 linears need every projection width to be a multiple of 16, got {dims}")
        self.precision = precision
        self.engine.w8 = Fp8Weights(self.lm, rowwise=precision == "fp8_rowwise") if precision != "bf16" else None
        return self

    def _loss_anchor(self):
        # a tiny differentiable input so autograd calls FusedStep.backward (parameters themselves bypass autograd)
        if not hasattr(self, "_anchor") or self._anchor.device != self.device:
            self._anchor = torch.zeros((), device=self.device, dtype=torch.float32, requires_grad=True)
        return self._anchor

    def _autograd_step(self, run):
        """loss tensor whose .backward() publishes the gradients of `run()` (see FusedStep); logits, if any, in self._last_logits."""
        return FusedStep.apply(self, self._loss_anchor(), run)

    def _gflat(self, first, last_incl, rows, cols):
        if first not in self._grad_offs:
            return None
        if last_incl not in self._grad_offs:
            raise NotImplementedError(f"{first}..{last_incl} must be trainable together (fused projection)")
        a = self._grad_offs[first]
        b = self._grad_offs[last_incl] + self._param(last_incl).numel()
        assert b - a == rows * cols
        return self.grad_arena[a:b].view(rows, cols)

    def _bucket_span(self, pred):
        """Contiguous slice of the gradient arena covering the trainable parameters selected by `pred` (None if none), up to the start
        of the next parameter: alignment pads belong to the bucket before them, so consecutive buckets tile the arena."""
        offs, key = self._grad_offs, self._grad_key
        idx = [i for i, n in enumerate(key) if pred(n)]
        if not idx:
            return None
        assert idx == list(range(idx[0], idx[-1] + 1)), "bucket is not contiguous"
        a = offs[key[idx[0]]]
        b = offs[key[idx[-1] + 1]] if idx[-1] + 1 < len(key) else self.grad_arena.numel()
        return self.grad_arena[a:b]

    def copy_state_dict(self, sd, rename=lambda k: k, ignorable=lambda k: False, strict=True):
        """Copy a reference state_dict into the arena views (values rounded to bf16).  Returns the list of missing keys."""
        own = dict(self.named_parameters())
        seen = set()
        with torch.no_grad():
            for k, v in sd.items():
                k2 = rename(k)
                if k2 not in own:
                    if ignorable(k2):
                        continue
                    if strict:
                        raise KeyError(f"unexpected key {k}")
                    continue
                t = torch.as_tensor(v)
                own[k2].copy_(t.to(own[k2].dtype).reshape(own[k2].shape))
                seen.add(k2)
        self._param_version += 1          # optimizers holding fp32 master copies re-snapshot (optim.FusedAdamW.step)
        missing = [k for k in own if k not in seen]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]}...")
        return missing


class FusedStep(torch.autograd.Function):
    """Bridge for callers that drive an ArenaModule through autograd (stock `Trainer.training_step`: `model(**batch).loss.backward()`).
    forward runs the engine's fused forward+backward into a scratch gradient arena (and clears the live arena when it follows a
    `zero_grad(set_to_none=True)`); backward adds `grad_output * scratch` to `.grad`.  (MantisHipTrainer bypasses this and accumulates in
    place with the right scale.)  `run()` = the engine step with compute_grads=True, overwrite_grads=True; it returns the engine's dict."""

    @staticmethod
    def forward(ctx, model, anchor, run):
        # True = every trainable .grad was None (the state after Trainer's model.zero_grad()): the arena views that were just
        # re-attached still hold the PREVIOUS step's gradients and this backward must overwrite them, not add to them.
        overwrite = model._ensure_grad_arena()
        live = model.grad_arena
        if overwrite:
            live.zero_()
        scratch = torch.zeros_like(live)
        # run the step with gradients redirected into `scratch`
        model.grad_arena = scratch
        model._grad_views_live = model._grad_views
        model._grad_views = {n: scratch[o: o + model._param(n).numel()].view(model._param(n).shape)
                             for n, o in model._grad_offs.items()}
        model._build_grad_views()
        try:
            out = run()
        finally:
            model.grad_arena = live
            model._grad_views = model._grad_views_live
            model._build_grad_views()
        model._last_logits = out["logits"]
        ctx.model, ctx.scratch = model, scratch
        return out["loss"].reshape(()).clone()

    @staticmethod
    def backward(ctx, grad_out):
        model, scratch = ctx.model, ctx.scratch
        model._ensure_grad_arena()
        # grad_out stays on the device (0-d fp32; torch multiplies the bf16 arena by it in fp32): no host sync
        model.grad_arena.add_(scratch.mul_(grad_out.to(torch.float32)))
        ctx.scratch = None
        return None, None, None
