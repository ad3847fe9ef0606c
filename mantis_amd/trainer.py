"""Drop-in for `transformers.Trainer.training_step` (contract 1 of SURVEY.md section 8b).

Reference: transformers/trainer.py:1892-1963, driven by /root/reference/mantis/train/train_mllava.py:312-329:
    model.train(); inputs = _prepare_inputs(inputs); loss = compute_loss(model, inputs); loss /= GA;
    accelerator.backward(loss)  (DDP bucket all-reduce fires here);  return loss.detach()
Here the forward, the backward and the gradient accumulation into `.grad` are one pass over hand-written gfx950 kernels
(`LlavaEngine.step`), with 1/GA folded into the loss-gradient seed, and the data-parallel bucket all-reduces (RCCL) are
launched from inside the backward as each bucket's gradients complete.

`MantisHipTrainer` is transformers-free; `as_hf_trainer()` returns a `transformers.Trainer` subclass whose
`training_step` is this one, for users who keep the stock HF training loop (the reference already subclasses Trainer the
same way in train_intern_vl_25.py:104-122)."""
import os

import torch

from .launch import LaunchContext


class MantisHipTrainer:
    def __init__(self, model=None, gradient_accumulation_steps=1, reducer=None, optimizer=None, fold_norm_into=None):
        """optimizer (a `FusedAdamW`, optional): its gradient-norm pass is taken bucket by bucket on a side stream during the
        backward of the boundary micro-batch instead of as a separate pass before the update (measured slower, off in bench.py).
        fold_norm_into (a `FusedAdamW`, optional): on the backward of an accumulation boundary, when NO gradient exchange is active
        (single rank), the weight-gradient GEMMs also leave the sum of squares of what they store and the optimizer's clip_grad_norm_
        needs no pass of its own over the gradients (`FusedAdamW.begin_fold`)."""
        self.model = model
        self.current_gradient_accumulation_steps = gradient_accumulation_steps
        self.reducer = reducer
        self.optimizer = optimizer
        self.fold_norm_into = fold_norm_into
        self._micro = 0
        # launch options of THIS trainer's steps (per-launch timers: bench.py sets `launch.timer`; the folded gradient norm's collector;
        # the GEMM scheduler's CU budget = what the reducer's RCCL channels leave): handed to the engine with every call, never a
        # process-wide switch -- two trainers / models in one process do not see each other's
        # an ACTIVE reducer's collectives hold compute units beside the backward's kernels: the GEMMs then run without persistent workgroups
        self.launch = LaunchContext(gemm_cus=getattr(reducer, "gemm_cus", 0) or 0, shared_gpu=bool(getattr(reducer, "active", False)))

    def _prepare_inputs(self, inputs):
        # HF:trainer.py:2203-2235 moves tensors to the device; here the engine does the H2D itself (non_blocking) because
        # it needs the host copy of input_ids for the packing-shape bookkeeping.
        if not isinstance(inputs, dict) or "input_ids" not in inputs:
            raise ValueError("training_step expects the batch dict produced by the Mantis collator "
                             "(input_ids, attention_mask, labels, pixel_values)")
        return inputs

    def training_step(self, model, inputs, num_items_in_batch=None, sync=None, next_inputs=None):
        """-> 0-dim detached loss tensor on the model's device, already divided by the accumulation steps.

        sync: is this micro-batch the accumulation boundary (gradients are all-reduced across ranks during its backward)?
        None = count micro-batches (`micro % GA == 0`), which is right for loops that always run whole GA windows; loops that
        can close a window early (HF closes one on the last batch of an epoch, trainer.py `do_sync_step`) must pass it --
        `as_hf_trainer()` passes `accelerator.sync_gradients`, the flag torch DDP's `no_sync` follows in the reference
        (HF:trainer.py:1744-1757).

        next_inputs: the batch the NEXT training_step will receive (what a DataLoader with prefetch_factor already holds).  On the
        accumulation boundary the frozen vision tower of that batch is enqueued on a side stream behind this micro-batch's backward, so
        it runs beside the HBM-bound clip + optimizer step that follows; the next training_step picks the result up (same arithmetic,
        same result -- the tower is frozen -- only earlier).  Engines without `prefetch_vision` ignore it."""
        model.train()
        inputs = self._prepare_inputs(inputs)
        ga = max(1, int(self.current_gradient_accumulation_steps))
        overwrite = model._ensure_grad_arena()
        if hasattr(model.engine, "weights_unchanged"):
            # 2nd.. micro-batch of an accumulation window: no optimizer step since the previous one (fp8 weight copies can be reused)
            model.engine.weights_unchanged = self._micro > 0
        self._micro += 1
        boundary = (self._micro % ga == 0) if sync is None else bool(sync)
        if boundary:
            self._micro = 0                 # the optimizer steps after this micro-batch: the next window starts at 0
        reduce_now = self.reducer is not None and boundary
        # the overlapped gradient norm reads each bucket behind its collective's async handle; only the nccl path hands such handles
        # out (under gloo the mean lands in finish(), after the hook) -- otherwise the optimizer takes its norm after finish()
        red = self.reducer
        red_ok = red is None or not getattr(red, "active", True) or bool(getattr(red, "_is_nccl", lambda: False)())
        norm_now = boundary and self.optimizer is not None and red_ok and self.optimizer.begin_norm()
        if reduce_now:
            self.reducer.begin()
        hook = None
        if reduce_now or norm_now:
            red, opt = self.reducer, self.optimizer

            def hook(key):
                handles = red.bucket_ready(key) if reduce_now else ()
                if norm_now:
                    opt.bucket_ready(key, after=handles or ())
        seg, attn = inputs.get("segment_ids"), inputs["attention_mask"]
        if attn is None or attn.dim() == 4:               # the reference's packed batch (data.py:1609-1671): 4-D block-diagonal mask
            from .data import segments_from_packed
            seg, attn = segments_from_packed(inputs)
        batch = inputs if attn is inputs["attention_mask"] else dict(inputs, attention_mask=attn)
        if batch is not inputs and batch.get("labels") is not None and tuple(batch["labels"].shape) != tuple(batch["input_ids"].shape):
            # the reference's pack_batch concatenates the [1, T] label rows of its items along dim 0 (data.py:1651): [n, T] for the
            # packed row [1, n*T]; row-major that is the row's label vector
            if batch["labels"].numel() != batch["input_ids"].numel():
                raise ValueError(f"packed labels {tuple(batch['labels'].shape)} do not cover the packed row {tuple(batch['input_ids'].shape)}")
            batch["labels"] = batch["labels"].reshape(batch["input_ids"].shape)
        if next_inputs is not None and getattr(self, "prefetch_early", False) and hasattr(model.engine, "prefetch_vision"):
            # early mode: the next batch's frozen tower is queued BEFORE this step's kernels, on `prefetch_stream` (meant to be a stream of
            # the lowest hardware-queue priority, hip_ops.priority_stream): its workgroups take the compute units this step's kernels
            # leave idle (incomplete tile rounds, epilogues) during the whole forward + backward
            eng, nxt_, launch_ = model.engine, next_inputs, self.launch

            def queue_tower():
                ev = torch.cuda.Event()
                ev.record()
                eng.prefetch_vision(nxt_, after_event=ev, stream=getattr(self, "prefetch_stream", None), launch=launch_)
            # prefetch_point "backward" (experiment, MANTIS_PREFETCH_POINT): queued where the backward starts instead of at the top of the step
            if getattr(self, "prefetch_point", os.environ.get("MANTIS_PREFETCH_POINT", "forward")) == "backward" and hasattr(eng, "vision_forward"):
                eng._on_backward_start = queue_tower
            else:
                queue_tower()
            next_inputs = None
        fold = None
        if boundary and self.fold_norm_into is not None and not norm_now and (self.reducer is None or not self.reducer.active):
            fold = self.fold_norm_into.begin_fold()
        self.launch.dw_sumsq = fold
        try:
            out = model.engine.step_from_batch(batch, launch=self.launch, grad_scale=1.0 / ga, loss_scale=1.0 / ga, compute_grads=True,
                                               overwrite_grads=overwrite, on_bucket_ready=hook, segment_ids=seg)
        finally:
            self.launch.dw_sumsq = None
        if fold is not None:
            self.fold_norm_into.end_fold()
        if reduce_now:
            self.reducer.finish()
        if norm_now:
            self.optimizer.end_norm()
        if next_inputs is not None and boundary and hasattr(model.engine, "prefetch_vision"):
            ev = torch.cuda.Event()
            ev.record()                       # end of this window's backward (and gradient reduction) on the compute stream
            model.engine.prefetch_vision(next_inputs, after_event=ev, stream=getattr(self, "prefetch_stream", None), launch=self.launch)
        return out["loss"].reshape(()).detach()


def _on_gpu():
    return torch.cuda.is_available()


class _LookAhead:
    """Iterator over a DataLoader that stays ONE batch ahead of its consumer and remembers what it handed out: `next_of(batch)` is the batch
    the loop will pass to its next `training_step` -- what the early tower prefetch needs.  HF's loop draws a whole accumulation window
    at a time (`get_batch_samples`, transformers >= 4.46) or one batch per iteration (older): either way the batch after the one in hand
    is the next entry of `recent`, or the look-ahead slot."""

    def __init__(self, source):
        """source: a DataLoader (iterated here) or an iterator the loop already holds (kept as `source`, for identity checks)"""
        from collections import deque
        self.source = source
        self.it = source if hasattr(source, "__next__") else iter(source)
        self.ahead = deque()
        self.recent = deque(maxlen=256)

    def __iter__(self):
        return self

    def __next__(self):
        if not self.ahead:
            self.ahead.append(next(self.it))          # StopIteration ends the epoch
        item = self.ahead.popleft()
        try:
            self.ahead.append(next(self.it))
        except StopIteration:
            pass
        self.recent.append(item)
        return item

    def next_of(self, item):
        r = list(self.recent)
        for i in range(len(r) - 1, -1, -1):
            if r[i] is item:
                return r[i + 1] if i + 1 < len(r) else (self.ahead[0] if self.ahead else None)
        return None


def _with_look_ahead(loader, owner):
    """The loader ITSELF, its class swapped for a one-off subclass whose iterators are `_LookAhead`s (the trainer finds the live iterator
    under `owner._mantis_iter`).  Round 5 wrapped the loader in a stand-in object instead: accelerate's `skip_first_batches` (mid-epoch resume)
    then no longer recognised a DataLoaderShard / DataLoaderDispatcher, rebuilt a plain DataLoader from the forwarded attributes and the
    shard's own behaviour (set_epoch, gradient_state registration, dispatcher mode) was lost for that epoch -- and `copy.copy(wrapper)`
    recursed forever in `__getattr__` (advisor finding).  A subclass instance IS a DataLoaderShard for every isinstance check."""
    base = type(loader)
    if getattr(base, "_mantis_look_ahead", False):
        loader._mantis_owner = owner
        return loader

    def __iter__(self):
        it = _LookAhead(base.__iter__(self))
        own = getattr(self, "_mantis_owner", None)
        if own is not None:
            own._mantis_iter = it
        return it
    try:
        cls = type("LookAhead" + base.__name__, (base,), {"__iter__": __iter__, "_mantis_look_ahead": True})
        # through object's own slot: accelerate's DataLoaderAdapter shadows `__class__` with a read-only property (it poses as the DataLoader
        # class it wraps), which a plain assignment would hit
        object.__dict__["__class__"].__set__(loader, cls)
        loader._mantis_owner = owner
    except (TypeError, AttributeError):      # a loader type that cannot be re-classed: the iterator-level hook (get_batch_samples) still applies
        pass
    return loader


def as_hf_trainer():
    """transformers.Trainer subclass using the fused step (imported lazily: transformers is optional)."""
    from transformers import Trainer

    class MantisHipHFTrainer(Trainer):
        """`mantis_fused_optimizer` (class attribute, default True): `create_optimizer` builds `optim.FusedAdamW` from the TrainingArguments
        (learning_rate, adam_beta1/2, adam_epsilon, weight_decay with HF's exemption of biases and norm weights, max_grad_norm) instead
        of torch's AdamW, so the loop's `lr_scheduler` (cosine + warm-up in train_mllava.sh:162-165), its checkpoints
        (`_save_optimizer_and_scheduler`) and its resume (train_mllava.py:281-294) drive the fused clip + AdamW pass; the loop's
        `clip_grad_norm_` call is routed to `FusedAdamW.clip_grad_norm`."""
        mantis_fused_optimizer = True
        #: "early": `training_step` hands the batch the loop will pass NEXT to `MantisHipTrainer.training_step(next_inputs=...)`, whose frozen
        #: vision tower is then queued on a lowest-priority stream at the start of the step and fills the compute units the step's GEMMs leave
        #: idle (measured -3 ... -5 ms per 8B step, same arithmetic: the tower is frozen); None: every step computes its own tower in line.
        #: The look-ahead costs one extra batch held in host memory.
        mantis_prefetch = "early"
        #: keep the batches the DataLoader yields on the HOST (pinned): the engine uploads them itself, non-blocking, and needs the host copy of
        #: input_ids / attention_mask / labels for its shape bookkeeping -- a batch that accelerate already moved to the GPU costs a
        #: device-to-host copy and a host synchronisation per step
        mantis_host_batches = True

        def get_train_dataloader(self):
            dl = super().get_train_dataloader()
            if self.mantis_host_batches and getattr(dl, "device", None) is not None:
                try:
                    dl.device = None                  # accelerate.data_loader.DataLoaderShard: no send_to_device
                except Exception:
                    pass
            if not self.mantis_prefetch:
                return dl
            return _with_look_ahead(dl, self)

        def get_batch_samples(self, epoch_iterator, num_batches, *args, **kwargs):
            """transformers >= 4.46 draws every accumulation window through this method, with the iterator the loop ACTUALLY walks -- also the
            one over the loader `accelerate.skip_first_batches` builds on a mid-epoch resume, which `get_train_dataloader` never sees.  An
            iterator that is not a look-ahead yet becomes one here (iterator-level wrapping: the advisor's remedy for round 5's lost prefetch)."""
            if self.mantis_prefetch and not isinstance(epoch_iterator, _LookAhead):
                la = getattr(self, "_mantis_iter", None)
                if la is None or getattr(la, "source", None) is not epoch_iterator:
                    la = _LookAhead(epoch_iterator)
                    self._mantis_iter = la
                epoch_iterator = la
            return super().get_batch_samples(epoch_iterator, num_batches, *args, **kwargs)

        def _fused(self):
            from .optim import FusedAdamW
            opt = self.optimizer
            while opt is not None and not isinstance(opt, FusedAdamW) and hasattr(opt, "optimizer"):
                opt = opt.optimizer                   # accelerate.AcceleratedOptimizer
            return opt if isinstance(opt, FusedAdamW) else None

        def create_optimizer(self, model=None):
            if self.optimizer is not None or not self.mantis_fused_optimizer:
                return super().create_optimizer(model) if model is not None else super().create_optimizer()
            from .optim import FusedAdamW
            m = self.model if model is None else model
            inner = m.module if hasattr(m, "module") else m
            a = self.args
            self.optimizer = FusedAdamW(inner, lr=a.learning_rate, betas=(a.adam_beta1, a.adam_beta2), eps=a.adam_epsilon,
                                        weight_decay=a.weight_decay, max_grad_norm=a.max_grad_norm,
                                        no_decay=lambda n: inner._param(n).dim() <= 1)
            # transformers < 4.5x clips inline through the accelerator (no _clip_grad_norm method to override): route that call too
            acc = getattr(self, "accelerator", None)
            if acc is not None and not getattr(acc, "_mantis_clip_patched", False):
                plain = acc.clip_grad_norm_

                def clip(parameters, max_norm, norm_type=2):
                    opt = self._fused()
                    if opt is None or norm_type != 2:
                        return plain(parameters, max_norm, norm_type)
                    return opt.clip_grad_norm(max_norm)
                acc.clip_grad_norm_ = clip
                acc._mantis_clip_patched = True
            return self.optimizer

        def _clip_grad_norm(self, model):
            opt = self._fused()
            if opt is None:
                return super()._clip_grad_norm(model)
            return opt.clip_grad_norm(self.args.max_grad_norm)

        def training_step(self, model, inputs, num_items_in_batch=None):
            ga = getattr(self, "current_gradient_accumulation_steps", None) or self.args.gradient_accumulation_steps
            impl = getattr(self, "_mantis_impl", None)
            if impl is None:
                impl = self._mantis_impl = MantisHipTrainer(model, ga, getattr(self, "mantis_reducer", None))
            impl.current_gradient_accumulation_steps = ga
            impl.fold_norm_into = self._fused() if getattr(self, "mantis_reducer", None) is None else None
            inner = model.module if hasattr(model, "module") else model
            # the loop's own notion of the accumulation boundary (set before every training_step call by HF's inner loop,
            # including the short window at the end of an epoch) -- never a private counter
            acc = getattr(self, "accelerator", None)
            sync = None if acc is None else bool(acc.sync_gradients)
            nxt = None
            it = getattr(self, "_mantis_iter", None)
            if self.mantis_prefetch and it is not None and hasattr(inner.engine, "prefetch_vision") and _on_gpu():
                nxt = it.next_of(inputs)
                if nxt is None and not any(r is inputs for r in it.recent) and not getattr(self, "_mantis_warned_no_lookahead", False):
                    # the loop walks an iterator the look-ahead never saw (old transformers without get_batch_samples after a mid-epoch
                    # resume): the tower runs in line for those steps -- same results, a few ms slower -- and that is said once
                    import warnings
                    warnings.warn("mantis_prefetch: this step's batch did not come through the look-ahead iterator; the next batch's vision "
                                  "tower is not prefetched for now (results are unaffected)")
                    self._mantis_warned_no_lookahead = True
                if self.mantis_prefetch == "early" and not getattr(impl, "prefetch_early", False):
                    from . import hip_ops
                    impl.prefetch_early = True
                    impl.prefetch_stream = hip_ops.priority_stream(1)
            return impl.training_step(inner, inputs, num_items_in_batch, sync=sync, next_inputs=nxt)

    return MantisHipHFTrainer
