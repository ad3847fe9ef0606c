"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

Replaces the torch DDP reducer that `accelerator.prepare` installs in the reference
(transformers/trainer.py:1615-1624; accelerate_config_ddp.yaml:2; 25 MB buckets, mean all-reduce overlapped with backward)
and the `no_sync` handling of non-boundary micro-batches (trainer.py:1744-1757).

Design for MI355X: gradients already live in ONE flat bf16 arena whose layout follows backward completion order, so a
bucket is a contiguous slice -- no gather/scatter copies, no per-parameter hooks.  The engine calls `bucket_ready(key)`
right after the last kernel that writes a bucket: the lm_head block, then per decoder layer n-1..0 the three slices
[down_proj] -> [gate|up] -> [norms|q|k|v|o] as their weight-gradient GEMMs land (117 / 235 / 101 MB for Llama-3-8B), then
projector+embedding.  Each call enqueues an async mean all-reduce on RCCL's stream, which runs while the remaining backward
kernels execute; the first byte of a layer moves after that layer's FIRST dW GEMM.  Messages stay >= 100 MB, large enough to
keep all 7 xGMI links of a GPU streaming.

Why torch.distributed is the boundary here and not a `mantis_dp_allreduce_bucket` entry in include/mantis_hip.h: the
communicator (ncclComm_t), its bootstrap (rendezvous over MASTER_ADDR/PORT) and its stream live inside the process group that
the reference's own launcher (`accelerate launch` -> torch.distributed) creates; a C-ABI entry would need either a second
communicator bootstrapped by this library (double the RCCL buffers and a second rendezvous per job) or torch's private
ncclComm_t, which torch does not expose.  The exchange is a single library call per bucket on a contiguous device range --
there is no kernel of ours on that path -- so the drop-in point is `GradReducer`, the object that takes the place of the DDP
reducer, and the C-ABI stays communicator-free.

Knobs (recorded defaults; none can be tuned without an 8-GPU node):
  MANTIS_DP_ALGO = allreduce (default) | rs_ag   reduce-scatter + all-gather on each bucket instead of one all-reduce: on the
                                                 fully connected xGMI node both phases use all 7 links at once
  MANTIS_DP_FORCE = 1                            run the collectives even at world size 1 (validates the RCCL calls on one GPU)
  NCCL_MIN_NCHANNELS / NCCL_MAX_NCHANNELS        RCCL's own knobs: each channel is one workgroup, i.e. one CU taken from the
                                                 GEMM that is running concurrently; left at RCCL's default
`stats` accumulates per-step evidence for bench.py: buckets, bytes, and the time the compute stream spent waiting for RCCL
in `finish()` (= exposed, un-overlapped communication)."""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, model, process_group=None, algo=None):
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self._handles = []
        self._nccl = None
        self.algo = algo or os.environ.get("MANTIS_DP_ALGO", "allreduce")
        if self.algo not in ("allreduce", "rs_ag"):
            raise ValueError(f"MANTIS_DP_ALGO={self.algo!r}: expected 'allreduce' or 'rs_ag'")
        # MANTIS_DP_FORCE=1 exercises the collective path even on one rank (used to validate the RCCL calls on a 1-GPU box)
        self._force = os.environ.get("MANTIS_DP_FORCE") == "1" and dist.is_initialized()
        self.stats = dict(steps=0, buckets=0, bytes=0, exposed_ms=[])
        self._wait_events = []
        if self.active and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
            import warnings
            warnings.warn("GPU_MAX_HW_QUEUES < 8: RCCL's stream may share a hardware queue with the compute stream, which serialises the "
                          "bucket collectives with the backward's kernels (no overlap; profiles/r03_dp_world1.md).  Import mantis_amd before "
                          "the first GPU call, or export GPU_MAX_HW_QUEUES=8.")

    @property
    def active(self):
        return self.world > 1 or self._force

    def begin(self):
        self._buckets = self.model.grad_buckets()
        self._handles = []
        self._pending = set(self._buckets) if self.active else set()

    def _is_nccl(self):
        if self._nccl is None:
            self._nccl = dist.get_backend(self.pg) == "nccl"
        return self._nccl

    def _reduce_mean(self, t):
        """Enqueue the mean over ranks of the flat bf16 slice `t`, in place.  Returns (handles, post) where post is host work to
        run after the handles complete (gloo only)."""
        if self._is_nccl():
            n = t.numel()
            if self.algo == "rs_ag" and n % self.world == 0 and n >= self.world:
                # in-place reduce-scatter into this rank's chunk, then all-gather the chunks back: both phases are one
                # direct exchange per peer on the fully connected node
                chunk = t[self.rank * (n // self.world): (self.rank + 1) * (n // self.world)]
                h1 = dist.reduce_scatter_tensor(chunk, t, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
                h2 = dist.all_gather_into_tensor(t, chunk, group=self.pg, async_op=True)
                return [h1, h2], None
            return [dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)], None
        # gloo (CPU tests): no AVG and no bf16 sum -> reduce an fp32 staging copy
        stage = t.float()
        h = dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        return [h], (stage, t)

    def bucket_ready(self, key):
        """Returns the async handles of the bucket's collective (nccl) so that follow-up work on another stream -- the overlapped
        gradient-norm pass -- can queue behind it; () when nothing was launched."""
        if not self.active:
            return ()
        b = self._buckets.get(key)
        if b is None:
            return ()
        self._pending.discard(key)
        hs, post = self._reduce_mean(b)
        self._handles.append((hs, post))
        self.stats["buckets"] += 1
        self.stats["bytes"] += b.numel() * b.element_size()
        return tuple(hs) if post is None else ()

    def finish(self):
        if self.active and self._pending:
            raise RuntimeError(f"data-parallel reducer: {len(self._pending)} gradient bucket(s) were never signalled by the "
                               f"backward ({sorted(map(str, self._pending))[:3]} ...): ranks would step on un-reduced gradients")
        timed = self.active and torch.cuda.is_available() and self._is_nccl()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for hs, post in self._handles:
            for h in hs:
                h.wait()          # nccl: the compute stream waits for RCCL's stream (no host block); gloo: blocks
            if post is not None:
                stage, t = post
                t.copy_((stage / self.world).to(t.dtype))
        if timed:
            e1.record()
            self._wait_events.append((e0, e1))
        self._handles = []
        self.stats["steps"] += 1

    def collect_exposed_ms(self):
        """Per-step time the compute stream was blocked on RCCL in finish() (call after a device sync)."""
        self.stats["exposed_ms"] += [a.elapsed_time(b) for a, b in self._wait_events]
        self._wait_events = []
        return self.stats["exposed_ms"]


def shard_batch(global_batch_size, rank, world):
    """rank r takes samples [r*B_local, (r+1)*B_local) of the global batch (DistributedSampler-equivalent)."""
    if global_batch_size % world:
        raise ValueError("global batch must divide evenly across ranks")
    bl = global_batch_size // world
    return range(rank * bl, (rank + 1) * bl)
