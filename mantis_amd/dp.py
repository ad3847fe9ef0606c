"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

Replaces the torch DDP reducer that `accelerator.prepare` installs in the reference
(transformers/trainer.py:1615-1624; accelerate_config_ddp.yaml:2; 25 MB buckets, mean all-reduce overlapped with backward)
and the `no_sync` handling of non-boundary micro-batches (trainer.py:1744-1757).

Design for MI355X: gradients already live in ONE flat bf16 arena whose layout follows backward completion order, so a
bucket is a contiguous slice -- no gather/scatter copies, no per-parameter hooks.  The engine calls `bucket_ready(key)`
right after the last kernel that writes a bucket (lm_head block, then decoder layers n-1..0, then projector+embedding);
each call enqueues an async mean all-reduce on RCCL's stream, which runs while the next layer's backward kernels execute.
Buckets are per decoder layer (436 MB for Llama-3-8B): large messages keep all 7 xGMI links busy, and RCCL picks the
direct/ring algorithm per size."""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, model, process_group=None):
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._handles = []
        self._avg = None
        # MANTIS_DP_FORCE=1 exercises the collective path even on one rank (used to validate the RCCL calls on a 1-GPU box)
        self._force = os.environ.get("MANTIS_DP_FORCE") == "1" and dist.is_initialized()

    def begin(self):
        self._buckets = self.model.grad_buckets()
        self._handles = []

    def _all_reduce_mean(self, t):
        if self._avg is None:
            self._avg = dist.get_backend(self.pg) == "nccl"
        if self._avg:
            return [dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)], None
        # gloo (CPU tests): no AVG and no bf16 sum -> reduce an fp32 staging copy
        stage = t.float()
        h = dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        return [h], (stage, t)

    def bucket_ready(self, key):
        if self.world == 1 and not self._force:
            return
        b = self._buckets.get(key)
        if b is None:
            return
        hs, post = self._all_reduce_mean(b)
        self._handles.append((hs, post))

    def finish(self):
        for hs, post in self._handles:
            for h in hs:
                h.wait()
            if post is not None:
                stage, t = post
                t.copy_((stage / self.world).to(t.dtype))
        self._handles = []


def shard_batch(global_batch_size, rank, world):
    """rank r takes samples [r*B_local, (r+1)*B_local) of the global batch (DistributedSampler-equivalent)."""
    if global_batch_size % world:
        raise ValueError("global batch must divide evenly across ranks")
    bl = global_batch_size // world
    return range(rank * bl, (rank + 1) * bl)
