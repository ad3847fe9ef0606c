"""Per-caller launch options of the GEMM wrappers (`hip_ops`), scoped to ONE call of an engine.

Until round 4 three process-wide switches sat on the hot path: `hip_ops.KERNEL_TIMER` (bench.py's per-launch HIP events),
`hip_ops.DW_SUMSQ` (the optimizer's collector for the gradient norm folded into the weight-gradient GEMMs) and the C library's CU
budget (`mantis_gemm_cu_budget`).  Two trainers or two models in one process interfered through them.  Now the caller owns a
`LaunchContext` and hands it to `engine.step_from_batch(..., launch=ctx)` / `engine.prefetch_vision(..., launch=ctx)`; the engine
installs it for the duration of that call (`with launch_context(ctx)`: thread-local, re-entrant, restored on exit -- also on an
exception) and the wrappers read it at every launch.  The CU budget travels to the library PER CALL (bits 16-27 of the GEMM entry points'
flags, include/mantis_hip.h).  Pure Python: importable without the HIP library (the host-logic tests run the trainer on the oracle)."""
import threading
from contextlib import contextmanager


class LaunchContext:
    """timer:    None, or a list that receives (kernel, algorithmic flops, algorithmic bytes, start event, end event, tag, stream) per
                 GEMM launch (bench.py's roofline, tools/gemm_step_replay.py)
    dw_sumsq:    None, or the optimizer's collector (`optim.FusedAdamW.begin_fold()`): every weight-gradient GEMM then also leaves the
                 sum of squares of what it stored (mantis_gemm_bf16_nt_sumsq)
    shared_gpu:  True when long-running kernels of other queues (RCCL collectives) hold compute units beside the step's kernels: the GEMM
                 entry points get flag 32768 (one tile per workgroup in the 176-row kernel); MantisHipTrainer sets it with an active reducer
    gemm_cus:    compute units the GEMM tile scheduler plans for (0 = the library's default: MANTIS_GEMM_CUS or the whole device);
                 with C RCCL channels active a data-parallel trainer plans for #CU - C"""
    __slots__ = ("timer", "dw_sumsq", "gemm_cus", "shared_gpu")

    def __init__(self, timer=None, dw_sumsq=None, gemm_cus=0, shared_gpu=False):
        self.shared_gpu = bool(shared_gpu)       # collectives of another queue run beside the step's kernels: no persistent GEMM workgroups
        gemm_cus = int(gemm_cus)
        if not 0 <= gemm_cus <= 4095:          # the budget travels in bits 16-27 of the GEMM flags word (round-5 advisor finding)
            raise ValueError(f"gemm_cus must be in [0, 4095] (0 = the library's default), got {gemm_cus}")
        self.timer, self.dw_sumsq, self.gemm_cus = timer, dw_sumsq, gemm_cus

    def __repr__(self):
        return (f"LaunchContext(timer={'on' if self.timer is not None else None}, dw_sumsq={'on' if self.dw_sumsq is not None else None}, "
                f"gemm_cus={self.gemm_cus})")


_DEFAULT = LaunchContext()
_TLS = threading.local()


def current():
    """The context of the innermost active `launch_context` of this thread (a neutral one outside any)."""
    return getattr(_TLS, "ctx", _DEFAULT)


@contextmanager
def launch_context(ctx):
    """Install `ctx` (None = leave the current one) for the calling thread until the block exits."""
    if ctx is None:
        yield current()
        return
    prev = getattr(_TLS, "ctx", _DEFAULT)
    _TLS.ctx = ctx
    try:
        yield ctx
    finally:
        _TLS.ctx = prev
