"""MI355X-native training step for Mantis' multi-image VLM paths."""
import os

# One HSA hardware queue per stream for the streams this package runs side by side (compute + RCCL's stream + the optional side
# streams).  ROCm's runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, handed out round-robin), and
# kernels that share a hardware queue execute strictly one after the other: measured on 1x MI355X (profiles/r03_dp_world1.md), torch's
# RCCL stream landed on the compute stream's queue and the bucket collectives launched from inside the backward did not overlap a single
# GEMM (0.0 of 51.9 ms) -- with 8 queues the same run overlaps 82.1 of 85.5 ms.  The variable is read when the HIP runtime initialises
# (first device call), so it is set at import, before torch touches the GPU; an explicit user setting wins.
import sys


def _hip_already_up():
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


# what the HIP runtime saw (or will see) when it initialised: the variable only counts if it was in the environment BEFORE the first
# device call.  dp.GradReducer consults hw_queues_at_init() -- and probes the streams themselves -- instead of re-reading os.environ,
# which after the setdefault below says 8 even when the runtime came up earlier with its default of 4.
_HIP_UP_AT_IMPORT = _hip_already_up()
_QUEUES_ENV_AT_IMPORT = os.environ.get("GPU_MAX_HW_QUEUES")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def hw_queues_at_init():
    """Hardware queues the HIP runtime was (or will be) created with -- BEST EFFORT: the value of GPU_MAX_HW_QUEUES that was in the
    environment when HIP initialised, ROCm's default of 4 if HIP was already up when this package was imported and the variable was not
    set.  "Already up" is taken from torch.cuda.is_initialized(), which stays False after torch.cuda.is_available() / device_count() --
    calls that do bring the HIP runtime up (and make it read the variable): in that case this function reports 8 although the runtime
    came up with 4 (round-4 advisor finding).  Nothing here can see that; only a probe of the streams themselves can
    (`dp.hw_queue_probe`, `dp.rccl_overlap_probe`: what `GradReducer` relies on).  A value that does not parse counts as the default."""
    def parse(v, default):
        try:
            return int(str(v).strip())
        except (TypeError, ValueError):
            return default
    if _HIP_UP_AT_IMPORT:
        return parse(_QUEUES_ENV_AT_IMPORT, 4)
    return parse(os.environ.get("GPU_MAX_HW_QUEUES"), 4)
