"""MI355X-native training step for Mantis' multi-image VLM paths."""
import os

# One HSA hardware queue per stream for the streams this package runs side by side (compute + RCCL's stream + the optional side
# streams).  ROCm's runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, handed out round-robin), and
# kernels that share a hardware queue execute strictly one after the other: measured on 1x MI355X (profiles/r03_dp_world1.md), torch's
# RCCL stream landed on the compute stream's queue and the bucket collectives launched from inside the backward did not overlap a single
# GEMM (0.0 of 51.9 ms) -- with 8 queues the same run overlaps 82.1 of 85.5 ms.  The variable is read when the HIP runtime initialises
# (first device call), so it is set at import, before torch touches the GPU; an explicit user setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
