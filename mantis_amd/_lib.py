"""ctypes binding of libmantis_hip.so (the C-ABI declared in include/mantis_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol cannot be resolved, importing
`mantis_amd.hip_ops` raises.  Build it with `python -m mantis_amd.build` (hipcc, --offload-arch=gfx950)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MANTIS_HIP_LIB") or os.path.join(_HERE, "libmantis_hip.so")   # override: A/B a differently built library (debugging only)

P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_int64
F = ctypes.c_float

# name -> argtypes (restype is always int).  Mirrors include/mantis_hip.h one-to-one.
SIGNATURES = {
    "mantis_pack_plan": [P, P, P, I, I, I, I, L, L, L, I, P, P, P, P, P, P, P, P, P, P, P],
    "mantis_pack_plan_mode": [P, P, P, I, I, I, I, L, L, L, I, I, P, P, P, P, P, P, P, P, P, P, P],
    "mantis_pack_segments": [P, P, P, I, I, I, L, I, P, P, P, P, P, P, P],
    "mantis_pack_rows_fwd": [P, P, P, P, P, I, I, I, I, L, P],
    "mantis_gather_rows": [P, P, P, L, I, P],
    "mantis_scatter_rows": [P, P, P, L, I, P],
    "mantis_embed_grad": [P, P, P, P, P, P, I, I, I, I, L, I, P],
    "mantis_rmsnorm_fwd": [P, P, P, P, L, I, F, P, P],
    "mantis_rmsnorm_bwd_partials": [L],
    "mantis_rmsnorm_bwd": [P, P, P, P, P, P, P, I, P, L, I, P, P],
    "mantis_layernorm_fwd": [P, P, P, P, L, I, F, P],
    "mantis_swiglu_fwd": [P, P, L, I, L, P, P],
    "mantis_swiglu_bwd": [P, P, P, L, I, L, P],
    "mantis_act_fwd": [P, P, L, I, P],
    "mantis_act_bwd": [P, P, P, L, I, P],
    "mantis_add": [P, P, P, L, P],
    "mantis_colsum_partials": [L],
    "mantis_colsum": [P, P, I, P, L, I, L, P],
    "mantis_rope_table": [P, P, P, P, L, I, P],
    "mantis_rope_table_sections": [P, P, P, P, P, L, I, P],
    "mantis_rope_apply": [P, P, P, L, I, I, L, I, P],
    "mantis_transpose": [P, P, I, I, I, L, L, I, I, L, L, L, L, P],
    "mantis_gemm_bf16_nt": [P, L, P, L, P, L, I, I, I, P, P, L, I, P, L, P],
    "mantis_gemm_bf16_nt_fused": [P, L, P, L, P, L, I, I, I, P, I, P, P, L, I, I, P, L, P],
    "mantis_gemm_bf16_nt_sumsq": [P, L, P, L, P, L, I, I, I, I, P, P, L, P],
    "mantis_gemm_workspace_bytes": [I, I, I],
    "mantis_gemm_cu_budget": [I],
    "mantis_gemm_pick_variant": [I, I, I],
    "mantis_gemm_pick_variant_cus": [I, I, I, I],
    "mantis_gemm_bf16_tn_pair": [P, L, P, L, P, L, I, I, P, P, L, P, L, P, L, I, I, P, I, I, P],
    "mantis_gemm_tn_pair_wins": [I, I, I, I, I, I],
    "mantis_gemm_remainder_plan": [I, I, I, I, P, I],
    "mantis_fp8_quantize_ws_floats": [],
    "mantis_fp8_quantize": [P, L, I, L, I, P, L, P, L, P, P, P, I, P],
    "mantis_fp8_quantize_2d": [P, L, I, L, I, P, L, P, P, L, P, P, P],
    "mantis_gemm_fp8_dx_swiglu": [P, L, P, L, P, L, I, I, I, P, P, I, P, L, P, P],
    "mantis_gemm_fp8_nt": [P, L, P, L, P, L, I, I, I, P, P, I, P, P, L, I, P],
    "mantis_attn_fwd": [P, P, P, P, P, P, P, I, I, I, I, I, L, L, L, L, F, I, P],
    "mantis_attn_dsum": [P, P, P, I, I, I, I, L, P],
    "mantis_attn_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, L, L, L, L, L, L, L, L, F, I, P],
    "mantis_attn_bwd_needs_workspace": [I, I, I],
    "mantis_attn_fwd_cross": [P, P, P, P, P, P, I, I, I, I, I, I, L, L, L, L, F, P],
    "mantis_attn_bwd_cross": [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, L, L, L, L, L, L, L, L, F, P],
    "mantis_ce_fwd_bwd": [P, P, I, I, L, F, F, I, P, P, P, P, P],
    "mantis_im2col": [P, P, I, I, I, I, I, I, P],
    "mantis_cast_pad_rows": [P, P, L, I, L, I, P],
    "mantis_vit_assemble": [P, P, P, P, I, I, I, P],
    "mantis_drop_cls": [P, P, I, I, I, P],
    "mantis_navit_prepare": [P, P, I, I, I, I, I, I, P, I, P, P, P, P, P],
    "mantis_adamw": [P, P, P, P, P, L, F, F, F, F, F, F, F, P, P],
    "mantis_adamw_split": [P, P, P, P, P, L, F, F, F, F, F, F, F, P, P],
    "mantis_master_join": [P, P, P, P, L, P],
    "mantis_master_split": [P, P, P, P, L, P],
    "mantis_sum_f32": [P, L, P, I, P],
    "mantis_sumsq_ranges": [P, P, I, P, P],
    "mantis_sumsq_partials": [L],
    "mantis_sumsq": [P, L, P, P, I, P],
    "mantis_clip_scale": [P, F, P, P, P],
    "mantis_stream_create_cu_mask": [I, I, P],
    "mantis_stream_create_priority": [I, P, P],
    "mantis_stream_destroy": [P],
    "mantis_version": [],
}

ERRORS = {-1: "invalid argument", -2: "unsupported shape/alignment", -3: "HIP launch failed"}


class MantisHipError(RuntimeError):
    pass


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X path has no fallback. Build it with `python -m mantis_amd.build`.")
    # PyTorch first: it brings its own HIP runtime (torch/lib/libamdhip64.so), and this library must bind to THAT instance.  Loaded the
    # other way round, libmantis_hip.so pulls in /opt/rocm's runtime before torch initialises its own and the first kernel launch fails
    # ("HIP launch failed" in smoke() when build() -- which loads the library -- ran first in the same process).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    return lib


def check(rc, what):
    if rc != 0:
        raise MantisHipError(f"{what} failed: {ERRORS.get(rc, rc)} (rc={rc})")
