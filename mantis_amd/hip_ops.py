"""Tensor-level wrappers over the C-ABI (include/mantis_hip.h): torch is used ONLY for device memory and the stream handle.

Every function launches hand-written gfx950 kernels through libmantis_hip.so; nothing here falls back to torch math.
Conventions: activations are 2-D bf16 [rows, cols] (possibly strided rows), weights are [out, in] as in nn.Linear."""
import math

import torch

from . import _lib
from .launch import LaunchContext, current as _launch, launch_context      # noqa: F401  (re-exported: K.LaunchContext, K.launch_context)

_L = _lib.load()
BF16 = torch.bfloat16
ACT_KIND = {"gelu": 0, "gelu_pytorch_tanh": 1, "quick_gelu": 2, "silu": 3}
GEMM_ACT = {None: 0, "gelu": 1, "gelu_pytorch_tanh": 2, "quick_gelu": 3}
# Launch options (per-launch HIP events for bench.py, the optimizer's sum-of-squares collector, the GEMM scheduler's CU budget) come from
# the caller's LaunchContext (mantis_amd/launch.py), installed by the engine for the duration of one step -- no module-level switches.
CUS_SHIFT = 16               # bits 16-27 of the GEMM entry points' flags: the CU budget of that launch (include/mantis_hip.h)
SK_INKERNEL = 16384          # flag: K-split remainder tiles reduced inside the GEMM kernel (round 4) instead of by the finishing kernel
SHARED_GPU = 32768           # flag: RCCL collectives hold CUs beside this launch (LaunchContext.shared_gpu): no persistent GEMM workgroups


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw accessor is ~30x cheaper than building a Stream
    object; 1700 launches per 8B step go through here)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    """Device address of a tensor argument (None -> NULL).  A host tensor handed to a kernel is a memory fault that takes the process
    down at the next synchronisation, far from its cause: refuse it here."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError(f"a {tuple(t.shape)} {t.dtype} tensor on {t.device} was passed to a gfx950 kernel (device tensors only)")
    return t.data_ptr()


def _chk2d(t, name):
    if t.dtype != BF16 or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
        raise ValueError(f"{name}: expected a 2-D bf16 CUDA tensor with unit column stride, got {t.dtype} {tuple(t.shape)} "
                         f"{t.stride()} {t.device}")


def pad8(n):
    return (n + 7) // 8 * 8


_GEMM_WS = {}


def _gemm_workspace():
    """Caller-owned split-K workspace of the ring GEMM (include/mantis_hip.h: mantis_gemm_workspace_bytes): one zero-initialised
    buffer per (device, stream); launches on a stream are ordered, and every launch leaves the ticket counters at zero."""
    key = (torch.cuda.current_device(), _stream())
    ws = _GEMM_WS.get(key)
    if ws is None:
        n = _L.mantis_gemm_workspace_bytes(0, 0, 0)
        ws = _GEMM_WS[key] = torch.zeros(n + 256, dtype=torch.uint8, device="cuda")
    off = (-ws.data_ptr()) % 256
    return ws.data_ptr() + off, ws.numel() - 256


# ----------------------------------------------------------------------------------------------------------- GEMM family
def gemm_nt(a, b, bias=None, act=None, residual=None, out=None, accumulate=False, n_valid=None, k=None, ldc=None, variant=0,
            a_kmajor=False, b_kmajor=False, cus=None, sk_inkernel=False):
    """out[M,N] = epi(A . B^T) with A = a[M,K] (or a[K,M] if a_kmajor), B = b[N,K] (or b[K,N] if b_kmajor).
    `k` overrides the contraction length (zero-padded operands).  cus: CU budget of this launch (None = the LaunchContext's);
    sk_inkernel: round 4's in-kernel reduction of K-split remainder tiles (tests: bit-identical to the finishing kernel)."""
    _chk2d(a, "a"), _chk2d(b, "b")
    if a_kmajor:
        K, M = a.shape
    else:
        M, K = a.shape
    if b_kmajor:
        N = b.shape[1]
    else:
        N = b.shape[0]
    N = N if n_valid is None else n_valid
    K = K if k is None else k
    if out is None:
        out = torch.empty((M, N if ldc is None else ldc), dtype=BF16, device=a.device)   # ldc > N: padded row stride
    else:
        _chk2d(out, "out")
    C = out
    ctx = _launch()
    if cus is not None and not 0 <= int(cus) <= 4095:
        raise ValueError(f"cus must be in [0, 4095] (0 = the library's default), got {cus}")
    flags = (1 if bias is not None else 0) | (GEMM_ACT[act] << 1) | (16 if residual is not None else 0) | (32 if accumulate else 0) \
        | (variant << 8) | (4096 if a_kmajor else 0) | (8192 if b_kmajor else 0) | ((ctx.gemm_cus if cus is None else cus) << CUS_SHIFT) \
        | (SK_INKERNEL if sk_inkernel else 0) | (SHARED_GPU if ctx.shared_gpu else 0)
    prof = ctx.timer
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()          # torch's current stream == the stream handed to the kernel below
    wsp, wsn = _gemm_workspace()
    rc = _L.mantis_gemm_bf16_nt(_p(a), a.stride(0), _p(b), b.stride(0), _p(C), C.stride(0), M, N, K, _p(bias), _p(residual),
                                0 if residual is None else residual.stride(0), flags, wsp, wsn, _stream())
    _lib.check(rc, f"gemm M={M} N={N} K={K} akm={a_kmajor} bkm={b_kmajor}")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        # algorithmic bytes: both operands read once, the result written once (+ residual / accumulate read)
        by = 2.0 * (M * K + N * K + M * N * (1 + (residual is not None) + bool(accumulate)))
        lay = ("T" if a_kmajor else "N") + ("N" if b_kmajor else "T")      # A^T? / B as [N,K] ("T": B^T is applied) -- NT = forward, NN = dX, TN = dW
        epi = "+".join(x for x, on in (("bias", bias is not None), (str(act), act is not None), ("res", residual is not None), ("acc", accumulate)) if on)
        prof.append(("gemm_nt_kernel", 2.0 * M * N * K, by, e0, e1, (M, N, K, lay, epi or "-"), _stream()))
    return out


# ----------------------------------------------------------------------------------------------------------- fp8 linears
FP8_E4M3, FP8_E5M2 = 0, 1
_Q_WS = {}


class Fp8Tensor:
    """q uint8 [rows, cols] (row-major fp8), qt uint8 [cols, rows_pad] (transposed copy, zero tail) or None, state fp32[3] on the device
    = {amax, scale, dequant factor}."""
    __slots__ = ("q", "qt", "state", "fmt", "rows", "cols", "row_dequant", "col_dequant")

    def __init__(self, q, qt, state, fmt, rows, cols, row_dequant=None, col_dequant=None):
        self.q, self.qt, self.state, self.fmt, self.rows, self.cols = q, qt, state, fmt, rows, cols
        self.row_dequant, self.col_dequant = row_dequant, col_dequant      # fp8_quantize(rowwise=True): fp32 [rows] / [cols]

    @property
    def rowwise(self):
        return self.state is None

    @property
    def dequant(self):
        """dequant factor(s) of the row-major copy `q`: fp32[1], or fp32[rows] when quantised rowwise"""
        return self.row_dequant if self.state is None else self.state[2:3]

    @property
    def dequant_t(self):
        """dequant factor(s) of the transposed copy `qt`: the same fp32[1], or fp32[cols] when quantised rowwise"""
        return self.col_dequant if self.state is None else self.state[2:3]


def fp8_quantize(x, fmt=FP8_E4M3, transposed=True, rowmajor=True, amax=None, rowwise=False):
    """Per-tensor just-in-time quantisation of a bf16 matrix: q = cvt(clamp(x * FMAX / amax)).  Returns Fp8Tensor.
    amax: max |x| already taken by x's producer -- fp32[1] (gemm_fp8_dx_swiglu) or an amax_parts_buffer() filled by rmsnorm_fwd /
    rmsnorm_bwd / swiglu_fwd -- the amax pass is skipped.
    rowwise=True: one scale per row for q and one per column of x for qt (mantis_fp8_quantize_2d); `amax` is not used."""
    _chk2d(x, "x")
    rows, cols = x.shape
    dev = x.device
    if rowwise:
        if not (rowmajor or transposed):
            raise ValueError("fp8_quantize: nothing to produce")
        rp = (rows + 15) // 16 * 16
        ws = torch.empty(rows + cols, dtype=torch.float32, device=dev)
        q = torch.empty((rows, cols), dtype=torch.uint8, device=dev) if rowmajor else None
        qt = torch.empty((cols, rp), dtype=torch.uint8, device=dev) if transposed else None
        rd = torch.empty(rows, dtype=torch.float32, device=dev) if rowmajor else None
        cd = torch.empty(cols, dtype=torch.float32, device=dev) if transposed else None
        _lib.check(_L.mantis_fp8_quantize_2d(_p(x), rows, cols, x.stride(0), fmt, _p(q), cols, _p(rd), _p(qt), rp, _p(cd), _p(ws), _stream()),
                   f"fp8_quantize_2d {rows}x{cols}")
        return Fp8Tensor(q, qt, None, fmt, rows, cols, rd, cd)
    if not rowmajor:
        raise ValueError("fp8_quantize: the per-tensor quantiser always writes the row-major copy (rowmajor=False needs rowwise=True)")
    key = (dev, _stream())          # one scratch buffer per (device, stream): launches on a stream are ordered
    ws = _Q_WS.get(key)
    if ws is None:
        ws = _Q_WS[key] = torch.zeros(_L.mantis_fp8_quantize_ws_floats(), dtype=torch.float32, device=dev)
    rp = (rows + 15) // 16 * 16
    q = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
    qt = torch.empty((cols, rp), dtype=torch.uint8, device=dev) if transposed else None
    state = torch.empty(3, dtype=torch.float32, device=dev)
    _lib.check(_L.mantis_fp8_quantize(_p(x), rows, cols, x.stride(0), fmt, _p(q), cols, _p(qt), rp, _p(state), _p(ws), _p(amax),
                                      0 if amax is None else amax.numel(), _stream()), f"fp8_quantize {rows}x{cols}")
    return Fp8Tensor(q, qt, state, fmt, rows, cols)


def gemm_fp8_nt(a8, a_dequant, b8, b_dequant, fmt_a=FP8_E4M3, bias=None, residual=None, out=None, accumulate=False, k=None, variant=0,
                rowwise=False):
    """out[M,N] bf16 = epi(dequant_a * dequant_b * a8[M,K] . b8[N,K]^T); a8 / b8 uint8 row-major fp8 (b8 e4m3).
    rowwise=True: a_dequant fp32[M] and b_dequant fp32[N], out[m, n] = epi(a_dequant[m] * b_dequant[n] * ...)."""
    M, K = a8.shape
    N = b8.shape[0]
    K = K if k is None else k
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a8.device)
    if rowwise and (a_dequant.numel() != M or b_dequant.numel() != N):
        raise ValueError(f"gemm_fp8_nt rowwise: dequant vectors {a_dequant.numel()} / {b_dequant.numel()} for M={M} N={N}")
    if not rowwise and (a_dequant.numel() != 1 or b_dequant.numel() != 1):
        raise ValueError("gemm_fp8_nt: per-tensor dequant factors are fp32[1]; pass rowwise=True for vectors")
    flags = ((1 if bias is not None else 0) | (16 if residual is not None else 0) | (32 if accumulate else 0) | (128 if rowwise else 0) |
             (variant << 8))
    prof = _launch().timer
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _L.mantis_gemm_fp8_nt(_p(a8), a8.stride(0), _p(b8), b8.stride(0), _p(out), out.stride(0), M, N, K, _p(a_dequant), _p(b_dequant),
                               fmt_a, _p(bias), _p(residual), 0 if residual is None else residual.stride(0), flags, _stream())
    _lib.check(rc, f"gemm_fp8 M={M} N={N} K={K}")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append(("gemm_fp8_nt_kernel", 2.0 * M * N * K, 1.0 * (M * K + N * K) + 2.0 * M * N * (1 + (residual is not None) + bool(accumulate)),
                     e0, e1))
    return out


def gemm_fp8_dx_swiglu(dy8, dy_dequant, wt8, w_dequant, gu, fmt_a=FP8_E5M2):
    """(dgu[M, 2I], amax fp32[1]) with dgu = swiglu_bwd(dequant * dy8[M, d] . wt8[I, d]^T, gu[M, 2I]) in one launch (the [M, I] activation
    gradient never goes to HBM) and amax = max |dgu|, taken by the same epilogue for the quantiser that follows."""
    M, d = dy8.shape
    I = wt8.shape[0]
    if gu.shape != (M, 2 * I):
        raise ValueError(f"gemm_fp8_dx_swiglu: dy8 {tuple(dy8.shape)} wt8 {tuple(wt8.shape)} gu {tuple(gu.shape)}")
    dgu = torch.empty_like(gu)
    amax = torch.empty(1, dtype=torch.float32, device=gu.device)
    prof = _launch().timer
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _L.mantis_gemm_fp8_dx_swiglu(_p(dy8), dy8.stride(0), _p(wt8), wt8.stride(0), _p(dgu), dgu.stride(0), M, I, d, _p(dy_dequant),
                                      _p(w_dequant), fmt_a, _p(gu), gu.stride(0), _p(amax), _stream())
    _lib.check(rc, f"gemm_fp8+swiglu_bwd M={M} I={I} d={d}")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append(("gemm_fp8_nt_kernel", 2.0 * M * I * d, 1.0 * (M * d + I * d) + 2.0 * 4 * M * I, e0, e1))
    return dgu, amax


def transpose(x, rpad=None):
    """[R, C] -> [C, Rpad] (Rpad = R rounded up to 8, zero filled)."""
    _chk2d(x, "x")
    R, C = x.shape
    Rp = pad8(R) if rpad is None else rpad
    out = torch.empty((C, Rp), dtype=BF16, device=x.device)
    rc = _L.mantis_transpose(_p(x), _p(out), R, C, Rp, x.stride(0), Rp, 1, 1, 0, 0, 0, 0, _stream())
    _lib.check(rc, f"transpose {R}x{C}")
    return out


def transpose_heads(x, B, Lseq, nheads, hd, col0, Lp):
    """x: [B*L, ld] activation; heads at columns col0 + h*hd.  Returns [B, nheads, hd, Lp] (sequence contiguous)."""
    _chk2d(x, "x")
    out = torch.empty((B, nheads, hd, Lp), dtype=BF16, device=x.device)
    ld = x.stride(0)
    base = x.data_ptr() + col0 * 2
    rc = _L.mantis_transpose(base, _p(out), Lseq, hd, Lp, ld, Lp, B, nheads, Lseq * ld, hd, nheads * hd * Lp, hd * Lp,
                             _stream())
    _lib.check(rc, "transpose_heads")
    return out


def linear_fwd(x, w, bias=None, act=None, residual=None):
    return gemm_nt(x, w, bias=bias, act=act, residual=residual)


def _prefers_256(M, N, K):
    """True when the library's tile heuristic (csrc/gemm.hip, gemm_pick_variant) takes the 256x256 ring kernel (for the CU budget the
    launch will carry)."""
    return _L.mantis_gemm_pick_variant_cus(M, N, K, _launch().gemm_cus) == 12


def linear_dx(dy, w, k=None):
    """dx[M, in] = dy[M, out] @ w[out, in].  Shapes that tile well at 256x256 consume the weight K-major as stored (ring kernel
    with transposing fragment reads); badly quantised ones run the faster 128x128 NT kernel on a transposed weight copy."""
    if _prefers_256(dy.shape[0], w.shape[1], w.shape[0]):
        return gemm_nt(dy, w, b_kmajor=True, k=w.shape[0])
    wt = transpose(w)                       # [in, pad8(out)]
    return gemm_nt(dy, wt, k=wt.shape[1] if k is None else k)


import os as _os
FUSE_FWD_EPILOGUES = _os.environ.get("MANTIS_NO_FUSE") != "1"      # A/B switch (measurements / tests): off = always the two-launch forms


def _gemm_fused(a, b, out, mode, aux0, aux1, aux_ld, aux_n, bias, variant, extra_bytes):
    """mantis_gemm_bf16_nt_fused; returns False when the library declines the shape (the caller runs the unfused launches)."""
    M, K = a.shape
    N = b.shape[0]
    prof = _launch().timer
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    wsp, wsn = _gemm_workspace()
    rc = _L.mantis_gemm_bf16_nt_fused(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, _p(bias), mode, _p(aux0),
                                      _p(aux1), aux_ld, aux_n, variant | (_launch().gemm_cus << CUS_SHIFT) | (128 if _launch().shared_gpu else 0),
                                      wsp, wsn, _stream())
    if rc == -2:
        return False
    _lib.check(rc, f"gemm_fused mode={mode} M={M} N={N} K={K}")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append(("gemm_nt_kernel", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N) + extra_bytes, e0, e1,
                     (M, N, K, "NT", ("swiglu" if mode == 1 else "rope") + ("+bias" if bias is not None else "")), _stream()))
    return True


def linear_gu_swiglu(x, w_gu, variant=0, amax_parts=None):
    """(gu [M, 2I], a [M, I]) = (x . [gate | up]^T, silu(gate) * up): the SwiGLU activation runs in the epilogue of the projection (one
    launch, the 2I-wide result is not read back); shapes the fused kernel declines take gemm_nt + swiglu_fwd (same results)."""
    _chk2d(x, "x"), _chk2d(w_gu, "w_gu")
    M, I2 = x.shape[0], w_gu.shape[0]
    if FUSE_FWD_EPILOGUES and amax_parts is None and I2 % 256 == 0:
        gu = torch.empty((M, I2), dtype=BF16, device=x.device)
        a = torch.empty((M, I2 // 2), dtype=BF16, device=x.device)
        if _gemm_fused(x, w_gu, gu, 1, a, None, a.stride(0), 0, None, variant, 2.0 * M * (I2 // 2)):
            return gu, a
    gu = gemm_nt(x, w_gu)
    return gu, swiglu_fwd(gu, amax_parts=amax_parts)


def linear_qkv_rope(x, w_qkv, bias, cos, sin, n_rope_heads, hd, variant=0):
    """qkv [M, N] = x . w_qkv^T (+ bias) with the rotary embedding already applied to the first n_rope_heads heads (q and k): RoPE runs in
    the epilogue of the projection for head dim 128; other geometries take gemm_nt + rope_apply_ (same results)."""
    _chk2d(x, "x"), _chk2d(w_qkv, "w_qkv")
    M, N = x.shape[0], w_qkv.shape[0]
    if (FUSE_FWD_EPILOGUES and hd == 128 and N % 256 == 0 and cos.dtype == BF16 and cos.dim() == 2 and cos.shape == (M, 64)
            and cos.stride(1) == 1 and sin.shape == cos.shape and sin.stride() == cos.stride()):
        qkv = torch.empty((M, N), dtype=BF16, device=x.device)
        if _gemm_fused(x, w_qkv, qkv, 2, cos, sin, cos.stride(0), n_rope_heads * hd, bias, variant, 4.0 * M * 64):
            return qkv
    qkv = gemm_nt(x, w_qkv, bias=bias)
    return rope_apply_(qkv, cos, sin, n_rope_heads, hd)


def linear_dx_swiglu(dy, w_down, gu, variant=0, sk_inkernel=False):
    """dgu[M, 2I] = swiglu_bwd(dy[M, d] @ w_down[d, I], gu[M, 2I]) in one launch: the SwiGLU backward runs in the GEMM epilogue
    (the [M, I] activation gradient never goes to HBM)."""
    _chk2d(dy, "dy"), _chk2d(w_down, "w_down"), _chk2d(gu, "gu")
    M, d = dy.shape
    I = w_down.shape[1]
    if gu.shape != (M, 2 * I) or w_down.shape[0] != d:
        raise ValueError(f"linear_dx_swiglu: dy {tuple(dy.shape)} w_down {tuple(w_down.shape)} gu {tuple(gu.shape)}")
    dgu = torch.empty_like(gu)
    prof = _launch().timer
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    wsp, wsn = _gemm_workspace()
    rc = _L.mantis_gemm_bf16_nt(_p(dy), dy.stride(0), _p(w_down), w_down.stride(0), _p(dgu), dgu.stride(0), M, I, d, None, _p(gu),
                                gu.stride(0), 64 | 8192 | (variant << 8) | (_launch().gemm_cus << CUS_SHIFT) | (SK_INKERNEL if sk_inkernel else 0)
                                | (SHARED_GPU if _launch().shared_gpu else 0), wsp, wsn,
                                _stream())
    _lib.check(rc, f"gemm+swiglu_bwd M={M} I={I} d={d}")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append(("gemm_nt_kernel", 2.0 * M * I * d, 2.0 * (M * d + I * d + 4 * M * I), e0, e1, (M, I, d, "NN", "swiglu_bwd"), _stream()))
    return dgu


def linear_dw(dy, x, grad_w, accumulate):
    """grad_w[out, in] (+)= dy[M, out]^T @ x[M, in]: both activations are consumed K-major as stored.
    With a collector in the LaunchContext (`dw_sumsq` = optim.FusedAdamW.begin_fold(), the backward of an accumulation boundary on a single
    rank) the GEMM also leaves the sum of squares of what it stored (mantis_gemm_bf16_nt_sumsq), and the optimizer's clip_grad_norm_ sums
    ~1e5 tile values instead of re-reading 16 GB of gradients."""
    ctx = _launch()
    col = ctx.dw_sumsq
    a = dy[:, : grad_w.shape[0]]
    if col is not None and grad_w.is_contiguous():
        Kd, M = a.shape
        N = x.shape[1]
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        slot = col.take(grad_w, tiles)
        if slot is not None:
            prof = ctx.timer
            if prof is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            wsp, wsn = _gemm_workspace()
            rc = _L.mantis_gemm_bf16_nt_sumsq(_p(a), a.stride(0), _p(x), x.stride(0), _p(grad_w), grad_w.stride(0), M, N, Kd,
                                              (32 if accumulate else 0) | 4096 | 8192 | (ctx.gemm_cus << CUS_SHIFT), slot, wsp, wsn, _stream())
            if rc == 0:
                if prof is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    prof.append(("gemm_nt_kernel", 2.0 * M * N * Kd, 2.0 * (M * Kd + N * Kd + M * N * (1 + bool(accumulate))), e0, e1,
                                 (M, N, Kd, "TN", "sumsq" + ("+acc" if accumulate else "")), _stream()))
                return
            if rc != -2:
                _lib.check(rc, f"gemm_sumsq M={M} N={N} K={Kd}")
            col.give_back(grad_w, tiles)             # shape outside the fused kernel's conditions: plain GEMM, norm by the separate pass
    gemm_nt(a, x, out=grad_w, accumulate=accumulate, a_kmajor=True, b_kmajor=True)


def dw_pair_wins(out1, in1, out2, in2, tokens):
    """True when linear_dw_pair(...) would run the two weight-gradient GEMMs grad[out1, in1] and grad[out2, in2] over `tokens` rows as ONE grid
    and the library's cost model predicts a gain over two launches for this context's CU budget (csrc/gemm.hip: tn_pair_wins)."""
    return bool(_L.mantis_gemm_tn_pair_wins(int(out1), int(in1), int(out2), int(in2), int(tokens), _launch().gemm_cus))


def linear_dw_pair(dy1, x1, grad_w1, dy2, x2, grad_w2, accumulate):
    """grad_w1 (+)= dy1^T @ x1 and grad_w2 (+)= dy2^T @ x2 in ONE launch (mantis_gemm_bf16_tn_pair: the two grids' tiles in one whole-round
    grid -- dW(down_proj) + dW(q|k|v) of a decoder layer are 896 + 384 = 1280 tiles = 5.0 rounds on 256 CUs instead of two launches with a
    K-split remainder round and a finishing pass each).  With a sum-of-squares collector in the LaunchContext both results leave their tile
    partials, as linear_dw does.  Falls back to two linear_dw calls for shapes outside the paired kernel's conditions."""
    ctx = _launch()
    a1, a2 = dy1[:, : grad_w1.shape[0]], dy2[:, : grad_w2.shape[0]]
    Kd = a1.shape[0]
    ok = (a2.shape[0] == Kd and x1.shape[0] == Kd and x2.shape[0] == Kd and grad_w1.is_contiguous() and grad_w2.is_contiguous()
          and grad_w1.shape[1] % 256 == 0 and grad_w2.shape[1] % 256 == 0)
    if not ok:
        linear_dw(dy1, x1, grad_w1, accumulate)
        linear_dw(dy2, x2, grad_w2, accumulate)
        return
    (M1, N1), (M2, N2) = grad_w1.shape, grad_w2.shape
    col = ctx.dw_sumsq
    s1 = s2 = None
    if col is not None:
        t1, t2 = ((M1 + 255) // 256) * (N1 // 256), ((M2 + 255) // 256) * (N2 // 256)
        s1 = col.take(grad_w1, t1)
        s2 = col.take(grad_w2, t2) if s1 is not None else None
        if s1 is not None and s2 is None:
            col.give_back(grad_w1, t1)
            s1 = None
    prof = ctx.timer
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _L.mantis_gemm_bf16_tn_pair(_p(a1), a1.stride(0), _p(x1), x1.stride(0), _p(grad_w1), grad_w1.stride(0), M1, N1, s1,
                                     _p(a2), a2.stride(0), _p(x2), x2.stride(0), _p(grad_w2), grad_w2.stride(0), M2, N2, s2,
                                     Kd, (32 if accumulate else 0) | (ctx.gemm_cus << CUS_SHIFT), _stream())
    if rc == -2:
        if s1 is not None:
            col.give_back(grad_w2, ((M2 + 255) // 256) * (N2 // 256))
            col.give_back(grad_w1, ((M1 + 255) // 256) * (N1 // 256))
        linear_dw(dy1, x1, grad_w1, accumulate)
        linear_dw(dy2, x2, grad_w2, accumulate)
        return
    _lib.check(rc, f"gemm_tn_pair {M1}x{N1} + {M2}x{N2} K={Kd}")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        fl = 2.0 * Kd * (M1 * N1 + M2 * N2)
        by = 2.0 * (Kd * (M1 + N1 + M2 + N2) + (M1 * N1 + M2 * N2) * (1 + bool(accumulate)))
        # one table row for the launch: the first problem's shape, the second named in the epilogue column (FLOPs and bytes are the pair's)
        prof.append(("gemm_nt_kernel", fl, by, e0, e1, (M1, N1, Kd, "TN", f"paired with {M2}x{N2}x{Kd}" + ("+sumsq" if s1 is not None else "")
                                                        + ("+acc" if accumulate else "")), _stream()))


def sum_f32(x, out, accumulate=False):
    _lib.check(_L.mantis_sum_f32(_p(x), x.numel(), _p(out), int(accumulate), _stream()), "sum_f32")


def colsum(x, grad, accumulate):
    """grad[N] (+)= sum_m x[m, n]  (bias gradient)."""
    _chk2d(x, "x")
    M, N = x.shape
    P = _L.mantis_colsum_partials(M)
    ws = torch.empty((P, N), dtype=torch.float32, device=x.device)
    _lib.check(_L.mantis_colsum(_p(x), _p(grad), int(accumulate), _p(ws), M, N, x.stride(0), _stream()), "colsum")


# ----------------------------------------------------------------------------------------------------------- norms / acts
def amax_parts_buffer(device):
    """Scratch for the producer-side amax protocol (rmsnorm_fwd / rmsnorm_bwd / swiglu_fwd -> fp8_quantize(amax=...))."""
    return torch.empty(_L.mantis_fp8_quantize_ws_floats(), dtype=torch.float32, device=device)


def rmsnorm_fwd(x, w, eps, want_rstd=True, amax_parts=None):
    """amax_parts: buffer from amax_parts_buffer(); receives the per-workgroup maxima of |y| for the fp8 quantiser that follows."""
    _chk2d(x, "x")
    rows, d = x.shape
    y = torch.empty_like(x)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device) if want_rstd else None
    _lib.check(_L.mantis_rmsnorm_fwd(_p(x), _p(w), _p(y), _p(rstd), rows, d, float(eps), _p(amax_parts), _stream()), "rmsnorm_fwd")
    return y, rstd


def rmsnorm_bwd(dy, x, w, rstd, dres, grad_w, accumulate, amax_parts=None):
    rows, d = x.shape
    dx = torch.empty_like(x)
    ws = None
    if grad_w is not None:
        ws = torch.empty((_L.mantis_rmsnorm_bwd_partials(rows), d), dtype=torch.float32, device=x.device)
    _lib.check(_L.mantis_rmsnorm_bwd(_p(dy), _p(x), _p(w), _p(rstd), _p(dres), _p(dx), _p(grad_w), int(accumulate), _p(ws),
                                     rows, d, _p(amax_parts), _stream()), "rmsnorm_bwd")
    return dx


def layernorm_fwd(x, w, b, eps):
    _chk2d(x, "x")
    y = torch.empty_like(x)
    _lib.check(_L.mantis_layernorm_fwd(_p(x), _p(w), _p(b), _p(y), x.shape[0], x.shape[1], float(eps), _stream()), "layernorm")
    return y


def swiglu_fwd(gu, amax_parts=None):
    M, I2 = gu.shape
    out = torch.empty((M, I2 // 2), dtype=BF16, device=gu.device)
    _lib.check(_L.mantis_swiglu_fwd(_p(gu), _p(out), M, I2 // 2, gu.stride(0), _p(amax_parts), _stream()), "swiglu_fwd")
    return out


def swiglu_bwd(dact, gu):
    M, I2 = gu.shape
    dgu = torch.empty_like(gu)
    _lib.check(_L.mantis_swiglu_bwd(_p(dact), _p(gu), _p(dgu), M, I2 // 2, gu.stride(0), _stream()), "swiglu_bwd")
    return dgu


def act_fwd(x, kind):
    y = torch.empty_like(x)
    _lib.check(_L.mantis_act_fwd(_p(x), _p(y), x.numel(), ACT_KIND[kind], _stream()), "act_fwd")
    return y


def act_bwd(dy, x, kind):
    dx = torch.empty_like(x)
    _lib.check(_L.mantis_act_bwd(_p(dy), _p(x), _p(dx), x.numel(), ACT_KIND[kind], _stream()), "act_bwd")
    return dx


def add(a, b):
    y = torch.empty_like(a)
    _lib.check(_L.mantis_add(_p(a), _p(b), _p(y), a.numel(), _stream()), "add")
    return y


# ----------------------------------------------------------------------------------------------------------- rope / attention
def rope_table(position_ids, inv_freq):
    """position_ids int64 [R]; inv_freq fp32 [hd/2] (computed on the host exactly as the reference does)."""
    R, half = position_ids.numel(), inv_freq.numel()
    cos = torch.empty((R, half), dtype=BF16, device=position_ids.device)
    sin = torch.empty_like(cos)
    _lib.check(_L.mantis_rope_table(_p(position_ids), _p(inv_freq), _p(cos), _p(sin), R, half, _stream()), "rope_table")
    return cos, sin


def rope_table_sections(position_ids, inv_freq, section_of_freq):
    """position_ids int64 [S, R]; frequency j of the table takes its position from row section_of_freq[j] (int32 [hd/2])."""
    S, R = position_ids.shape
    half = inv_freq.numel()
    cos = torch.empty((R, half), dtype=BF16, device=position_ids.device)
    sin = torch.empty_like(cos)
    _lib.check(_L.mantis_rope_table_sections(_p(position_ids), _p(inv_freq), _p(section_of_freq), _p(cos), _p(sin), R, half, _stream()),
               "rope_table_sections")
    return cos, sin


def rope_apply_(x, cos, sin, nheads, hd, backward=False):
    _chk2d(x, "x")
    _lib.check(_L.mantis_rope_apply(_p(x), _p(cos), _p(sin), x.shape[0], nheads, hd, x.stride(0), int(backward), _stream()),
               "rope_apply")
    return x


def attn_fwd_qkv(q, k, v, B, Lseq, H, Hkv, hd, kmask, scale, causal, want_lse=True, kstart=None, out=None):
    """q [B*L, >=H*hd], k / v [B*L, >=Hkv*hd] bf16 row views (any row stride, heads contiguous inside a row; they may be column
    slices of one fused projection output).  Returns o [B*L, H*hd], lse [B,H,L] fp32."""
    _chk2d(q, "q"), _chk2d(k, "k"), _chk2d(v, "v")
    o = torch.empty((B * Lseq, H * hd), dtype=BF16, device=q.device) if out is None else out     # `out`: a row view of a larger buffer
    lse = torch.empty((B, H, Lseq), dtype=torch.float32, device=q.device) if want_lse else None
    rc = _L.mantis_attn_fwd(_p(q), _p(k), _p(v), _p(kmask), _p(kstart), _p(o), _p(lse), B, Lseq, H, Hkv, hd, q.stride(0), k.stride(0),
                            v.stride(0), o.stride(0), float(scale), int(causal), _stream())
    _lib.check(rc, f"attn_fwd hd={hd}")
    return o, lse


def attn_bwd_qkv(q, k, v, o, do, lse, dq, dk, dv, B, Lseq, H, Hkv, hd, kmask, scale, causal, kstart=None, qend=None, no_workspace=False):
    """Gradients w.r.t. q, k, v written into the given row views dq [B*L, H*hd], dk / dv [B*L, Hkv*hd] (any row stride).
    no_workspace (tests): withhold the per-query-head workspace, which holds GQA geometries to the group dK/dV kernel where it exists."""
    dsum = torch.empty((B, H, Lseq), dtype=torch.float32, device=q.device)     # rowsum(dO * O): filled by the dQ kernel
    ws = torch.empty((2, B * Lseq, H * hd), dtype=BF16, device=q.device) if (_L.mantis_attn_bwd_needs_workspace(H, Hkv, hd) and
                                                                              not no_workspace) else None
    rc = _L.mantis_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(do), _p(kmask), _p(kstart), _p(qend), _p(lse), _p(dsum), _p(dq), _p(dk),
                            _p(dv), _p(ws), B, Lseq,
                            H, Hkv, hd, q.stride(0), k.stride(0), v.stride(0), o.stride(0), do.stride(0), dq.stride(0), dk.stride(0),
                            dv.stride(0), float(scale), int(causal), _stream())
    _lib.check(rc, f"attn_bwd hd={hd}")


def attn_fwd_cross(q, k, v, B, Lq, Lk, H, Hkv, hd, kmask, scale):
    """Non-causal cross attention: q [B*Lq, >=H*hd] over k / v [B*Lk, >=Hkv*hd] row views (any row stride), kmask int32 [B, Lk] or None.
    Returns o [B*Lq, H*hd], lse fp32 [B, H, Lq]."""
    _chk2d(q, "q"), _chk2d(k, "k"), _chk2d(v, "v")
    o = torch.empty((B * Lq, H * hd), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device)
    rc = _L.mantis_attn_fwd_cross(_p(q), _p(k), _p(v), _p(kmask), _p(o), _p(lse), B, Lq, Lk, H, Hkv, hd, q.stride(0), k.stride(0),
                                  v.stride(0), o.stride(0), float(scale), _stream())
    _lib.check(rc, f"attn_fwd_cross hd={hd}")
    return o, lse


def attn_bwd_cross(q, k, v, o, do, lse, dq, dk, dv, B, Lq, Lk, H, Hkv, hd, kmask, scale):
    """Backward of attn_fwd_cross into the given row views dq [B*Lq, H*hd], dk / dv [B*Lk, Hkv*hd] (any row stride)."""
    dsum = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device)
    ws = torch.empty((2, B * Lk, H * hd), dtype=BF16, device=q.device) if H != Hkv else None
    rc = _L.mantis_attn_bwd_cross(_p(q), _p(k), _p(v), _p(o), _p(do), _p(kmask), _p(lse), _p(dsum), _p(dq), _p(dk), _p(dv), _p(ws), B,
                                  Lq, Lk, H, Hkv, hd, q.stride(0), k.stride(0), v.stride(0), o.stride(0), do.stride(0), dq.stride(0),
                                  dk.stride(0), dv.stride(0), float(scale), _stream())
    _lib.check(rc, f"attn_bwd_cross hd={hd}")


def _split_qkv(qkv, H, Hkv, hd):
    return qkv[:, : H * hd], qkv[:, H * hd: (H + Hkv) * hd], qkv[:, (H + Hkv) * hd: (H + 2 * Hkv) * hd]


def attn_fwd(qkv, B, Lseq, H, Hkv, hd, kmask, scale, causal, want_lse=True, kstart=None):
    """qkv: [B*L, (H+2Hkv)*hd] fused projection output (q | k | v).  Returns o [B*L, H*hd], lse [B,H,L].
    kstart int32 [B,L] (packed samples): first key position each query may attend."""
    _chk2d(qkv, "qkv")
    q, k, v = _split_qkv(qkv, H, Hkv, hd)
    return attn_fwd_qkv(q, k, v, B, Lseq, H, Hkv, hd, kmask, scale, causal, want_lse, kstart=kstart)


def attn_bwd(qkv, o, do, lse, B, Lseq, H, Hkv, hd, kmask, scale, causal, kstart=None, qend=None, no_workspace=False):
    """Returns dqkv [B*L, (H+2Hkv)*hd] (gradient w.r.t. the post-RoPE q, k and v)."""
    dqkv = torch.empty_like(qkv)
    q, k, v = _split_qkv(qkv, H, Hkv, hd)
    dq, dk, dv = _split_qkv(dqkv, H, Hkv, hd)
    attn_bwd_qkv(q, k, v, o, do, lse, dq, dk, dv, B, Lseq, H, Hkv, hd, kmask, scale, causal, kstart=kstart, qend=qend,
                 no_workspace=no_workspace)
    return dqkv


# ----------------------------------------------------------------------------------------------------------- packing / loss
class PackPlan:
    __slots__ = ("B", "T", "L", "N", "I", "src", "attention_mask", "labels", "position_ids", "kmask", "text_pos", "img_slot",
                 "ce_row", "ce_tgt", "status", "kstart", "qend")


def pack_plan(input_ids, attention_mask, labels, num_patches, num_images, image_token_index, pad_token_id, ignore_index, L,
              fix_unequal_counts=False):
    """The integer plan of _merge_input_ids_with_image_features.  fix_unequal_counts=False: the reference's placement bit for bit
    (including its quirk for right-padded batches with unequal image counts); True: the index-only placement of SURVEY appendix A
    (mantis_pack_plan_mode, mode 1)."""
    B, T = input_ids.shape
    dev = input_ids.device
    pl = PackPlan()
    pl.B, pl.T, pl.L, pl.N, pl.I = B, T, L, num_patches, num_images
    pl.kstart = pl.qend = None
    pl.src = torch.empty((B, L), dtype=torch.int32, device=dev)
    pl.attention_mask = torch.empty((B, L), dtype=torch.int64, device=dev)
    pl.labels = torch.empty((B, L), dtype=torch.int64, device=dev)
    pl.position_ids = torch.empty((B, L), dtype=torch.int64, device=dev)
    pl.kmask = torch.empty((B, L), dtype=torch.int32, device=dev)
    pl.text_pos = torch.empty((B, T), dtype=torch.int32, device=dev)
    pl.img_slot = torch.full((max(1, num_images * num_patches),), -1, dtype=torch.int32, device=dev)
    pl.ce_row = torch.empty((B * T,), dtype=torch.int32, device=dev)
    pl.ce_tgt = torch.empty((B * T,), dtype=torch.int32, device=dev)
    pl.status = torch.zeros((4,), dtype=torch.int32, device=dev)
    rc = _L.mantis_pack_plan_mode(_p(input_ids), _p(attention_mask), _p(labels), B, T, num_patches, num_images, image_token_index,
                                  pad_token_id, ignore_index, L, 1 if fix_unequal_counts else 0, _p(pl.src), _p(pl.attention_mask),
                                  _p(pl.labels), _p(pl.position_ids), _p(pl.kmask), _p(pl.text_pos), _p(pl.img_slot), _p(pl.ce_row),
                                  _p(pl.ce_tgt), _p(pl.status), _stream())
    _lib.check(rc, "pack_plan")
    return pl


def pack_segments(plan, input_ids, segment_ids, image_token_index):
    """Packed samples: adds plan.kstart / plan.qend (int32 [B,L]) and rewrites plan.position_ids / ce_row / ce_tgt in place."""
    dev = input_ids.device
    plan.kstart = torch.empty((plan.B, plan.L), dtype=torch.int32, device=dev)
    plan.qend = torch.empty((plan.B, plan.L), dtype=torch.int32, device=dev)
    ws = torch.empty((plan.B, plan.L), dtype=torch.int32, device=dev)
    rc = _L.mantis_pack_segments(_p(input_ids), _p(segment_ids), _p(plan.attention_mask), plan.B, plan.T, plan.N, image_token_index,
                                 plan.L, _p(plan.position_ids), _p(plan.ce_row), _p(plan.ce_tgt), _p(plan.kstart), _p(plan.qend),
                                 _p(ws), _stream())
    _lib.check(rc, "pack_segments")
    return plan


def pack_rows_fwd(plan, input_ids, embed_weight, image_features):
    d = embed_weight.shape[1]
    out = torch.empty((plan.B * plan.L, d), dtype=BF16, device=embed_weight.device)
    rc = _L.mantis_pack_rows_fwd(_p(plan.src), _p(input_ids), _p(embed_weight), _p(image_features), _p(out), plan.B, plan.T,
                                 plan.L, d, embed_weight.shape[0], _stream())
    _lib.check(rc, "pack_rows_fwd")
    return out


def gather_rows(x, idx):
    out = torch.empty((idx.numel(), x.shape[1]), dtype=BF16, device=x.device)
    _lib.check(_L.mantis_gather_rows(_p(x), _p(idx), _p(out), idx.numel(), x.shape[1], _stream()), "gather_rows")
    return out


def scatter_rows(x, idx, nrows_out, out=None):
    """out[idx[r]] = x[r] (unique idx; idx < 0 skipped).  `out`: an existing contiguous [nrows_out, d] buffer to scatter into."""
    if out is None:
        out = torch.zeros((nrows_out, x.shape[1]), dtype=BF16, device=x.device)
    _lib.check(_L.mantis_scatter_rows(_p(x), _p(idx), _p(out), idx.numel(), x.shape[1], _stream()), "scatter_rows")
    return out


def embed_grad(dmerged, input_ids, plan, grad_weight, accumulate):
    n = plan.B * plan.T
    ws = torch.empty((2, n), dtype=torch.int32, device=dmerged.device)
    rc = _L.mantis_embed_grad(_p(dmerged), _p(input_ids), _p(plan.text_pos), ws[0].data_ptr(), ws[1].data_ptr(), _p(grad_weight),
                              plan.B, plan.T, plan.L, dmerged.shape[1], grad_weight.shape[0], int(accumulate), _stream())
    _lib.check(rc, "embed_grad")


def ce_fwd_bwd(logits, targets, V, grad_scale, loss_scale, write_grad=True):
    """logits [R, ld>=pad8(V)] bf16 -> overwritten with dlogits; returns (loss[1] fp32, count[2] int32 = [valid rows,
    rows with an out-of-range target])"""
    R = logits.shape[0]
    ws = torch.empty((R,), dtype=torch.float32, device=logits.device)
    count = torch.empty((2,), dtype=torch.int32, device=logits.device)
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    rc = _L.mantis_ce_fwd_bwd(_p(logits), _p(targets), R, V, logits.stride(0), float(grad_scale), float(loss_scale),
                              int(write_grad), _p(ws), None, _p(count), _p(loss), _stream())
    _lib.check(rc, "ce_fwd_bwd")
    return loss, count


# ----------------------------------------------------------------------------------------------------------- ViT front end
def im2col(pixels, patch, kp):
    I, C, H, W = pixels.shape
    n = (H // patch) * (W // patch)
    out = torch.empty((I * n, kp), dtype=BF16, device=pixels.device)
    _lib.check(_L.mantis_im2col(_p(pixels), _p(out), I, C, H, W, patch, kp, _stream()), "im2col")
    return out


def cast_pad_rows(x, kp):
    """fp32 [R, K] -> bf16 [R, kp], zero tail."""
    R, Kk = x.shape
    out = torch.empty((R, kp), dtype=BF16, device=x.device)
    _lib.check(_L.mantis_cast_pad_rows(_p(x), _p(out), R, Kk, x.stride(0), kp, _stream()), "cast_pad_rows")
    return out


def vit_assemble(patch_out, pos_emb, cls_emb, I, N):
    d = patch_out.shape[1]
    nt = N + (1 if cls_emb is not None else 0)
    out = torch.empty((I * nt, d), dtype=BF16, device=patch_out.device)
    _lib.check(_L.mantis_vit_assemble(_p(patch_out), _p(pos_emb), _p(cls_emb), _p(out), I, N, d, _stream()), "vit_assemble")
    return out


def navit_prepare(pixels, pixel_mask, patch, side, bucket):
    """pixels fp32 [n, C, H, W] on the device, pixel_mask uint8/bool [n, H, W] or None, bucket int32 [tab_n, tab_n] (device).
    -> (real int32 [n], patch_mask int32 [n, ph*pw], pos_ids int32 [n, ph*pw], status int32 [n]), all on the device."""
    n, C, H, W = pixels.shape
    npatch = (H // patch) * (W // patch)
    dev = pixels.device
    real = torch.empty((n,), dtype=torch.int32, device=dev)
    pm = torch.empty((n, npatch), dtype=torch.int32, device=dev)
    pos = torch.empty((n, npatch), dtype=torch.int32, device=dev)
    status = torch.empty((n,), dtype=torch.int32, device=dev)
    if pixel_mask is not None:
        pixel_mask = pixel_mask.to(torch.uint8).contiguous()
    _lib.check(_L.mantis_navit_prepare(_p(pixels), _p(pixel_mask), n, C, H, W, patch, side, _p(bucket), bucket.shape[0], _p(real), _p(pm),
                                       _p(pos), _p(status), _stream()), "navit_prepare")
    return real, pm, pos, status


def drop_cls(x, I, N):
    out = torch.empty((I * N, x.shape[1]), dtype=BF16, device=x.device)
    _lib.check(_L.mantis_drop_cls(_p(x), _p(out), I, N, x.shape[1], _stream()), "drop_cls")
    return out


# ----------------------------------------------------------------------------------------------------------- optimizer
def adamw_flat(param, grad, master, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=None):
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    rc = _L.mantis_adamw(_p(param), _p(grad), _p(master), _p(m), _p(v), param.numel(), lr, beta1, beta2, eps, wd, bc1, bc2,
                         _p(grad_scale), _stream())
    _lib.check(rc, "adamw")


def adamw_split_flat(param, grad, master_lo, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=None):
    """AdamW with the fp32 master stored as (bf16 parameter, low 16 bits `master_lo` int16, tie bit in the sign of `v`): 26 B / parameter."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    rc = _L.mantis_adamw_split(_p(param), _p(grad), _p(master_lo), _p(m), _p(v), param.numel(), lr, beta1, beta2, eps, wd, bc1, bc2,
                               _p(grad_scale), _stream())
    _lib.check(rc, "adamw_split")


def master_join(param, master_lo, v, out=None):
    """fp32 masters of a split-master range (a new tensor unless `out`)."""
    if out is None:
        out = torch.empty(param.numel(), dtype=torch.float32, device=param.device)
    _lib.check(_L.mantis_master_join(_p(param), _p(master_lo), _p(v), _p(out), param.numel(), _stream()), "master_join")
    return out


def master_split(master, param, master_lo, v):
    """param <- bf16(master), master_lo <- its low 16 bits, the tie bit into the sign of v (|v| kept)."""
    _lib.check(_L.mantis_master_split(_p(master), _p(param), _p(master_lo), _p(v), master.numel(), _stream()), "master_split")


def grad_sumsq(x, out, accumulate=False, ws=None):
    if ws is None:
        ws = torch.empty((_L.mantis_sumsq_partials(x.numel()),), dtype=torch.float32, device=x.device)
    _lib.check(_L.mantis_sumsq(_p(x), x.numel(), _p(ws), _p(out), int(accumulate), _stream()), "sumsq")


def sumsq_ranges(x, off_len, partials):
    """partials[r] = sum of squares of x[off .. off + len) for the int64 (off, len) pairs of `off_len` (device tensor [n, 2])"""
    _lib.check(_L.mantis_sumsq_ranges(_p(x), _p(off_len), off_len.shape[0], _p(partials), _stream()), "sumsq_ranges")


def clip_scale(sumsq, max_norm):
    scale = torch.empty((1,), dtype=torch.float32, device=sumsq.device)
    norm = torch.empty((1,), dtype=torch.float32, device=sumsq.device)
    _lib.check(_L.mantis_clip_scale(_p(sumsq), float(max_norm), _p(scale), _p(norm), _stream()), "clip_scale")
    return scale, norm


def num_cus():
    return torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count


def cu_masked_stream(first_cu, n_cus):
    """torch stream whose kernels run on compute units [first_cu, first_cu + n_cus) only (hipExtStreamCreateWithCUMask)."""
    import ctypes
    h = ctypes.c_void_p()
    _lib.check(_L.mantis_stream_create_cu_mask(int(first_cu), int(n_cus), ctypes.byref(h)), "stream_create_cu_mask")
    return torch.cuda.ExternalStream(h.value, device=torch.device("cuda", torch.cuda.current_device()))


class SideStream:
    """A second stream for work off the critical path (the weight-gradient GEMMs of the backward): `run(fn, *inputs)` enqueues fn on it
    behind everything queued on the current stream so far; `join()` makes the current stream wait for it.  With its own hardware queue
    (GPU_MAX_HW_QUEUES, mantis_amd/__init__.py) its workgroups fill the compute units that the critical path's kernels leave idle in
    their incomplete last tile rounds."""

    def __init__(self, priority=None):
        """priority: None = a plain torch stream; "low" / "high" = a stream on the lowest- / highest-priority hardware queue the device offers
        (mantis_stream_create_priority): on "low" the side work only takes compute units the current stream's kernels leave idle."""
        if priority is None:
            self.stream = torch.cuda.Stream()
        else:
            self.stream = priority_stream({"low": 1, "high": -1}[priority])

    def run(self, fn, *inputs):
        ev = torch.cuda.Event()
        ev.record()
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            fn()
        for t in inputs:                       # the caching allocator must not hand these out again before the side stream is done
            if t is not None:
                t.record_stream(self.stream)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)


def priority_stream(level):
    """torch stream on a hardware queue of the highest (level < 0) / default (0) / lowest (level > 0) priority."""
    import ctypes
    h = ctypes.c_void_p()
    _lib.check(_L.mantis_stream_create_priority(int(level), ctypes.byref(h), None), "stream_create_priority")
    return torch.cuda.ExternalStream(h.value, device=torch.device("cuda", torch.cuda.current_device()))


_SIDE = {}


def side_stream(default="0"):
    """The process' side stream for the current device when MANTIS_DW_STREAM is 1 (a plain stream) or "low" (a stream on the
    lowest-priority hardware queue), else None (everything on the current stream).  `default`: the mode when the variable is unset (the
    fp8 layer loop asks for "low" on a single GPU, decoder_fp8.decoder_backward; MANTIS_DW_STREAM=0 switches it off)."""
    mode = _os.environ.get("MANTIS_DW_STREAM") or default
    if mode not in ("1", "low"):
        return None
    key = (torch.cuda.current_device(), mode)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = SideStream("low" if mode == "low" else None)
    return s


def synchronize():
    torch.cuda.synchronize()
