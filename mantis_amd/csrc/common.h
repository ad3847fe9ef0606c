// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libmantis_hip.so.
// wave = 64 lanes; bf16 is handled as raw 16-bit patterns so loads/stores vectorise to 16 B per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MANTIS_OK 0
#define MANTIS_EINVAL (-1)
#define MANTIS_EUNSUPPORTED (-2)
#define MANTIS_ELAUNCH (-3)

typedef unsigned short bf16_t;  // raw bits
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((unsigned int)x) << 16); }
__device__ __forceinline__ float bf2f_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf2f_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
// fp32 -> bf16, round-to-nearest-even (same rounding torch uses): the native __bf16 conversions lower to the hardware
// v_cvt_pk_bf16_f32 on gfx950 (one instruction per PAIR, no branches) instead of ~6 integer VALU ops + a NaN branch per element.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) {
    union { __bf16 b; bf16_t u; } c;
    c.b = (__bf16)f;
    return c.u;
}
__device__ __forceinline__ unsigned int pack_bf2(float lo, float hi) {
    union { bf16x2_t b; unsigned int u; } c;
    c.b = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return c.u;
}

// Wave-wide reductions on the VALU (DPP): xor 1 and xor 2 inside a quad, mirror inside 8 and inside 16 lanes -- every lane of a 16-lane
// row then holds the row's result -- and the four rows meet through v_readlane.  No LDS round trips and no lane-index registers (the
// six dependent ds_bpermute of the __shfl_xor form cost rmsnorm_bwd 4-5 us of 61 and seven spilled VGPRs, tools/rmsnorm_bwd_probe.hip).
// ALL 64 LANES MUST BE ACTIVE (full EXEC): v_readlane of lanes 0 / 16 / 32 / 48 reads whatever an inactive lane's register holds, so a
// partially active wave gets a wrong sum silently.  That includes block_sum / block_max below, which call these: every call site
// (norm, ce, optim, attn_dsum) is wave-uniform by construction; a caller inside a divergent branch must use __shfl_xor instead.
// -DMANTIS_DEBUG_EXEC makes a violated precondition trap.  The result is wave-uniform.
#ifdef MANTIS_DEBUG_EXEC
#define MANTIS_ASSERT_FULL_EXEC() do { if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap(); } while (0)
#else
#define MANTIS_ASSERT_FULL_EXEC() do { } while (0)
#endif
template <int CTRL>
__device__ __forceinline__ float mantis_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    MANTIS_ASSERT_FULL_EXEC();
    v += mantis_dpp<0xB1>(v);         // quad_perm [1,0,3,2]
    v += mantis_dpp<0x4E>(v);         // quad_perm [2,3,0,1]
    v += mantis_dpp<0x141>(v);        // row_half_mirror
    v += mantis_dpp<0x140>(v);        // row_mirror
    const int i = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
    MANTIS_ASSERT_FULL_EXEC();
    v = fmaxf(v, mantis_dpp<0xB1>(v));
    v = fmaxf(v, mantis_dpp<0x4E>(v));
    v = fmaxf(v, mantis_dpp<0x141>(v));
    v = fmaxf(v, mantis_dpp<0x140>(v));
    const int i = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// block-wide sum for blockDim.x <= 1024 (<=16 waves); `red` is >=16 floats of LDS; result broadcast to all threads.  Every thread of
// the block must call it with full waves (see wave_sum: blockDim.x % 64 == 0, no divergent caller).
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = (lane < nw) ? red[lane] : 0.f;
    t = wave_sum(t);
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = (lane < nw) ? red[lane] : -INFINITY;
    t = wave_max(t);
    return t;
}

// hipGetLastError() reports the last error of ANY runtime call of this thread, including the host process' own probing calls (PyTorch asks
// hipPointerGetAttributes about pageable host pointers, which fails by design and is not always cleared): a stale error would be
// reported as a failed launch of a kernel that ran.  So every launch clears the thread's error state first (MANTIS_LAUNCH) and
// mantis_check_launch() then sees only what the launches of this entry point produced.
#define MANTIS_LAUNCH(...)            \
    do {                              \
        (void)hipGetLastError();      \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

// fp8 quantiser protocol (csrc/gemm_fp8.hip): a kernel that PRODUCES a tensor which is quantised next can take the maximum |value| of what
// it writes on the way: every workgroup stores its maximum (bf16 bit pattern << 16, as a float) at amax_parts[blockIdx.x], workgroup 0
// clears the entries from gridDim.x to MANTIS_AMAX_PARTS.  The quantiser then skips its own pass over the tensor.
#define MANTIS_AMAX_PARTS 2048
#ifdef __HIPCC__
// umax: this thread's running maximum of (bits & 0x7fff) << 16 over the bf16 values it stored; 256- or 512-thread workgroups
__device__ __forceinline__ void mantis_store_amax_part(unsigned int umax, float* __restrict__ amax_parts) {
    __shared__ unsigned int s_amax[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int y = (unsigned int)__shfl_xor((int)umax, o);
        umax = umax > y ? umax : y;
    }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) s_amax[w] = umax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) umax = umax > s_amax[i] ? umax : s_amax[i];
        amax_parts[blockIdx.x] = __uint_as_float(umax);
    }
    if (blockIdx.x == 0)
        for (int i = gridDim.x + threadIdx.x; i < MANTIS_AMAX_PARTS; i += blockDim.x) amax_parts[i] = 0.f;
}
__device__ __forceinline__ unsigned int mantis_umax_bf2(unsigned int umax, unsigned int packed) {
    const unsigned int lo = (packed << 16) & 0x7fff0000u, hi = packed & 0x7fff0000u;
    umax = umax > lo ? umax : lo;
    return umax > hi ? umax : hi;
}
#endif

static inline int mantis_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MANTIS_OK : MANTIS_ELAUNCH;
}
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
