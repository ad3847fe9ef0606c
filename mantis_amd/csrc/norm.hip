// RMSNorm fwd/bwd (Llama) and LayerNorm fwd (frozen ViT) for gfx950.
//
// Replaces (reference path): HF LlamaRMSNorm.forward (transformers/models/llama/modeling_llama.py:60-65) and its autograd
// backward, reached from /root/reference/mantis/models/mllava/modeling_llava.py:510; nn.LayerNorm inside the SigLIP/CLIP
// encoder layers (transformers/models/siglip/modeling_siglip.py:325-358), reached from modeling_llava.py:456.
//
// HBM-bound: one 64-lane wave owns a row (16 B per lane per access, wave-shuffle reductions, no LDS, no barriers).
// Algorithmic bytes: fwd 2*rows*d*2 B; bwd reads dy,x and writes dx (3*rows*d*2 B) + a [partials,d] fp32 dW slab.
#include "common.h"

#define NORM_WAVES 4
#define NORM_MAXC 16  // chunks of 8 per lane -> d <= 64*8*16 = 8192

// y = w * bf16(x * rsqrt(mean(x^2) + eps))      (stats fp32, two roundings exactly as the reference does)
__global__ __launch_bounds__(64 * NORM_WAVES) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x,
                                                                      const bf16_t* __restrict__ w,
                                                                      bf16_t* __restrict__ y, float* __restrict__ rstd_out,
                                                                      long rows, int d, float eps, float* __restrict__ amax_parts) {
    const int lane = threadIdx.x & 63;
    unsigned int umax = 0;
    const long wave = (long)blockIdx.x * NORM_WAVES + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * NORM_WAVES;
    const int cpr = d >> 3;
    for (long r = wave; r < rows; r += nwaves) {
        const bf16_t* xr = x + r * d;
        float ss = 0.f;
        for (int c = lane; c < cpr; c += 64) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xr + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bf2f_lo(v[e]), b = bf2f_hi(v[e]);
                ss += a * a + b * b;
            }
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / (float)d + eps);
        if (lane == 0 && rstd_out) rstd_out[r] = rstd;
        for (int c = lane; c < cpr; c += 64) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xr + c * 8);
            const u32x4 g = *reinterpret_cast<const u32x4*>(w + c * 8);
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bf2f(f2bf(bf2f_lo(v[e]) * rstd)) * bf2f_lo(g[e]);
                const float b = bf2f(f2bf(bf2f_hi(v[e]) * rstd)) * bf2f_hi(g[e]);
                o[e] = pack_bf2(a, b);
                umax = mantis_umax_bf2(umax, o[e]);
            }
            *reinterpret_cast<u32x4*>(y + r * d + c * 8) = o;
        }
    }
    if (amax_parts) mantis_store_amax_part(umax, amax_parts);
}

// The same for d = NCH * 512 (4096: Llama-3 / Mistral, 3584: Qwen2-7B, ...) with the row held in registers: the generic kernel walks the
// row twice in loops of run-time length, one 16-B load per lane and iteration, each waited for before the next is issued -- 2 x d / 512
// memory round trips per row, 30 us for 5624 x 4096 (3 TB/s).  Here the NCH loads of a row are issued together, the sum of squares is
// taken in the same order (bit-identical), and the output is produced from the registers: one round trip per row.
template <int NCH>
__global__ __launch_bounds__(64 * NORM_WAVES) void rmsnorm_fwd_regs_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                           bf16_t* __restrict__ y, float* __restrict__ rstd_out,
                                                                           long rows, float eps, float* __restrict__ amax_parts) {
    constexpr int d = NCH * 512;
    const int lane = threadIdx.x & 63;
    unsigned int umax = 0;
    const long wave = (long)blockIdx.x * NORM_WAVES + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * NORM_WAVES;
    long r = wave;
    if (r < rows) {
        // first the row, then the weights (cache hits), one wait for both; a wave normally has ONE row
        u32x4 v[NCH], g[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) v[k] = *reinterpret_cast<const u32x4*>(x + r * d + (lane + 64 * k) * 8);
#pragma unroll
        for (int k = 0; k < NCH; ++k) g[k] = *reinterpret_cast<const u32x4*>(w + (lane + 64 * k) * 8);
        while (true) {
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = bf2f_lo(v[k][e]), b = bf2f_hi(v[k][e]);
                    ss += a * a + b * b;
                }
            ss = wave_sum(ss);
            const float rstd = rsqrtf(ss / (float)d + eps);
            if (lane == 0 && rstd_out) rstd_out[r] = rstd;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = bf2f(f2bf(bf2f_lo(v[k][e]) * rstd)) * bf2f_lo(g[k][e]);
                    const float b = bf2f(f2bf(bf2f_hi(v[k][e]) * rstd)) * bf2f_hi(g[k][e]);
                    o[e] = pack_bf2(a, b);
                    umax = mantis_umax_bf2(umax, o[e]);
                }
                *reinterpret_cast<u32x4*>(y + r * d + (lane + 64 * k) * 8) = o;
            }
            r += nwaves;
            if (r >= rows) break;
#pragma unroll
            for (int k = 0; k < NCH; ++k) v[k] = *reinterpret_cast<const u32x4*>(x + r * d + (lane + 64 * k) * 8);
        }
    }
    if (amax_parts) mantis_store_amax_part(umax, amax_parts);
}

// dx = rstd * (g - xhat * mean(g * xhat)) [+ dres],  g = dy * w, xhat = x * rstd;   dW partial[workgroup] += dy * xhat
// One wave per row, 8 waves per workgroup, 2 workgroups per CU (4 waves/SIMD: at 1 wave/SIMD the kernel was latency bound --
// 75 us for 5624 x 4096 = 2.5 TB/s); the 8 waves' column sums are folded through one LDS row in wave order (deterministic), so
// a workgroup emits ONE fp32 partial row for reduce_partials_kernel.
#define RMSB_WAVES 8
template <int MAXC>
__global__ __launch_bounds__(64 * RMSB_WAVES, MAXC <= 8 ? 4 : 2) void rmsnorm_bwd_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
    const float* __restrict__ rstd_in, const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
    float* __restrict__ dw_partial, long rows, int d, float* __restrict__ amax_parts) {
    unsigned int umax = 0;
    __shared__ float fold[MAXC * 64 * 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long wave = (long)blockIdx.x * RMSB_WAVES + wv;
    const long nwaves = (long)gridDim.x * RMSB_WAVES;
    const int cpr = d >> 3;
    float dwacc[MAXC][8];
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) dwacc[k][e] = 0.f;
    for (long r = wave; r < rows; r += nwaves) {
        const float rstd = rstd_in[r];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            const int c = lane + 64 * k;
            if (c < cpr) {
                const u32x4 vd = *reinterpret_cast<const u32x4*>(dy + r * d + c * 8);
                const u32x4 vx = *reinterpret_cast<const u32x4*>(x + r * d + c * 8);
                const u32x4 vw = *reinterpret_cast<const u32x4*>(w + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = bf2f_lo(vd[e]), d1 = bf2f_hi(vd[e]);
                    const float x0 = bf2f_lo(vx[e]) * rstd, x1 = bf2f_hi(vx[e]) * rstd;
                    dot += d0 * bf2f_lo(vw[e]) * x0 + d1 * bf2f_hi(vw[e]) * x1;
                    dwacc[k][2 * e] += d0 * x0;
                    dwacc[k][2 * e + 1] += d1 * x1;
                }
            }
        }
        dot = wave_sum(dot) / (float)d;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            const int c = lane + 64 * k;
            if (c < cpr) {
                const u32x4 vd = *reinterpret_cast<const u32x4*>(dy + r * d + c * 8);
                const u32x4 vx = *reinterpret_cast<const u32x4*>(x + r * d + c * 8);
                const u32x4 vw = *reinterpret_cast<const u32x4*>(w + c * 8);
                u32x4 vr = {0u, 0u, 0u, 0u};
                if (dres) vr = *reinterpret_cast<const u32x4*>(dres + r * d + c * 8);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = bf2f_lo(vx[e]) * rstd, x1 = bf2f_hi(vx[e]) * rstd;
                    const float a = rstd * (bf2f_lo(vd[e]) * bf2f_lo(vw[e]) - x0 * dot) + bf2f_lo(vr[e]);
                    const float b = rstd * (bf2f_hi(vd[e]) * bf2f_hi(vw[e]) - x1 * dot) + bf2f_hi(vr[e]);
                    o[e] = pack_bf2(a, b);
                    umax = mantis_umax_bf2(umax, o[e]);
                }
                *reinterpret_cast<u32x4*>(dx + r * d + c * 8) = o;
            }
        }
    }
    if (dw_partial) {
        // fold[k][e][lane]: lanes hit consecutive banks; waves add in index order
        for (int wq = 0; wq < RMSB_WAVES; ++wq) {
            if (wv == wq) {
#pragma unroll
                for (int k = 0; k < MAXC; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float* f = fold + (k * 8 + e) * 64 + lane;
                        *f = (wq == 0) ? dwacc[k][e] : *f + dwacc[k][e];
                    }
            }
            __syncthreads();
        }
        float* out = dw_partial + (long)blockIdx.x * d;
        for (int j = threadIdx.x; j < d; j += 64 * RMSB_WAVES) {
            const int c = j >> 3, e = j & 7;            // column j = chunk c (lane c & 63, k = c >> 6), element e
            out[j] = fold[((c >> 6) * 8 + e) * 64 + (c & 63)];
        }
    }
    if (amax_parts) mantis_store_amax_part(umax, amax_parts);
}

// ---- d = 4096 (Llama-3 / Mistral hidden size): the same arithmetic with a row split over FOUR waves.  Measured on 5624 x 4096
// (tools/rmsnorm_bwd_probe.hip, profiles/r02_rmsnorm_bwd_probe.log): 38-39 us against 60-66 us of the kernel above (4.7-4.8 vs 2.8-3.1
// TB/s).  What changed:
//   * 2 chunks of 8 per lane instead of 8: dy / x of a row stay in registers between the two passes (no second read), the NEXT row's
//     dy / x / rstd are requested before the current row is reduced, the current row's residual gradient at the top of the iteration
//     (it is needed only after the barrier) -- 126 VGPRs, no scratch, 4 waves / SIMD;
//   * buffer loads: descriptor over the whole tensor, the row's byte offset as the scalar offset, one 32-bit lane offset; rows past the
//     end load the last row with rstd = 0 (xhat = 0: nothing added to the dot or to dW);
//   * wave-wide sum by DPP (xor 1, xor 2 in a quad, mirrors inside 8 and 16 lanes, v_readlane across the four rows) instead of six
//     dependent ds_bpermute; the four partial dots of a row meet in a double-buffered LDS cell, one barrier per row pair.
// The dot is summed in another order than above (33-65 of 23 M bf16 values of dx round differently).
#define RMSQ_NQ 4        // waves per row
#define RMSQ_SLOTS 2     // rows per workgroup iteration
#define RMSQ_MAXC 2      // chunks per lane: d = 8 * 64 * RMSQ_NQ * RMSQ_MAXC
#define RMSQ_D (8 * 64 * RMSQ_NQ * RMSQ_MAXC)
typedef unsigned int rmsq_u32x4v __attribute__((vector_size(16)));

template <int CTRL>
__device__ __forceinline__ float rmsq_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {            // result is wave-uniform
    v = rmsq_dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]
    v = rmsq_dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
    v = rmsq_dpp_add<0x141>(v);       // row_half_mirror
    v = rmsq_dpp_add<0x140>(v);       // row_mirror
    const int i = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ u32x4 rmsq_ld(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const rmsq_u32x4v t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    u32x4 o;
    o[0] = t[0], o[1] = t[1], o[2] = t[2], o[3] = t[3];
    return o;
}
struct RmsqRow {
    u32x4 vd[RMSQ_MAXC], vx[RMSQ_MAXC];
    float rstd;
};
// Rows past the end: the loads are issued for the LAST row instead (the scalar offset is not part of the descriptor's range check) and
// rstd = 0 makes xhat = 0, so they add nothing to the dot or to dW; their dx is not stored.
__device__ __forceinline__ void rmsq_load(RmsqRow& t, __amdgpu_buffer_rsrc_t rsD, __amdgpu_buffer_rsrc_t rsX, const float* __restrict__ rstd_in,
                                          long r, long rows, unsigned lane_bytes) {
    const long rc = r < rows ? r : rows - 1;
    const unsigned soff = (unsigned)(rc * (RMSQ_D * 2));
#pragma unroll
    for (int k = 0; k < RMSQ_MAXC; ++k) {
        t.vd[k] = rmsq_ld(rsD, lane_bytes + 1024u * k, soff);
        t.vx[k] = rmsq_ld(rsX, lane_bytes + 1024u * k, soff);
    }
    t.rstd = r < rows ? rstd_in[r < rows ? r : 0] : 0.f;
}

template <bool AMAX>
__global__ __launch_bounds__(64 * RMSQ_NQ * RMSQ_SLOTS, 4) void rmsnorm_bwd_d4096_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
    const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, float* __restrict__ dw_partial, long rows, float* __restrict__ amax_parts) {
    constexpr int d = RMSQ_D;
    __shared__ float dotbuf[2][RMSQ_SLOTS][RMSQ_NQ];
    __shared__ float fold[d];                                  // one fp32 row
    unsigned int umax = 0;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: row index and row offsets live in SGPRs
    const int slot = wv / RMSQ_NQ, q = wv % RMSQ_NQ;
    const int c0 = q * (RMSQ_MAXC * 64) + lane;                // chunk index of k = 0; k-th chunk = c0 + 64 k
    u32x4 vw[RMSQ_MAXC];
#pragma unroll
    for (int k = 0; k < RMSQ_MAXC; ++k) vw[k] = *reinterpret_cast<const u32x4*>(w + (long)(c0 + 64 * k) * 8);
    float dwacc[RMSQ_MAXC][8];
#pragma unroll
    for (int k = 0; k < RMSQ_MAXC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) dwacc[k][e] = 0.f;
    const long stride = (long)gridDim.x * RMSQ_SLOTS;
    const int niter = (int)((rows + stride - 1) / stride);    // the same for every wave of the grid: the barriers stay uniform
    long r = (long)blockIdx.x * RMSQ_SLOTS + slot;
    const unsigned lane_bytes = (unsigned)c0 * 16u;
    const int nbytes = (int)(unsigned)(rows * d * 2);          // < 4 GiB: checked by the launcher
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)dres, 0, dres ? nbytes : 0, 0x00020000);   // none: zeros
    RmsqRow cur, nxt;
    rmsq_load(cur, rsD, rsX, rstd_in, r, rows, lane_bytes);
#pragma unroll 1
    for (int it = 0; it < niter; ++it) {
        rmsq_load(nxt, rsD, rsX, rstd_in, r + stride, rows, lane_bytes);
        const unsigned soff = (unsigned)((r < rows ? r : rows - 1) * (d * 2));
        u32x4 vr[RMSQ_MAXC];
#pragma unroll
        for (int k = 0; k < RMSQ_MAXC; ++k) vr[k] = rmsq_ld(rsR, lane_bytes + 1024u * k, soff);
        const float rstd = cur.rstd;
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < RMSQ_MAXC; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = bf2f_lo(cur.vd[k][e]), d1 = bf2f_hi(cur.vd[k][e]);
                const float x0 = bf2f_lo(cur.vx[k][e]) * rstd, x1 = bf2f_hi(cur.vx[k][e]) * rstd;
                dot += d0 * bf2f_lo(vw[k][e]) * x0 + d1 * bf2f_hi(vw[k][e]) * x1;
                dwacc[k][2 * e] += d0 * x0;
                dwacc[k][2 * e + 1] += d1 * x1;
            }
        dot = wave_sum_dpp(dot);
        if (lane == 0) dotbuf[it & 1][slot][q] = dot;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < RMSQ_NQ; ++j) tot += dotbuf[it & 1][slot][j];      // fixed order: deterministic
        tot /= (float)d;
        {
#pragma unroll
            for (int k = 0; k < RMSQ_MAXC; ++k) {
                rmsq_u32x4v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = bf2f_lo(cur.vx[k][e]) * rstd, x1 = bf2f_hi(cur.vx[k][e]) * rstd;
                    const float a = rstd * (bf2f_lo(cur.vd[k][e]) * bf2f_lo(vw[k][e]) - x0 * tot) + bf2f_lo(vr[k][e]);
                    const float b = rstd * (bf2f_hi(cur.vd[k][e]) * bf2f_hi(vw[k][e]) - x1 * tot) + bf2f_hi(vr[k][e]);
                    o[e] = pack_bf2(a, b);
                    if (AMAX) umax = mantis_umax_bf2(umax, o[e]);
                }
                // per-row store descriptor with an IMMEDIATE soffset: range-checked by the hardware (rows past the end: 0 records,
                // the store is dropped) and hazard-padded by the compiler.  NOT `rsO + SGPR soffset`: a >64-bit MUBUF store whose
                // soffset is an SGPR gets no padding against a following VALU write of its data registers (0.1-0.2 % garbage on
                // gfx950, tools/rmsnorm_bwd_probe.hip forms 2 / 3).
                const __amdgpu_buffer_rsrc_t rsRow = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(reinterpret_cast<char*>(dx) + (size_t)soff), 0, (r < rows) ? d * 2 : 0, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(o, rsRow, lane_bytes + 1024u * k, 0, 0);
            }
        }
        cur = nxt;
        r += stride;
    }
    if (dw_partial) {
        // the two row slots own the same columns: slot 0 writes, slot 1 adds (fixed order), then the row goes out coalesced
        for (int s = 0; s < RMSQ_SLOTS; ++s) {
            if (slot == s) {
#pragma unroll
                for (int k = 0; k < RMSQ_MAXC; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float* f = fold + ((q * RMSQ_MAXC + k) * 8 + e) * 64 + lane;
                        *f = (s == 0) ? dwacc[k][e] : *f + dwacc[k][e];
                    }
            }
            __syncthreads();
        }
        float* out = dw_partial + (long)blockIdx.x * d;
        for (int j = threadIdx.x; j < d; j += 64 * RMSQ_NQ * RMSQ_SLOTS) {
            const int c = j >> 3, e = j & 7;                   // column j = chunk c = (q * MAXC + k) * 64 + lane
            out[j] = fold[((c >> 6) * 8 + e) * 64 + (c & 63)];
        }
    }
    if (AMAX) mantis_store_amax_part(umax, amax_parts);
}


// grad[j] (+)= sum_p partial[p][j]   (fixed summation order -> deterministic): thread (column c, row group g) adds rows g, g + 16, ... in
// that order, then the 16 group sums are added in order.  16 columns x 16 row groups per workgroup: d / 16 workgroups instead of the
// d / 64 of round 3 (64 workgroups on 256 CUs: 10.7 us for 512 x 4096 partials), and the loads of eight rows are in flight before their
// adds -- same order of additions, bit-identical.
#define REDP_COLS 16
__global__ __launch_bounds__(REDP_COLS * 16) void reduce_partials_kernel(const float* __restrict__ partial, int P, int d,
                                                                         bf16_t* __restrict__ grad, int accumulate) {
    __shared__ float red[16][REDP_COLS];
    const int c = threadIdx.x % REDP_COLS, g = threadIdx.x / REDP_COLS;
    const int j = blockIdx.x * REDP_COLS + c;
    float s = 0.f;
    if (j < d) {
        int p = g;
        for (; p + 16 * 7 < P; p += 16 * 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(long)(p + 16 * k) * d + j];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; p < P; p += 16) s += partial[(long)p * d + j];
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && j < d) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][c];
        if (accumulate) t += bf2f(grad[j]);
        grad[j] = f2bf(t);
    }
}

// LayerNorm forward: y = bf16((x - mean) * rsqrt(var + eps) * w + b), stats fp32 (matches at::layer_norm on bf16 input)
__global__ __launch_bounds__(64 * NORM_WAVES) void layernorm_fwd_kernel(const bf16_t* __restrict__ x,
                                                                        const bf16_t* __restrict__ w,
                                                                        const bf16_t* __restrict__ bias,
                                                                        bf16_t* __restrict__ y, long rows, int d,
                                                                        float eps) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * NORM_WAVES + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * NORM_WAVES;
    const int cpr = d >> 3;
    for (long r = wave; r < rows; r += nwaves) {
        const bf16_t* xr = x + r * d;
        float s = 0.f;
        for (int c = lane; c < cpr; c += 64) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xr + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) s += bf2f_lo(v[e]) + bf2f_hi(v[e]);
        }
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
        for (int c = lane; c < cpr; c += 64) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xr + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bf2f_lo(v[e]) - mean, b = bf2f_hi(v[e]) - mean;
                q += a * a + b * b;
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
        for (int c = lane; c < cpr; c += 64) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xr + c * 8);
            const u32x4 g = *reinterpret_cast<const u32x4*>(w + c * 8);
            const u32x4 bb = *reinterpret_cast<const u32x4*>(bias + c * 8);
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = (bf2f_lo(v[e]) - mean) * rstd * bf2f_lo(g[e]) + bf2f_lo(bb[e]);
                const float b = (bf2f_hi(v[e]) - mean) * rstd * bf2f_hi(g[e]) + bf2f_hi(bb[e]);
                o[e] = pack_bf2(a, b);
            }
            *reinterpret_cast<u32x4*>(y + r * d + c * 8) = o;
        }
    }
}

static inline int norm_grid(long rows) {
    long g = (rows + NORM_WAVES - 1) / NORM_WAVES;
    return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}

extern "C" {

// amax_parts (nullable, here and in mantis_rmsnorm_bwd): MANTIS_AMAX_PARTS floats <- per-workgroup maxima of |y| (see common.h), for
// the fp8 quantiser that consumes y next
int mantis_rmsnorm_fwd(const void* x, const void* weight, void* y, float* rstd, int64_t rows, int d, float eps, float* amax_parts,
                       void* stream) {
    if (d % 8 || d <= 0) return MANTIS_EUNSUPPORTED;
    if (rows == 0) return MANTIS_OK;
    if (d % 512 == 0 && d / 512 >= 1 && d / 512 <= 16 && !((uintptr_t)x & 15) && !((uintptr_t)y & 15) && !((uintptr_t)weight & 15)) {
        // one wave per row (no second trip through the row loop) up to the number of amax slots
        long g = (rows + NORM_WAVES - 1) / NORM_WAVES;
        g = g < 1 ? 1 : (g > MANTIS_AMAX_PARTS ? MANTIS_AMAX_PARTS : g);
#define RMSF_LAUNCH(N_) case N_: MANTIS_LAUNCH(rmsnorm_fwd_regs_kernel<N_>, dim3((int)g), dim3(64 * NORM_WAVES), 0, (hipStream_t)stream, \
                       (const bf16_t*)x, (const bf16_t*)weight, (bf16_t*)y, rstd, (long)rows, eps, amax_parts); return mantis_check_launch();
        switch (d / 512) {
            RMSF_LAUNCH(1) RMSF_LAUNCH(2) RMSF_LAUNCH(3) RMSF_LAUNCH(4) RMSF_LAUNCH(5) RMSF_LAUNCH(6) RMSF_LAUNCH(7) RMSF_LAUNCH(8)
            RMSF_LAUNCH(10) RMSF_LAUNCH(12) RMSF_LAUNCH(16)
            default: break;
        }
#undef RMSF_LAUNCH
    }
    MANTIS_LAUNCH(rmsnorm_fwd_kernel, dim3(norm_grid(rows)), dim3(64 * NORM_WAVES), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)weight, (bf16_t*)y, rstd, (long)rows, d, eps, amax_parts);
    return mantis_check_launch();
}

// workspace: >= mantis_rmsnorm_bwd_partials(rows) * d floats.  grad_weight (+)= dW (bf16).  dres optional (fused residual-grad add).
int mantis_rmsnorm_bwd_partials(int64_t rows) {
    long g = (rows + RMSB_WAVES - 1) / RMSB_WAVES;
    return (int)(g < 1 ? 1 : (g > 512 ? 512 : g));      // one partial row per workgroup, 2 workgroups per CU
}

int mantis_rmsnorm_bwd(const void* dy, const void* x, const void* weight, const float* rstd, const void* dres, void* dx,
                       void* grad_weight, int accumulate, float* workspace, int64_t rows, int d, float* amax_parts, void* stream) {
    if (d % 8 || d <= 0 || d > 64 * 8 * NORM_MAXC) return MANTIS_EUNSUPPORTED;
    if (rows == 0) return MANTIS_OK;
    const int P = mantis_rmsnorm_bwd_partials(rows);
    if (d == RMSQ_D && (long)rows * d * 2 < (1L << 32) - (1L << 16)) {
#define RMSQ_LAUNCH(AMAX) MANTIS_LAUNCH(rmsnorm_bwd_d4096_kernel<AMAX>, dim3(P), dim3(64 * RMSQ_NQ * RMSQ_SLOTS), 0, (hipStream_t)stream, \
                       (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)weight, rstd, (const bf16_t*)dres, (bf16_t*)dx, \
                       grad_weight ? workspace : nullptr, (long)rows, amax_parts)
        if (amax_parts) RMSQ_LAUNCH(true);
        else RMSQ_LAUNCH(false);
#undef RMSQ_LAUNCH
        if (grad_weight)
            MANTIS_LAUNCH(reduce_partials_kernel, dim3(cdiv(d, REDP_COLS)), dim3(REDP_COLS * 16), 0, (hipStream_t)stream, workspace, P, d,
                          (bf16_t*)grad_weight, accumulate);
        return mantis_check_launch();
    }
#define RMSB_LAUNCH(MAXC) MANTIS_LAUNCH(rmsnorm_bwd_kernel<MAXC>, dim3(P), dim3(64 * RMSB_WAVES), 0, (hipStream_t)stream, \
                       (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)weight, rstd, (const bf16_t*)dres, (bf16_t*)dx, \
                       grad_weight ? workspace : nullptr, (long)rows, d, amax_parts)
    if (d <= 64 * 8 * 2) RMSB_LAUNCH(2);
    else if (d <= 64 * 8 * 8) RMSB_LAUNCH(8);
    else RMSB_LAUNCH(NORM_MAXC);
#undef RMSB_LAUNCH
    if (grad_weight)
        MANTIS_LAUNCH(reduce_partials_kernel, dim3(cdiv(d, REDP_COLS)), dim3(REDP_COLS * 16), 0, (hipStream_t)stream, workspace, P, d,
                           (bf16_t*)grad_weight, accumulate);
    return mantis_check_launch();
}

int mantis_layernorm_fwd(const void* x, const void* weight, const void* bias, void* y, int64_t rows, int d, float eps,
                         void* stream) {
    if (d % 8 || d <= 0) return MANTIS_EUNSUPPORTED;
    if (rows == 0) return MANTIS_OK;
    MANTIS_LAUNCH(layernorm_fwd_kernel, dim3(norm_grid(rows)), dim3(64 * NORM_WAVES), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)weight, (const bf16_t*)bias, (bf16_t*)y, (long)rows, d, eps);
    return mantis_check_launch();
}

}  // extern "C"
