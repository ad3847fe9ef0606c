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

// grad[j] (+)= sum_p partial[p][j]   (fixed summation order -> deterministic).  64 columns x 16 row groups per workgroup.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partial, int P, int d,
                                                               bf16_t* __restrict__ grad, int accumulate) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    float s = 0.f;
    if (j < d)
        for (int p = g; p < P; p += 16) s += partial[(long)p * d + j];
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
#define RMSB_LAUNCH(MAXC) MANTIS_LAUNCH(rmsnorm_bwd_kernel<MAXC>, dim3(P), dim3(64 * RMSB_WAVES), 0, (hipStream_t)stream, \
                       (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)weight, rstd, (const bf16_t*)dres, (bf16_t*)dx, \
                       grad_weight ? workspace : nullptr, (long)rows, d, amax_parts)
    if (d <= 64 * 8 * 2) RMSB_LAUNCH(2);
    else if (d <= 64 * 8 * 8) RMSB_LAUNCH(8);
    else RMSB_LAUNCH(NORM_MAXC);
#undef RMSB_LAUNCH
    if (grad_weight)
        MANTIS_LAUNCH(reduce_partials_kernel, dim3(cdiv(d, 64)), dim3(1024), 0, (hipStream_t)stream, workspace, P, d,
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
