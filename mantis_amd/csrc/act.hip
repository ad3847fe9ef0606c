// Element-wise activations (SwiGLU, GELU variants) fwd/bwd and the column-sum used for bias gradients, gfx950.
//
// Replaces (reference path): LlamaMLP act_fn(gate)*up (transformers/models/llama/modeling_llama.py:163-176),
// the projector's nn.GELU (/root/reference/mantis/models/mllava/modeling_llava.py:106-118), the ViT MLP activations
// gelu_pytorch_tanh / quick_gelu (transformers/models/siglip/modeling_siglip.py:310-322), and their autograd backward.
// All HBM-bound, 16 B per lane, grid-stride.
#include "common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// gate_up: [M, 2I] (gate | up), out: [M, I].  silu(gate) rounded to bf16 before the multiply, as the reference's bf16 graph does.
__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, long M, int I, long ld_gu,
                                  float* __restrict__ amax_parts) {
    const int cpr = I >> 3;
    const long total = M * cpr;
    unsigned int umax = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cpr;
        const int c = (int)(i - r * cpr);
        const u32x4 g = *reinterpret_cast<const u32x4*>(gu + r * ld_gu + c * 8);
        const u32x4 u = *reinterpret_cast<const u32x4*>(gu + r * ld_gu + I + c * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g0 = bf2f_lo(g[e]), g1 = bf2f_hi(g[e]);
            const float s0 = bf2f(f2bf(g0 * sigmoidf_(g0))), s1 = bf2f(f2bf(g1 * sigmoidf_(g1)));
            o[e] = pack_bf2(s0 * bf2f_lo(u[e]), s1 * bf2f_hi(u[e]));
            umax = mantis_umax_bf2(umax, o[e]);
        }
        *reinterpret_cast<u32x4*>(out + r * I + c * 8) = o;
    }
    if (amax_parts) mantis_store_amax_part(umax, amax_parts);
}

// dgu[:, :I] = dact * up * silu'(gate);  dgu[:, I:] = dact * silu(gate)
__global__ void swiglu_bwd_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ gu, bf16_t* __restrict__ dgu,
                                  long M, int I, long ld_gu) {
    const int cpr = I >> 3;
    const long total = M * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cpr;
        const int c = (int)(i - r * cpr);
        const u32x4 g = *reinterpret_cast<const u32x4*>(gu + r * ld_gu + c * 8);
        const u32x4 u = *reinterpret_cast<const u32x4*>(gu + r * ld_gu + I + c * 8);
        const u32x4 da = *reinterpret_cast<const u32x4*>(dact + r * I + c * 8);
        u32x4 og, ou;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gv[2] = {bf2f_lo(g[e]), bf2f_hi(g[e])};
            float uv[2] = {bf2f_lo(u[e]), bf2f_hi(u[e])};
            float dv[2] = {bf2f_lo(da[e]), bf2f_hi(da[e])};
            float rg[2], ru[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float s = sigmoidf_(gv[h]);
                const float silu = gv[h] * s;
                rg[h] = dv[h] * uv[h] * (s + silu * (1.f - s));
                ru[h] = dv[h] * silu;
            }
            og[e] = pack_bf2(rg[0], rg[1]);
            ou[e] = pack_bf2(ru[0], ru[1]);
        }
        *reinterpret_cast<u32x4*>(dgu + r * ld_gu + c * 8) = og;
        *reinterpret_cast<u32x4*>(dgu + r * ld_gu + I + c * 8) = ou;
    }
}

// kind: 0 = gelu(erf), 1 = gelu(tanh), 2 = quick_gelu (x * sigmoid(1.702 x)), 3 = silu
__device__ __forceinline__ float act_fwd(float x, int kind) {
    switch (kind) {
        case 0: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        case 1: {                  // 0.5 x (1 + tanh(u)) == x * sigmoid(2u): the same form as the GEMM epilogue (gemm.hip: gemm_act)
            const float k2 = 2.f * 0.7978845608028654f;
            return x * __builtin_amdgcn_rcpf(1.f + __expf(-k2 * (x + 0.044715f * x * x * x)));
        }
        case 2: return x * sigmoidf_(1.702f * x);
        default: return x * sigmoidf_(x);
    }
}
__device__ __forceinline__ float act_grad(float x, int kind) {
    switch (kind) {
        case 0: {
            const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
            return cdf + x * pdf;
        }
        case 1: {
            const float k = 0.7978845608028654f;
            const float inner = k * (x + 0.044715f * x * x * x);
            const float t = tanhf(inner);
            return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * k * (1.f + 3.f * 0.044715f * x * x);
        }
        case 2: {
            const float s = sigmoidf_(1.702f * x);
            return s + 1.702f * x * s * (1.f - s);
        }
        default: {
            const float s = sigmoidf_(x);
            return s + x * s * (1.f - s);
        }
    }
}

__global__ void act_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n8, int kind) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(act_fwd(bf2f_lo(v[e]), kind), act_fwd(bf2f_hi(v[e]), kind));
        *reinterpret_cast<u32x4*>(y + i * 8) = o;
    }
}

__global__ void act_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, bf16_t* __restrict__ dx, long n8,
                               int kind) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
        const u32x4 g = *reinterpret_cast<const u32x4*>(dy + i * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = pack_bf2(bf2f_lo(g[e]) * act_grad(bf2f_lo(v[e]), kind), bf2f_hi(g[e]) * act_grad(bf2f_hi(v[e]), kind));
        *reinterpret_cast<u32x4*>(dx + i * 8) = o;
    }
}

// partial[blockIdx.y][n] = sum over this block's row range of x[m][n]; 64 columns x 4 row-lanes per workgroup
__global__ void colsum_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, long M, int N, long ld,
                                      int rows_per_block) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const long m0 = (long)blockIdx.y * rows_per_block;
    const long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
    float s = 0.f;
    if (col < N)
        for (long m = m0 + rl; m < m1; m += 4) s += bf2f(x[m * ld + col]);
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < N)
        partial[(long)blockIdx.y * N + col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void reduce_partials_kernel2(const float* __restrict__ partial, int P, int d, bf16_t* __restrict__ grad,
                                        int accumulate) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += partial[(long)p * d + j];
    if (accumulate) s += bf2f(grad[j]);
    grad[j] = f2bf(s);
}

// y = a + b (bf16), used for residual joins that are not fused into a GEMM epilogue
__global__ void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const u32x4 va = *reinterpret_cast<const u32x4*>(a + i * 8);
        const u32x4 vb = *reinterpret_cast<const u32x4*>(b + i * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(bf2f_lo(va[e]) + bf2f_lo(vb[e]), bf2f_hi(va[e]) + bf2f_hi(vb[e]));
        *reinterpret_cast<u32x4*>(y + i * 8) = o;
    }
}

static inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

extern "C" {

// amax_parts (nullable): MANTIS_AMAX_PARTS floats <- per-workgroup maxima of |out| for the fp8 quantiser that consumes it next
int mantis_swiglu_fwd(const void* gate_up, void* out, int64_t M, int I, int64_t ld_gate_up, float* amax_parts, void* stream) {
    if (I % 8 || ld_gate_up % 8) return MANTIS_EUNSUPPORTED;
    if (M == 0) return MANTIS_OK;
    MANTIS_LAUNCH(swiglu_fwd_kernel, dim3(ew_grid(M * (I / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)gate_up, (bf16_t*)out, (long)M, I, (long)ld_gate_up, amax_parts);
    return mantis_check_launch();
}

int mantis_swiglu_bwd(const void* dact, const void* gate_up, void* dgate_up, int64_t M, int I, int64_t ld_gate_up,
                      void* stream) {
    if (I % 8 || ld_gate_up % 8) return MANTIS_EUNSUPPORTED;
    if (M == 0) return MANTIS_OK;
    MANTIS_LAUNCH(swiglu_bwd_kernel, dim3(ew_grid(M * (I / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dact, (const bf16_t*)gate_up, (bf16_t*)dgate_up, (long)M, I, (long)ld_gate_up);
    return mantis_check_launch();
}

int mantis_act_fwd(const void* x, void* y, int64_t n, int kind, void* stream) {
    if (n % 8 || kind < 0 || kind > 3) return MANTIS_EUNSUPPORTED;
    if (n == 0) return MANTIS_OK;
    MANTIS_LAUNCH(act_fwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, (long)(n / 8), kind);
    return mantis_check_launch();
}

int mantis_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int kind, void* stream) {
    if (n % 8 || kind < 0 || kind > 3) return MANTIS_EUNSUPPORTED;
    if (n == 0) return MANTIS_OK;
    MANTIS_LAUNCH(act_bwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)x, (bf16_t*)dx, (long)(n / 8), kind);
    return mantis_check_launch();
}

int mantis_add(const void* a, const void* b, void* y, int64_t n, void* stream) {
    if (n % 8) return MANTIS_EUNSUPPORTED;
    if (n == 0) return MANTIS_OK;
    MANTIS_LAUNCH(add_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                       (const bf16_t*)b, (bf16_t*)y, (long)(n / 8));
    return mantis_check_launch();
}

// grad[n] (+)= sum_m x[m][n].  workspace: >= mantis_colsum_partials(M) * N floats.
int mantis_colsum_partials(int64_t M) {
    long p = (M + 127) / 128;
    return (int)(p < 1 ? 1 : (p > 64 ? 64 : p));
}

int mantis_colsum(const void* x, void* grad, int accumulate, float* workspace, int64_t M, int N, int64_t ld, void* stream) {
    if (M <= 0 || N <= 0) return MANTIS_EINVAL;
    const int P = mantis_colsum_partials(M);
    const int rpb = (int)((M + P - 1) / P);
    MANTIS_LAUNCH(colsum_partial_kernel, dim3(cdiv(N, 64), P), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       workspace, (long)M, N, (long)ld, rpb);
    MANTIS_LAUNCH(reduce_partials_kernel2, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, workspace, P, N,
                       (bf16_t*)grad, accumulate);
    return mantis_check_launch();
}

}  // extern "C"
