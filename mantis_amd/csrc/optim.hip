// Fused AdamW over flat buffers + global-norm clipping helpers, gfx950.
//
// Replaces (reference path): optimizer.step() / clip_grad_norm_ of the HF training loop
// (transformers/trainer.py:1785-1796, :2535-2545; lr 1e-5, wd 0, max_grad_norm 1.0 per
// /root/reference/mantis/train/scripts/train_mllava.sh:162-165) which the reference runs through DeepSpeed's fused Adam
// with fp32 master weights.  All parameters live in ONE flat bf16 arena (and one flat grad arena), so the whole model is
// a single HBM-bound launch.
//   adamw_kernel        fp32 master array: per element read g(2)+p32(4)+m(4)+v(4), write p32(4)+m(4)+v(4)+p16(2) = 28 B
//   adamw_split_kernel  the SAME fp32 master, stored as  bf16 parameter (its round-to-nearest-even upper half, the GEMM operand that
//                       exists anyway) + the master's low 16 bits + one tie bit: read g(2)+p16(2)+lo(2)+m(4)+v(4), write
//                       p16(2)+lo(2)+m(4)+v(4) = 26 B, and 2 instead of 4 bytes of state per parameter.  Reconstruction is exact:
//                       p16 = hi16 + up, up = lo > 0x8000 | (lo == 0x8000 & hi16 odd); the one case the 32 stored bits leave open
//                       (an exact tie, which rounds to the even neighbour from EITHER side) is recorded in the sign bit of exp_avg_sq,
//                       a value that is never negative.  Same arithmetic (adamw_update), so the two kernels agree bit for bit.
#include "common.h"

__device__ __forceinline__ void adamw_update(float& p, float& m, float& v, float g, float lr, float b1, float b2, float eps, float wd,
                                             float bc1, float bc2) {
    p *= (1.f - lr * wd);                         // decoupled weight decay (torch.optim.AdamW)
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    const float denom = sqrtf(v / bc2) + eps;
    p -= (lr / bc1) * (m / denom);
}

__global__ void adamw_kernel(bf16_t* __restrict__ p16, const bf16_t* __restrict__ g16, float* __restrict__ p32,
                             float* __restrict__ m, float* __restrict__ v, long n8, float lr, float b1, float b2, float eps,
                             float wd, float bc1, float bc2, const float* __restrict__ gscale) {
    const float gs = gscale ? *gscale : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const u32x4 g = *reinterpret_cast<const u32x4*>(g16 + i * 8);
        f32x4 pa = *reinterpret_cast<f32x4*>(p32 + i * 8), pb = *reinterpret_cast<f32x4*>(p32 + i * 8 + 4);
        f32x4 ma = *reinterpret_cast<f32x4*>(m + i * 8), mb = *reinterpret_cast<f32x4*>(m + i * 8 + 4);
        f32x4 va = *reinterpret_cast<f32x4*>(v + i * 8), vb = *reinterpret_cast<f32x4*>(v + i * 8 + 4);
        float gf[8], pf[8], mf[8], vf[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gf[2 * e] = bf2f_lo(g[e]) * gs; gf[2 * e + 1] = bf2f_hi(g[e]) * gs;
            pf[e] = pa[e]; pf[4 + e] = pb[e]; mf[e] = ma[e]; mf[4 + e] = mb[e]; vf[e] = va[e]; vf[4 + e] = vb[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) adamw_update(pf[e], mf[e], vf[e], gf[e], lr, b1, b2, eps, wd, bc1, bc2);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = pack_bf2(pf[2 * e], pf[2 * e + 1]);
            pa[e] = pf[e]; pb[e] = pf[4 + e]; ma[e] = mf[e]; mb[e] = mf[4 + e]; va[e] = vf[e]; vb[e] = vf[4 + e];
        }
        *reinterpret_cast<f32x4*>(p32 + i * 8) = pa; *reinterpret_cast<f32x4*>(p32 + i * 8 + 4) = pb;
        *reinterpret_cast<f32x4*>(m + i * 8) = ma;   *reinterpret_cast<f32x4*>(m + i * 8 + 4) = mb;
        *reinterpret_cast<f32x4*>(v + i * 8) = va;   *reinterpret_cast<f32x4*>(v + i * 8 + 4) = vb;
        *reinterpret_cast<u32x4*>(p16 + i * 8) = o;
    }
}

// fp32 master <- (bf16 parameter, low 16 bits, tie bit): see the header.  p, lo: 16-bit values; tie: 0 / 1
__device__ __forceinline__ float master_join(unsigned p, unsigned lo, unsigned tie) {
    const unsigned up = (lo > 0x8000u) | ((lo == 0x8000u) & tie);
    return __uint_as_float((((p - up) & 0xffffu) << 16) | lo);
}
// the tie bit of a master: its low half is exactly one half ulp of bf16 AND its upper half is odd (round-to-nearest-even went UP)
__device__ __forceinline__ unsigned master_tie(float x) {
    const unsigned b = __float_as_uint(x);
    return ((b & 0xffffu) == 0x8000u) & (b >> 16) & 1u;
}

__global__ void adamw_split_kernel(bf16_t* __restrict__ p16, const bf16_t* __restrict__ g16, unsigned short* __restrict__ lo16,
                                   float* __restrict__ m, float* __restrict__ v, long n8, float lr, float b1, float b2, float eps,
                                   float wd, float bc1, float bc2, const float* __restrict__ gscale) {
    const float gs = gscale ? *gscale : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const u32x4 g = *reinterpret_cast<const u32x4*>(g16 + i * 8);
        const u32x4 pw = *reinterpret_cast<const u32x4*>(p16 + i * 8);
        const u32x4 lw = *reinterpret_cast<const u32x4*>(lo16 + i * 8);
        f32x4 ma = *reinterpret_cast<f32x4*>(m + i * 8), mb = *reinterpret_cast<f32x4*>(m + i * 8 + 4);
        f32x4 va = *reinterpret_cast<f32x4*>(v + i * 8), vb = *reinterpret_cast<f32x4*>(v + i * 8 + 4);
        float gf[8], pf[8], mf[8], vf[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gf[2 * e] = bf2f_lo(g[e]) * gs; gf[2 * e + 1] = bf2f_hi(g[e]) * gs;
            mf[e] = ma[e]; mf[4 + e] = mb[e]; vf[e] = va[e]; vf[4 + e] = vb[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned t0 = __float_as_uint(vf[2 * e]) >> 31, t1 = __float_as_uint(vf[2 * e + 1]) >> 31;
            pf[2 * e] = master_join(pw[e] & 0xffffu, lw[e] & 0xffffu, t0);
            pf[2 * e + 1] = master_join(pw[e] >> 16, lw[e] >> 16, t1);
            vf[2 * e] = __uint_as_float(__float_as_uint(vf[2 * e]) & 0x7fffffffu);
            vf[2 * e + 1] = __uint_as_float(__float_as_uint(vf[2 * e + 1]) & 0x7fffffffu);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) adamw_update(pf[e], mf[e], vf[e], gf[e], lr, b1, b2, eps, wd, bc1, bc2);
        u32x4 o, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = pack_bf2(pf[2 * e], pf[2 * e + 1]);
            l[e] = (__float_as_uint(pf[2 * e]) & 0xffffu) | (__float_as_uint(pf[2 * e + 1]) << 16);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[e] = __uint_as_float(__float_as_uint(vf[e]) | (master_tie(pf[e]) << 31));
#pragma unroll
        for (int e = 0; e < 4; ++e) { ma[e] = mf[e]; mb[e] = mf[4 + e]; va[e] = vf[e]; vb[e] = vf[4 + e]; }
        *reinterpret_cast<u32x4*>(lo16 + i * 8) = l;
        *reinterpret_cast<f32x4*>(m + i * 8) = ma;   *reinterpret_cast<f32x4*>(m + i * 8 + 4) = mb;
        *reinterpret_cast<f32x4*>(v + i * 8) = va;   *reinterpret_cast<f32x4*>(v + i * 8 + 4) = vb;
        *reinterpret_cast<u32x4*>(p16 + i * 8) = o;
    }
}
// out[i] = the fp32 master of element i (checkpoints, tests)
__global__ void master_join_kernel(const bf16_t* __restrict__ p16, const unsigned short* __restrict__ lo16, const float* __restrict__ v,
                                   float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = master_join(p16[i], lo16[i], __float_as_uint(v[i]) >> 31);
}
// the inverse: p16 = bf16(master) (round-to-nearest-even), lo16 = its low half, the tie bit into the sign of v (|v| kept)
__global__ void master_split_kernel(const float* __restrict__ master, bf16_t* __restrict__ p16, unsigned short* __restrict__ lo16,
                                    float* __restrict__ v, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = master[i];
        p16[i] = f2bf(x);
        lo16[i] = (unsigned short)(__float_as_uint(x) & 0xffffu);
        v[i] = __uint_as_float((__float_as_uint(v[i]) & 0x7fffffffu) | (master_tie(x) << 31));
    }
}

#define SUMSQ_BLOCKS 8192
__global__ void sumsq_partial_kernel(const bf16_t* __restrict__ x, long n8, float* __restrict__ partial) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = bf2f_lo(v[e]), b = bf2f_hi(v[e]); s += a * a + b * b; }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void sumsq_finish_kernel(const float* __restrict__ partial, int P, float* __restrict__ out, int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < P; i += blockDim.x) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}
// partial[r] = sum of squares of x[off[r] .. off[r] + len[r]), one workgroup per range (the many small gradient ranges -- norm weights,
// biases -- that no fused dW GEMM covers: one launch instead of one per range).  off / len in elements, multiples of 8.
__global__ void sumsq_ranges_kernel(const bf16_t* __restrict__ x, const long* __restrict__ off_len, float* __restrict__ partial) {
    __shared__ float red[16];
    const long off = off_len[2 * blockIdx.x], n8 = off_len[2 * blockIdx.x + 1] >> 3;
    float s = 0.f;
    for (long i = threadIdx.x; i < n8; i += blockDim.x) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + off + i * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = bf2f_lo(v[e]), b = bf2f_hi(v[e]); s += a * a + b * b; }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// scale = min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ void clip_scale_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ scale, float* __restrict__ norm) {
    const float nrm = sqrtf(sumsq[0]);
    if (norm) norm[0] = nrm;
    const float c = max_norm / (nrm + 1e-6f);
    scale[0] = c < 1.f ? c : 1.f;
}

extern "C" {

int mantis_adamw(void* param_bf16, const void* grad_bf16, float* master, float* exp_avg, float* exp_avg_sq, int64_t n,
                 float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                 const float* grad_scale_dev, void* stream) {
    if (n % 8) return MANTIS_EUNSUPPORTED;
    if (n == 0) return MANTIS_OK;
    long g = (n / 8 + 255) / 256;
    g = g > 131072 ? 131072 : g;      // measured: 4096 blocks 5.6-5.8 TB/s, 65536-262144 blocks 6.0 TB/s (grid-stride loop, 28 B/element)
    MANTIS_LAUNCH(adamw_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (bf16_t*)param_bf16,
                       (const bf16_t*)grad_bf16, master, exp_avg, exp_avg_sq, (long)(n / 8), lr, beta1, beta2, eps,
                       weight_decay, bias_corr1, bias_corr2, grad_scale_dev);
    return mantis_check_launch();
}

int mantis_adamw_split(void* param_bf16, const void* grad_bf16, void* master_lo, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                       const float* grad_scale_dev, void* stream) {
    if (n % 8) return MANTIS_EUNSUPPORTED;
    if (n == 0) return MANTIS_OK;
    long g = (n / 8 + 255) / 256;
    g = g > 131072 ? 131072 : g;
    MANTIS_LAUNCH(adamw_split_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (bf16_t*)param_bf16,
                       (const bf16_t*)grad_bf16, (unsigned short*)master_lo, exp_avg, exp_avg_sq, (long)(n / 8), lr, beta1, beta2, eps,
                       weight_decay, bias_corr1, bias_corr2, grad_scale_dev);
    return mantis_check_launch();
}

int mantis_master_join(const void* param_bf16, const void* master_lo, const float* exp_avg_sq, float* master_out, int64_t n, void* stream) {
    if (n == 0) return MANTIS_OK;
    long g = (n + 255) / 256;
    g = g > 65536 ? 65536 : g;
    MANTIS_LAUNCH(master_join_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)param_bf16,
                       (const unsigned short*)master_lo, exp_avg_sq, master_out, (long)n);
    return mantis_check_launch();
}

int mantis_master_split(const float* master, void* param_bf16, void* master_lo, float* exp_avg_sq, int64_t n, void* stream) {
    if (n == 0) return MANTIS_OK;
    long g = (n + 255) / 256;
    g = g > 65536 ? 65536 : g;
    MANTIS_LAUNCH(master_split_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, master, (bf16_t*)param_bf16,
                       (unsigned short*)master_lo, exp_avg_sq, (long)n);
    return mantis_check_launch();
}

// A stream whose kernels are confined to n_cus compute units (bits [first_cu, first_cu + n_cus) of the device's CU mask; the driver
// deals consecutive bits round-robin over the XCDs, so a contiguous range is spread over all eight).  Used to run the HBM-bound
// clip + AdamW pass on 192 CUs beside the next batch's frozen vision tower on the other 64 (tools/cu_mask_probe.hip: a streaming kernel
// gets 5.5 TB/s from 192 CUs -- as much as from all 256 -- and the two masked streams run truly side by side).
int mantis_stream_create_cu_mask(int first_cu, int n_cus, void** stream_out) {
    int dev = 0, total = 0;
    if (!stream_out || hipGetDevice(&dev) != hipSuccess) return MANTIS_EINVAL;
    if (hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || total <= 0) return MANTIS_ELAUNCH;
    if (first_cu < 0 || n_cus <= 0 || first_cu + n_cus > total) return MANTIS_EINVAL;
    uint32_t mask[32] = {0};
    if (total > 32 * 32) return MANTIS_EUNSUPPORTED;
    for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask) != hipSuccess) return MANTIS_ELAUNCH;
    *stream_out = (void*)s;
    return MANTIS_OK;
}

// A stream of the LOWEST (level < 0 ... > 0: -1 highest, 0 default, 1 lowest; clamped to what the device offers) hardware-queue
// priority.  The backward's weight-gradient GEMMs are off the critical path: queued on a lowest-priority stream, their workgroups are
// dispatched where the critical path's kernels leave compute units idle (incomplete last tile rounds, tile epilogues) instead of
// sharing every round with them.  levels_out (nullable): {greatest, least} priority values of the device.
int mantis_stream_create_priority(int level, void** stream_out, int* levels_out) {
    if (!stream_out) return MANTIS_EINVAL;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return MANTIS_ELAUNCH;
    if (levels_out) { levels_out[0] = greatest; levels_out[1] = least; }
    const int prio = level < 0 ? greatest : (level > 0 ? least : 0);
    hipStream_t s = nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio) != hipSuccess) return MANTIS_ELAUNCH;
    *stream_out = (void*)s;
    return MANTIS_OK;
}

int mantis_stream_destroy(void* stream) { return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? MANTIS_OK : MANTIS_ELAUNCH; }

// out (+)= sum of n floats, one workgroup, fixed order (the tile partials of mantis_gemm_bf16_nt_sumsq): lane-strided partial sums, then the
// block-wide reduction every kernel here uses.  n is small (~1e5 for an 8 B model).
__global__ __launch_bounds__(1024) void sum_f32_kernel(const float* __restrict__ x, long n, float* __restrict__ out, int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}

int mantis_sum_f32(const float* x, int64_t n, float* out, int accumulate, void* stream) {
    if (!x || !out || n < 0) return MANTIS_EINVAL;
    MANTIS_LAUNCH(sum_f32_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, (long)n, out, accumulate);
    return mantis_check_launch();
}

int mantis_sumsq_ranges(const void* x_bf16, const int64_t* off_len, int n_ranges, float* partials, void* stream) {
    if (!x_bf16 || !off_len || !partials || n_ranges <= 0) return MANTIS_EINVAL;
    MANTIS_LAUNCH(sumsq_ranges_kernel, dim3(n_ranges), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, (const long*)off_len, partials);
    return mantis_check_launch();
}

int mantis_sumsq_partials(int64_t n) { return SUMSQ_BLOCKS; }

int mantis_sumsq(const void* x_bf16, int64_t n, float* partials_ws, float* out, int accumulate, void* stream) {
    if (n % 8) return MANTIS_EUNSUPPORTED;
    MANTIS_LAUNCH(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16,
                       (long)(n / 8), partials_ws);
    MANTIS_LAUNCH(sumsq_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partials_ws, SUMSQ_BLOCKS, out,
                       accumulate);
    return mantis_check_launch();
}

int mantis_clip_scale(const float* sumsq, float max_norm, float* scale_out, float* norm_out, void* stream) {
    MANTIS_LAUNCH(clip_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, scale_out, norm_out);
    return mantis_check_launch();
}

int mantis_version(void) { return 1; }

}  // extern "C"
