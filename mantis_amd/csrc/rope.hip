// RoPE (Llama-3 rotate_half convention) fwd/bwd and the batched bf16 transpose, gfx950.
//
// Replaces (reference path): LlamaRotaryEmbedding.forward + apply_rotary_pos_emb
// (transformers/models/llama/modeling_llama.py:113-160) driven by the position_ids that
// /root/reference/mantis/models/mllava/modeling_llava.py:355 produces, and its autograd backward.
// cos/sin are computed in fp32 and stored bf16 (the reference casts them to the activation dtype, :127).
// HBM-bound: in-place over the q|k columns of the fused qkv rows, 16 B per lane.
#include "common.h"

__global__ void rope_table_kernel(const long* __restrict__ pos, const float* __restrict__ inv_freq,
                                  bf16_t* __restrict__ cosb, bf16_t* __restrict__ sinb, long R, int half) {
    const long total = R * half;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / half;
        const int j = (int)(i - r * half);
        const float f = (float)pos[r] * inv_freq[j];
        cosb[i] = f2bf(cosf(f));
        sinb[i] = f2bf(sinf(f));
    }
}

// Sectioned table: rotary frequency j takes its position from row sec[j] of pos[S][R].  Serves Qwen2-VL's multimodal RoPE
// (S = 3: temporal / height / width ids, transformers/models/qwen2_vl/modeling_qwen2_vl.py:156-170,207-213) and its vision tower's
// 2-D rotary embedding (S = 2: patch row / column, :239-248 with inv_freq = [f | f]).
__global__ void rope_table_sections_kernel(const long* __restrict__ pos, const float* __restrict__ inv_freq, const int* __restrict__ sec,
                                           bf16_t* __restrict__ cosb, bf16_t* __restrict__ sinb, long R, int half) {
    const long total = R * half;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / half;
        const int j = (int)(i - r * half);
        const float f = (float)pos[(long)sec[j] * R + r] * inv_freq[j];
        cosb[i] = f2bf(cosf(f));
        sinb[i] = f2bf(sinf(f));
    }
}

// x: [R, ld]; heads 0..nheads-1 start at column h*hd.  DIR = +1 forward (reference's bf16 rounding sequence),
// -1 backward (transpose of the rotation, single rounding).
template <int DIR>
__global__ void rope_apply_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ cosb, const bf16_t* __restrict__ sinb,
                                  long R, int nheads, int hd, long ld) {
    const int half = hd >> 1;
    const int cph = half >> 3;  // 16-B chunks per half head
    const long total = R * nheads * cph;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cph);
        const long t = i / cph;
        const int h = (int)(t % nheads);
        const long r = t / nheads;
        bf16_t* p1 = x + r * ld + (long)h * hd + c * 8;
        bf16_t* p2 = p1 + half;
        const u32x4 a = *reinterpret_cast<const u32x4*>(p1);
        const u32x4 b = *reinterpret_cast<const u32x4*>(p2);
        const u32x4 vc = *reinterpret_cast<const u32x4*>(cosb + r * half + c * 8);
        const u32x4 vs = *reinterpret_cast<const u32x4*>(sinb + r * half + c * 8);
        u32x4 o1, o2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x1[2] = {bf2f_lo(a[e]), bf2f_hi(a[e])};
            float x2[2] = {bf2f_lo(b[e]), bf2f_hi(b[e])};
            float cs[2] = {bf2f_lo(vc[e]), bf2f_hi(vc[e])};
            float sn[2] = {bf2f_lo(vs[e]), bf2f_hi(vs[e])};
            float r1[2], r2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (DIR > 0) {
                    // q*cos + rotate_half(q)*sin with every product / sum rounded to bf16 (modeling_llama.py:157-158)
                    r1[k] = bf2f(f2bf(x1[k] * cs[k])) - bf2f(f2bf(x2[k] * sn[k]));
                    r2[k] = bf2f(f2bf(x2[k] * cs[k])) + bf2f(f2bf(x1[k] * sn[k]));
                } else {
                    r1[k] = x1[k] * cs[k] + x2[k] * sn[k];
                    r2[k] = x2[k] * cs[k] - x1[k] * sn[k];
                }
            }
            o1[e] = pack_bf2(r1[0], r1[1]);
            o2[e] = pack_bf2(r2[0], r2[1]);
        }
        *reinterpret_cast<u32x4*>(p1) = o1;
        *reinterpret_cast<u32x4*>(p2) = o2;
    }
}

// Batched transpose: out[z][c][r] = in[z][r][c] for r < R, zero for R <= r < Rpad.  z = (b, h) with separate strides.
// 64x64 tile per 256-thread workgroup; pairs of rows are packed into dwords on the LDS write so the transposed
// image is written with ds_write_b32 (pitch 66 -> <=2-way conflicts) and read back as dwords, 16 B global accesses.
#define TP 66
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R,
                                                        int C, int Rpad, long ld_in, long ld_out, int nh, long in_sb,
                                                        long in_sh, long out_sb, long out_sh) {
    __shared__ unsigned int T[64 * TP / 2];
    const int z = blockIdx.z;
    const int zb = z / nh, zh = z - zb * nh;
    const bf16_t* src = in + zb * in_sb + zh * in_sh;
    bf16_t* dst = out + zb * out_sb + zh * out_sh;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    {
        const int rp = t >> 3, cc = t & 7;
        const int r = r0 + 2 * rp, c = c0 + cc * 8;
        u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
        if (c < C) {
            if (r < R) v0 = *reinterpret_cast<const u32x4*>(src + (long)r * ld_in + c);
            if (r + 1 < R) v1 = *reinterpret_cast<const u32x4*>(src + (long)(r + 1) * ld_in + c);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int lo = (v0[e] & 0xffffu) | (v1[e] << 16);
            const unsigned int hi = (v0[e] >> 16) | (v1[e] & 0xffff0000u);
            T[((cc * 8 + 2 * e) * TP + 2 * rp) >> 1] = lo;
            T[((cc * 8 + 2 * e + 1) * TP + 2 * rp) >> 1] = hi;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int cl = (t >> 3) + 32 * k, rc = t & 7;
        const int c = c0 + cl, r = r0 + rc * 8;
        if (c < C && r < Rpad) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = T[(cl * TP + rc * 8 + 2 * e) >> 1];
            *reinterpret_cast<u32x4*>(dst + (long)c * ld_out + r) = o;
        }
    }
}

static inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

extern "C" {

int mantis_rope_table(const int64_t* position_ids, const float* inv_freq, void* cos_out, void* sin_out, int64_t R,
                      int half_dim, void* stream) {
    if (R == 0) return MANTIS_OK;
    MANTIS_LAUNCH(rope_table_kernel, dim3(ew_grid(R * half_dim)), dim3(256), 0, (hipStream_t)stream,
                       (const long*)position_ids, inv_freq, (bf16_t*)cos_out, (bf16_t*)sin_out, (long)R, half_dim);
    return mantis_check_launch();
}

int mantis_rope_table_sections(const int64_t* position_ids /*[S, R]*/, const float* inv_freq /*[half]*/, const int32_t* section_of_freq,
                               void* cos_out, void* sin_out, int64_t R, int half_dim, void* stream) {
    if (R == 0) return MANTIS_OK;
    MANTIS_LAUNCH(rope_table_sections_kernel, dim3(ew_grid(R * half_dim)), dim3(256), 0, (hipStream_t)stream,
                       (const long*)position_ids, inv_freq, (const int*)section_of_freq, (bf16_t*)cos_out, (bf16_t*)sin_out, (long)R,
                       half_dim);
    return mantis_check_launch();
}

int mantis_rope_apply(void* x, const void* cos_tab, const void* sin_tab, int64_t R, int nheads, int head_dim, int64_t ld,
                      int backward, void* stream) {
    if (head_dim % 16 || ld % 8) return MANTIS_EUNSUPPORTED;
    if (R == 0 || nheads == 0) return MANTIS_OK;
    const long total = R * nheads * (head_dim / 16);
    if (backward)
        MANTIS_LAUNCH(rope_apply_kernel<-1>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x,
                           (const bf16_t*)cos_tab, (const bf16_t*)sin_tab, (long)R, nheads, head_dim, (long)ld);
    else
        MANTIS_LAUNCH(rope_apply_kernel<1>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x,
                           (const bf16_t*)cos_tab, (const bf16_t*)sin_tab, (long)R, nheads, head_dim, (long)ld);
    return mantis_check_launch();
}

// out[b][h][c][r] = in[b][h][r][c]; element strides for the two batch levels are explicit so head slices of a fused
// [rows, ld] activation can be transposed in place of a copy.  Columns r in [R, Rpad) are zero-filled.
int mantis_transpose(const void* in, void* out, int R, int C, int Rpad, int64_t ld_in, int64_t ld_out, int nb, int nh,
                     int64_t in_stride_b, int64_t in_stride_h, int64_t out_stride_b, int64_t out_stride_h, void* stream) {
    if (C % 8 || ld_in % 8 || ld_out % 8 || Rpad % 8 || Rpad < R || nb <= 0 || nh <= 0) return MANTIS_EUNSUPPORTED;
    if (in_stride_b % 8 || in_stride_h % 8 || out_stride_b % 8 || out_stride_h % 8) return MANTIS_EUNSUPPORTED;
    if (R == 0 || C == 0) return MANTIS_OK;
    if ((long)nb * nh > 65535) return MANTIS_EUNSUPPORTED;
    MANTIS_LAUNCH(transpose_kernel, dim3(cdiv(Rpad, 64), cdiv(C, 64), nb * nh), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, (bf16_t*)out, R, C, Rpad, (long)ld_in, (long)ld_out, nh, (long)in_stride_b,
                       (long)in_stride_h, (long)out_stride_b, (long)out_stride_h);
    return mantis_check_launch();
}

}  // extern "C"
