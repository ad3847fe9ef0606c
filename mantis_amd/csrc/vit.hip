// ViT patch-embedding front end for gfx950: pixel cast + im2col into GEMM-ready patch rows, and the
// position-embedding / class-token assembly after the patch GEMM.
//
// Replaces (reference path): the fp32->bf16 pixel cast at /root/reference/mantis/models/mllava/modeling_llava.py:434-435
// and Siglip/CLIP VisionEmbeddings.forward (transformers/models/siglip/modeling_siglip.py:175-185,
// transformers/models/clip/modeling_clip.py: class_embedding + patch conv + position add), whose Conv2d(k = stride = P)
// is exactly a [I*N, 3*P*P] x [3*P*P, d_v] GEMM over non-overlapping patches.
// HBM-bound: reads the fp32 image once (coalesced along x), writes bf16 patch rows padded to Kp (Kp % 8 == 0, zero tail).
#include "common.h"

// patches[i*N + py*G + px][c*P*P + y*P + x] = bf16(pixels[i][c][py*P + y][px*P + x])
__global__ void im2col_kernel(const float* __restrict__ pix, bf16_t* __restrict__ patches, int I, int C, int H, int W, int P,
                              int Kp) {
    const int G = W / P, Gy = H / P;
    const int K = C * P * P;
    const long total = (long)I * Gy * G * Kp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const long row = i / Kp;
        bf16_t v = 0;
        if (k < K) {
            const int px = (int)(row % G);
            const int py = (int)((row / G) % Gy);
            const int img = (int)(row / ((long)G * Gy));
            const int c = k / (P * P);
            const int rem = k - c * P * P;
            const int y = rem / P, x = rem - y * P;
            v = f2bf(pix[(((long)img * C + c) * H + py * P + y) * W + px * P + x]);
        }
        patches[i] = v;
    }
}

// out[i][n + has_cls][:] = bf16(patch_out[i*N + n][:] + pos[n + has_cls][:]);  out[i][0][:] = bf16(cls + pos[0]) if has_cls
__global__ void vit_assemble_kernel(const bf16_t* __restrict__ patch_out, const bf16_t* __restrict__ pos,
                                    const bf16_t* __restrict__ cls, bf16_t* __restrict__ out, int I, int N, int d,
                                    int has_cls) {
    const int cpr = d >> 3;
    const int NT = N + has_cls;
    const long total = (long)I * NT * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        const long row = i / cpr;
        const int n = (int)(row % NT);
        const int img = (int)(row / NT);
        const u32x4 p = *reinterpret_cast<const u32x4*>(pos + (long)n * d + c * 8);
        u32x4 a;
        if (has_cls && n == 0)
            a = *reinterpret_cast<const u32x4*>(cls + c * 8);
        else
            a = *reinterpret_cast<const u32x4*>(patch_out + ((long)img * N + n - has_cls) * d + c * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(bf2f_lo(a[e]) + bf2f_lo(p[e]), bf2f_hi(a[e]) + bf2f_hi(p[e]));
        *reinterpret_cast<u32x4*>(out + row * d + c * 8) = o;
    }
}

// drop token 0 of every image ("default" feature-select strategy, modeling_llava.py:460-461): out[i][n] = in[i][n+1]
__global__ void drop_cls_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int I, int N, int d) {
    const int cpr = d >> 3;
    const long total = (long)I * N * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        const long row = i / cpr;
        const int n = (int)(row % N);
        const int img = (int)(row / N);
        *reinterpret_cast<u32x4*>(out + row * d + c * 8) =
            *reinterpret_cast<const u32x4*>(in + ((long)img * (N + 1) + n + 1) * d + c * 8);
    }
}

// Qwen2-VL's processor already delivers flattened patches [rows, C*tp*P*P] fp32 (Conv3d with kernel == stride,
// transformers/models/qwen2_vl/modeling_qwen2_vl.py:251-274): cast to bf16 and zero-pad the row to Kp (GEMM K % 8 == 0).
__global__ void cast_pad_rows_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long rows, int K, long ld_in, int Kp) {
    const long total = rows * Kp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const long r = i / Kp;
        out[i] = k < K ? f2bf(in[r * ld_in + k]) : (bf16_t)0;
    }
}

static inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" {

int mantis_im2col(const float* pixels, void* patches, int I, int C, int H, int W, int P, int Kp, void* stream) {
    if (P <= 0 || H % P || W % P || Kp < C * P * P || Kp % 8) return MANTIS_EINVAL;
    if (I == 0) return MANTIS_OK;
    const long total = (long)I * (H / P) * (W / P) * Kp;
    MANTIS_LAUNCH(im2col_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, pixels, (bf16_t*)patches, I,
                       C, H, W, P, Kp);
    return mantis_check_launch();
}

int mantis_cast_pad_rows(const float* in, void* out, int64_t rows, int K, int64_t ld_in, int Kp, void* stream) {
    if (K <= 0 || Kp < K || Kp % 8 || ld_in < K) return MANTIS_EINVAL;
    if (rows == 0) return MANTIS_OK;
    MANTIS_LAUNCH(cast_pad_rows_kernel, dim3(ew_grid(rows * Kp)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, (long)rows, K,
                       (long)ld_in, Kp);
    return mantis_check_launch();
}

int mantis_vit_assemble(const void* patch_out, const void* pos_emb, const void* cls_emb, void* out, int I, int N, int d,
                        void* stream) {
    if (d % 8) return MANTIS_EUNSUPPORTED;
    if (I == 0) return MANTIS_OK;
    const int has_cls = cls_emb != nullptr;
    const long total = (long)I * (N + has_cls) * (d / 8);
    MANTIS_LAUNCH(vit_assemble_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)patch_out, (const bf16_t*)pos_emb, (const bf16_t*)cls_emb, (bf16_t*)out, I, N, d,
                       has_cls);
    return mantis_check_launch();
}

int mantis_drop_cls(const void* in, void* out, int I, int N, int d, void* stream) {
    if (d % 8) return MANTIS_EUNSUPPORTED;
    if (I == 0) return MANTIS_OK;
    MANTIS_LAUNCH(drop_cls_kernel, dim3(ew_grid((long)I * N * (d / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, (bf16_t*)out, I, N, d);
    return mantis_check_launch();
}

}  // extern "C"
