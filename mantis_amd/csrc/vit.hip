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

// NaViT image preparation of the Idefics2 path on the device, one workgroup per image slot
// (/root/reference/mantis/models/idefics2/modeling_idefics2.py):
//   real[i]        = the image has a non-zero pixel (padding images are all zero and are removed, :1636-1639)
//   patch_mask[i]  = patch (py, px) holds at least one attended pixel (:1653-1658); all ones without a pixel mask (:1642-1647)
//   pos_ids[i]     = bucketised position id of every attended patch (:190-210): nh = attended patches in column 0, nw = in row 0; the k-th
//                    attended patch in row-major order gets bucket[nh][k / nw] * side + bucket[nw][k % nw]; 0 on padding patches.
// bucket[n][j] = torch.bucketize(torch.arange(0, 1 - 1e-6, 1 / n), boundaries, right=True)[j] is a host-built table (the reference's own
// float32 arithmetic, so the ids are bit-identical by construction; [tab_n][tab_n] int32).  status[i] = 1 when the attended patches are
// not nh * nw in number (the reference's indexed assignment raises a shape mismatch there) or nh / nw exceed the table.
__global__ __launch_bounds__(256) void navit_prepare_kernel(const float* __restrict__ pix, const unsigned char* __restrict__ pmask,
                                                            int C, int H, int W, int P, int side, const int* __restrict__ bucket,
                                                            int tab_n, int* __restrict__ real, int* __restrict__ patch_mask,
                                                            int* __restrict__ pos_ids, int* __restrict__ status) {
    __shared__ int s_red[4];
    __shared__ int s_scan[256];
    __shared__ int s_nh, s_nw, s_base;
    const int img = blockIdx.x, tid = threadIdx.x;
    const int ph = H / P, pw = W / P, np = ph * pw;
    // 1. any non-zero pixel (16 B per lane; C*H*W % 4 == 0 is checked by the entry point)
    const long n4 = (long)C * H * W / 4;
    const u32x4* p4 = reinterpret_cast<const u32x4*>(pix + (long)img * C * H * W);
    unsigned nz = 0;
    for (long i = tid; i < n4; i += 256) {
        const u32x4 v = p4[i];
        nz |= (v[0] | v[1] | v[2] | v[3]) & 0x7fffffffu;      // -0.0 == 0.0 in the reference's comparison
    }
    const unsigned long long any = __ballot(nz != 0);
    if ((tid & 63) == 0) s_red[tid >> 6] = any != 0;
    if (tid == 0) { s_nh = 0; s_nw = 0; s_base = 0; }
    __syncthreads();
    if (tid == 0) real[img] = (s_red[0] | s_red[1] | s_red[2] | s_red[3]) ? 1 : 0;
    // 2. patch mask
    int* pm_out = patch_mask + (long)img * np;
    for (int q = tid; q < np; q += 256) {
        int m = 1;
        if (pmask) {
            const int py = q / pw, px = q - py * pw;
            const unsigned char* base = pmask + ((long)img * H + py * P) * W + px * P;
            m = 0;
            for (int y = 0; y < P && !m; ++y)
                for (int x = 0; x < P; ++x) m |= base[(long)y * W + x] != 0;
        }
        pm_out[q] = m;
    }
    __syncthreads();          // this workgroup's own global writes are visible to it after the barrier
    // 3. nh = attended patches in column 0, nw = in row 0
    int cnt_h = 0, cnt_w = 0;
    for (int r = tid; r < ph; r += 256) cnt_h += pm_out[r * pw];
    for (int c = tid; c < pw; c += 256) cnt_w += pm_out[c];
    if (cnt_h) atomicAdd(&s_nh, cnt_h);
    if (cnt_w) atomicAdd(&s_nw, cnt_w);
    __syncthreads();
    const int nh = s_nh, nw = s_nw;
    const bool tab_ok = nh <= tab_n - 1 && nw <= tab_n - 1 && nw > 0;
    // 4. rank of every attended patch in row-major order (chunked block scan), bucket lookup
    int* pos_out = pos_ids + (long)img * np;
    for (int q0 = 0; q0 < np; q0 += 256) {
        const int q = q0 + tid;
        const int m = q < np ? pm_out[q] : 0;
        s_scan[tid] = m;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int v = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const int k = s_base + s_scan[tid] - m;          // exclusive rank
        if (q < np) {
            int v = 0;
            if (m && tab_ok && k < nh * nw) v = bucket[nh * tab_n + k / nw] * side + bucket[nw * tab_n + k % nw];
            pos_out[q] = v;
        }
        __syncthreads();
        if (tid == 255) s_base += s_scan[255];
        __syncthreads();
    }
    if (tid == 0) status[img] = (tab_ok || (nh == 0 && nw == 0 && s_base == 0)) && s_base == nh * nw ? 0 : 1;
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

int mantis_navit_prepare(const float* pixels, const uint8_t* pixel_mask, int n_images, int C, int H, int W, int P, int side,
                         const int32_t* bucket, int tab_n, int32_t* real, int32_t* patch_mask, int32_t* pos_ids, int32_t* status,
                         void* stream) {
    if (P <= 0 || H % P || W % P || side <= 0 || tab_n <= 0 || ((long)C * H * W) % 4) return MANTIS_EINVAL;
    if (((uintptr_t)pixels & 15) || !bucket || !real || !patch_mask || !pos_ids || !status) return MANTIS_EINVAL;
    if (n_images == 0) return MANTIS_OK;
    MANTIS_LAUNCH(navit_prepare_kernel, dim3(n_images), dim3(256), 0, (hipStream_t)stream, pixels, pixel_mask, C, H, W, P, side, bucket,
                       tab_n, real, patch_mask, pos_ids, status);
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
