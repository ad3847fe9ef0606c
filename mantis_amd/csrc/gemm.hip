// bf16 "NT" GEMM with fused epilogues for gfx950:   C[M,N] = epi( A[M,K] . B[N,K]^T )
//
// Replaces (reference path): every nn.Linear on the hot path -- ViT q/k/v/out/fc1/fc2 and the patch-embedding conv
// (transformers/models/siglip/modeling_siglip.py:124-130,267-322), the projector
// (/root/reference/mantis/models/mllava/modeling_llava.py:106-118), Llama q/k/v/o/gate/up/down and lm_head
// (transformers/models/llama/modeling_llama.py:163-176,229-280,438-492) -- which today run in cuBLAS/hipBLASLt.
// Backward GEMMs (dX = dY.W, dW = dY^T.X) use the same kernel on transposed operands (rope.hip: mantis_transpose).
//
// Structure (MFMA-bound, fp32 accumulate):
//   * 128x128 output tile per 256-thread workgroup (4 waves, 2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 blocks
//   * K step 64; A/B tiles go HBM -> LDS with global_load_lds (16 B per lane, no VGPR round trip), double buffered,
//     one barrier per K step; the LDS image is XOR-swizzled through the *source* address so the ds_read_b128
//     fragment reads are <= 2-way bank conflicted (cdna guide T2 / rule 21)
//   * operands are fed swapped (mfma(a = B rows, b = A rows)) so each lane owns ONE output row m and 4 consecutive n:
//     the epilogue (bias, GELU variants, residual add, grad accumulation) is 8-byte vector loads/stores
//   * edges: rows beyond M/N are clamped on load and predicated on store; K tails read a zero page (K % 8 == 0)
//   * workgroup -> tile map is XCD-aware (contiguous tile range per XCD, 8-row groups) so the 64 tiles resident on
//     one XCD share A/B panels in that XCD's private 4 MiB L2
// Algorithmic FLOPs per launch: 2*M*N*K.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (128 * BK * 2)  // one operand tile: 16 KiB

#define EPI_BIAS 1
#define EPI_ACT_SHIFT 1
#define EPI_ACT_MASK (7 << EPI_ACT_SHIFT)  // 0 none, 1 gelu(erf), 2 gelu(tanh), 3 quick_gelu
#define EPI_RESIDUAL 16
#define EPI_ACCUM 32

typedef __attribute__((address_space(3))) void lds_void;

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[16];

__device__ __forceinline__ float gemm_act(float x, int kind) {
    switch (kind) {
        case 1: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        case 2: {
            const float k = 0.7978845608028654f;
            return 0.5f * x * (1.f + tanhf(k * (x + 0.044715f * x * x * x)));
        }
        case 3: return x / (1.f + __expf(-1.702f * x));
        default: return x;
    }
}

// Stage one 128 x 64 operand tile.  Each wave issues 4 global_load_lds, each moving 8 rows x 128 B; lane l lands at
// LDS byte (rowblock*1024 + l*16) = (row = l>>3, slot = l&7) and fetches logical 16-B chunk (slot ^ (row&7)).
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, long ld, int row0, int rows_total, int k0, int K,
                                           char* lds_tile, int wave, int lane) {
    const int rl = lane >> 3;
    const int chunk = (lane & 7) ^ rl;
    const int k = k0 + chunk * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rb = wave * 4 + j;
        int grow = row0 + rb * 8 + rl;
        grow = grow < rows_total ? grow : rows_total - 1;
        const bf16_t* src = (k < K) ? (G + (long)grow * ld + k) : reinterpret_cast<const bf16_t*>(g_zero_page);
        __builtin_amdgcn_global_load_lds(src, (lds_void*)(lds_tile + rb * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const char* tile, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         bf16_t* __restrict__ C, int M, int N, int K, long lda, long ldb,
                                                         long ldc, const bf16_t* __restrict__ bias,
                                                         const bf16_t* __restrict__ res, long ldr, int flags, int tiles_m,
                                                         int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile assignment: workgroup b runs on XCD b % 8; give every XCD a contiguous range of tile ids.
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int g = tile_id / per_group;
    const int first_m = g * GROUP;
    const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int in_g = tile_id - g * per_group;
    const int tm_idx = first_m + in_g % gsz;
    const int tn_idx = in_g / gsz;
    const int m0 = tm_idx * BM, n0 = tn_idx * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (K + BK - 1) / BK;
    stage_tile(A, lda, m0, M, 0, K, smem, wave, lane);
    stage_tile(B, ldb, n0, N, 0, K, smem + TILE_BYTES, wave, lane);

    for (int t = 0; t < nk; ++t) {
        char* cur = smem + (t & 1) * 2 * TILE_BYTES;
        char* nxt = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nk) {
            stage_tile(A, lda, m0, M, (t + 1) * BK, K, nxt, wave, lane);
            stage_tile(B, ldb, n0, N, (t + 1) * BK, K, nxt + TILE_BYTES, wave, lane);
        }
        const char* At = cur;
        const char* Bt = cur + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = ks * 2 + (lane >> 5);
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fb[i] = read_frag(Bt, wn * 64 + i * 32 + (lane & 31), chunk);
                fa[i] = read_frag(At, wm * 64 + i * 32 + (lane & 31), chunk);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tn], fa[tm], acc[tn][tm], 0, 0, 0);
        }
    }

    // epilogue: lane owns row m, columns n = nb + 8*g4 + {0..3}
    const int act = (flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int m = m0 + wm * 64 + tm * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int nb = n0 + wn * 64 + tn * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = nb + 8 * g4;
                if (n >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[tn][tm][4 * g4 + e];
                const bool full = (n + 3 < N);
                if (flags & EPI_BIAS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (full || n + e < N) v[e] += bf2f(bias[n + e]);
                }
                if (act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gemm_act(bf2f(f2bf(v[e])), act);
                }
                bf16_t* cp = C + (long)m * ldc + n;
                if (full && ((ldc & 3) == 0) && ((ldr & 3) == 0 || !(flags & EPI_RESIDUAL))) {
                    if (flags & EPI_RESIDUAL) {
                        const u32x2 rv = *reinterpret_cast<const u32x2*>(res + (long)m * ldr + n);
                        v[0] = bf2f(f2bf(v[0])) + bf2f_lo(rv[0]);
                        v[1] = bf2f(f2bf(v[1])) + bf2f_hi(rv[0]);
                        v[2] = bf2f(f2bf(v[2])) + bf2f_lo(rv[1]);
                        v[3] = bf2f(f2bf(v[3])) + bf2f_hi(rv[1]);
                    }
                    if (flags & EPI_ACCUM) {
                        const u32x2 cv = *reinterpret_cast<const u32x2*>(cp);
                        v[0] += bf2f_lo(cv[0]);
                        v[1] += bf2f_hi(cv[0]);
                        v[2] += bf2f_lo(cv[1]);
                        v[3] += bf2f_hi(cv[1]);
                    }
                    u32x2 o;
                    o[0] = pack_bf2(v[0], v[1]);
                    o[1] = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(cp) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < N) {
                            float x = v[e];
                            if (flags & EPI_RESIDUAL) x = bf2f(f2bf(x)) + bf2f(res[(long)m * ldr + n + e]);
                            if (flags & EPI_ACCUM) x += bf2f(cp[e]);
                            cp[e] = f2bf(x);
                        }
                    }
                }
            }
        }
    }
}

extern "C" {

// C[M,N] (bf16, row stride ldc) = epilogue(A[M,K] . B[N,K]^T); A,B,C 16-B aligned, lda/ldb % 8 == 0, K % 8 == 0.
// flags: bit0 bias[n] add | bits1-3 activation (1 gelu-erf, 2 gelu-tanh, 3 quick-gelu) | bit4 + residual[m,n] (stride ldr)
//        | bit5 accumulate into C (C += result, used for gradient accumulation)
int mantis_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                        const void* bias, const void* residual, int64_t ldr, int flags, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return MANTIS_EINVAL;
    if (K % 8 || lda % 8 || ldb % 8 || lda < K || ldb < K || ldc < N) return MANTIS_EUNSUPPORTED;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return MANTIS_EUNSUPPORTED;
    if ((flags & EPI_BIAS) && !bias) return MANTIS_EINVAL;
    if ((flags & EPI_RESIDUAL) && !residual) return MANTIS_EINVAL;
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    hipLaunchKernelGGL(gemm_nt_kernel, dim3(tiles_m * tiles_n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)A,
                       (const bf16_t*)B, (bf16_t*)C, M, N, K, (long)lda, (long)ldb, (long)ldc, (const bf16_t*)bias,
                       (const bf16_t*)residual, (long)ldr, flags, tiles_m, tiles_n);
    return mantis_check_launch();
}

}  // extern "C"
