// fp8 (OCP e4m3 / e5m2) "NT" GEMM with per-tensor scaling for gfx950, and the quantisers that feed it.
//      C[M,N] (bf16) = epi( sa * sb * A8[M,K] . B8[N,K]^T ),  fp32 accumulate in the MFMA
//
// SURVEY.md section 8 row f3 / BASELINE.json configs[4] ("Qwen2-VL-7B ... fp8 MFMA"): the reference has no fp8 anywhere (its linears
// are bf16 nn.Linear: transformers/models/qwen2_vl/modeling_qwen2_vl.py:453-466,501-504), so this is an accelerated variant of those
// linears whose tolerance is stated against the bf16/fp32 oracle (tests/gpu_checks.py: fp8_*), and whose arithmetic is restated
// exactly (same rounding, same scales) by oracle/ops_ref.py: fp8_quantize / gemm_fp8_nt.
//
// Recipe (per-tensor, just-in-time scaling; no history):  q = cvt_fp8( clamp(x * (FMAX / amax(|x|))) ),  dequant factor = amax / FMAX.
//   forward   Y  = X8 . W8^T          X, W in e4m3
//   dX        dX = dY8 . (W8T)^T      dY in e5m2 (range), W8T = transposed e4m3 copy of the weight (K-contiguous along N)
//   dW        dW = dY8T . (X8T)^T     both transposed copies come out of the quantiser for free (one LDS tile transpose)
// so ONE "NT" kernel (both operands K-contiguous, what the fp8 MFMA wants: 32 consecutive K bytes per lane) serves all three.
//
// MFMA: v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (E8M0 127 = 2^0) -- on gfx950 the K = 64 scaled form is the only fp8
// matrix instruction at 2x the bf16 rate (the non-scaled 32x32x16 fp8 form runs at the bf16 rate, MI355X guide section MFMA).
// Structure: 256x256 tile per 512-thread workgroup (8 waves of 128x64) or 128x128 (4 waves of 64x64), K step 128 B, A/B tiles
// HBM -> LDS by LDS-DMA (16 B per lane) into the same [rows][128 B] rotation-swizzled image as the bf16 kernel (conflict-free
// ds_read_b128), two stages; XCD-aware tile map.  Operands are fed swapped (mfma(a = B rows, b = A rows)) so a lane owns ONE output
// row m and 4 consecutive n.  Algorithmic FLOPs per launch: 2*M*N*K; algorithmic bytes: M*K + N*K + 2*M*N.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) void f8_lds_void;

#define F8_EPI_BIAS 1
#define F8_EPI_RESIDUAL 16
#define F8_EPI_ACCUM 32
#define F8_EPI_VECSCALE 128                  // dequant_a / dequant_b are vectors: one factor per row of A (m) and per row of B (n)
#define F8_FMT_E4M3 0
#define F8_FMT_E5M2 1

static __device__ __attribute__((aligned(16))) unsigned int f8_zero_page[16];

// One global_load_lds: 8 rows x 128 B of an operand tile (row block rb) -> LDS; 16-B chunk c of row r lands at slot (c + (r >> 1)) & 7.
__device__ __forceinline__ void f8_stage_piece(const unsigned char* __restrict__ G, long ld, int row0, int rows_total, int k0, int K,
                                               char* lds_tile, int rb, int lane) {
    const int rl = lane >> 3;
    const int f = (rb * 4 + (rl >> 1)) & 7;
    const int chunk = ((lane & 7) - f) & 7;
    const int k = k0 + chunk * 16;
    int grow = row0 + rb * 8 + rl;
    grow = grow < rows_total ? grow : rows_total - 1;
    const void* src = (k < K) ? (const void*)(G + (long)grow * ld + k) : (const void*)f8_zero_page;
    __builtin_amdgcn_global_load_lds(src, (f8_lds_void*)(lds_tile + rb * 1024), 16, 0, 0);
}

template <int ROWS, int NW>
__device__ __forceinline__ void f8_stage_tile(const unsigned char* __restrict__ G, long ld, int row0, int rows_total, int k0, int K,
                                              char* lds_tile, int wave, int lane) {
    constexpr int PER = ROWS / 8 / NW;
#pragma unroll
    for (int j = 0; j < PER; ++j) f8_stage_piece(G, ld, row0, rows_total, k0, K, lds_tile, wave * PER + j, lane);
}

// 32 consecutive K bytes of one row = chunks c0, c0 + 1 (c0 even, so both sit in the same 128-B row)
__device__ __forceinline__ i32x8 f8_read_frag(const char* tile, int row, int c0) {
    const int f = (row >> 1) & 7;
    const i32x4 lo = *reinterpret_cast<const i32x4*>(tile + row * 128 + (((c0 + f) & 7) << 4));
    const i32x4 hi = *reinterpret_cast<const i32x4*>(tile + row * 128 + (((c0 + 1 + f) & 7) << 4));
    i32x8 r;
    r[0] = lo[0], r[1] = lo[1], r[2] = lo[2], r[3] = lo[3], r[4] = hi[0], r[5] = hi[1], r[6] = hi[2], r[7] = hi[3];
    return r;
}

template <int TM, int TN>
__device__ __forceinline__ void f8_epilogue(f32x16 (&acc)[TN][TM], bf16_t* __restrict__ C, int M, int N, long ldc, float scale,
                                            const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res, long ldr, int flags,
                                            int mw0, int nw0, int lane) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = mw0 + tm * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int nb = nw0 + tn * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = nb + 8 * g4;
                if (n >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[tn][tm][4 * g4 + e] * scale;
                const bool full = (n + 3 < N);
                if (flags & F8_EPI_BIAS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (full || n + e < N) v[e] += bf2f(bias[n + e]);
                }
                bf16_t* cp = C + (long)m * ldc + n;
                if (full && ((ldc & 3) == 0) && ((ldr & 3) == 0 || !(flags & F8_EPI_RESIDUAL))) {
                    if (flags & F8_EPI_RESIDUAL) {          // same rounding order as the bf16 kernel: round, then add the residual
                        const u32x2 rv = *reinterpret_cast<const u32x2*>(res + (long)m * ldr + n);
                        v[0] = bf2f(f2bf(v[0])) + bf2f_lo(rv[0]);
                        v[1] = bf2f(f2bf(v[1])) + bf2f_hi(rv[0]);
                        v[2] = bf2f(f2bf(v[2])) + bf2f_lo(rv[1]);
                        v[3] = bf2f(f2bf(v[3])) + bf2f_hi(rv[1]);
                    }
                    if (flags & F8_EPI_ACCUM) {
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
                            if (flags & F8_EPI_RESIDUAL) x = bf2f(f2bf(x)) + bf2f(res[(long)m * ldr + n + e]);
                            if (flags & F8_EPI_ACCUM) x += bf2f(cp[e]);
                            cp[e] = f2bf(x);
                        }
                    }
                }
            }
        }
    }
}

// Coalesced epilogue (same scheme as the bf16 ring kernel, csrc/gemm.hip): the MFMA layout gives every lane ONE output row, so storing
// from registers touches 32 rows x 8 B per instruction.  Instead every wave transposes its tile through a wave-private LDS strip, 64
// rows x 64 fp32 per pass (row pitch 272 B: conflict-free ds_write_b128), and reads it back as 8 lanes per row x 8 columns per lane:
// bias / residual / accumulate run on 16-B vectors and every global instruction covers whole 128-B lines.  Arithmetic and rounding
// order are those of f8_epilogue (bit-identical).
#define F8_EPI_PITCH 272
#define F8_EPI_STRIP (64 * F8_EPI_PITCH)
// SWIGLU: C = [dgate | dup][M, 2N] from dact = scale * A.B^T (rounded to bf16 first, as the unfused path stores it) and res = [gate | up]
// [M, 2N] (SwiGLU backward behind dX = dY . W_down, HF:models/qwen2_vl/modeling_qwen2_vl.py:453-466 autograd); amax_out (nullable): the
// maximum |value| of everything written (bf16 bit pattern << 16, atomicMax), i.e. exactly what the quantiser's first pass would find.
template <int TM, int TN, bool SWIGLU = false>
__device__ __forceinline__ void f8_epilogue_lds(f32x16 (&acc)[TN][TM], char* __restrict__ strip, bf16_t* __restrict__ C, int M, int N,
                                                long ldc, float scale, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                long ldr, int flags, int mw0, int nw0, int lane, unsigned int* __restrict__ amax_out = nullptr,
                                                const float* __restrict__ sa_vec = nullptr, const float* __restrict__ sb_vec = nullptr) {
    static_assert(TN == 2 && (TM % 2) == 0, "strip is 64 columns wide, two 32-row blocks per pass");
    const bool vec_ok = !(ldc & 7) && !((uintptr_t)C & 15) && (!((flags & F8_EPI_RESIDUAL) || SWIGLU) || (!(ldr & 7) && !((uintptr_t)res & 15)))
                        && (!SWIGLU || !(N & 7));
    unsigned int amax = 0;
    const int rr = lane >> 3, cc = lane & 7;
    const int n = nw0 + cc * 8;
#pragma unroll
    for (int pass = 0; pass < TM / 2; ++pass) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 v = {acc[tn][pass * 2 + t2][4 * g4] * scale, acc[tn][pass * 2 + t2][4 * g4 + 1] * scale,
                                     acc[tn][pass * 2 + t2][4 * g4 + 2] * scale, acc[tn][pass * 2 + t2][4 * g4 + 3] * scale};
                    *reinterpret_cast<f32x4*>(strip + (t2 * 32 + (lane & 31)) * F8_EPI_PITCH + (tn * 32 + 8 * g4 + 4 * (lane >> 5)) * 4) = v;
                }
        // round 4 (as in csrc/gemm.hip, profiles/r04_gemm_anatomy.md): with one load per iteration the read-back was a chain of memory
        // round trips (load -> s_waitcnt vmcnt(0) -> use); when every column of the strip lies inside N (`fastv`, wave-uniform) the
        // global operands of the whole pass -- residual / accumulate / gate|up rows, bias, per-row and per-column scales -- are requested
        // up front.  Same arithmetic in the same order.
        const bool fastv = vec_ok && (nw0 + 64 <= N);
        u32x4 pa[8], pb[8];
        float psm[8], bsv[8], sbv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bsv[e] = 0.f; sbv[e] = 1.f; }
        if (fastv) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                int m = mw0 + pass * 64 + it * 8 + rr;
                m = m < M ? m : M - 1;
                if constexpr (SWIGLU) {
                    pa[it] = *reinterpret_cast<const u32x4*>(res + (long)m * ldr + n);
                    pb[it] = *reinterpret_cast<const u32x4*>(res + (long)m * ldr + N + n);
                } else {
                    if (flags & F8_EPI_RESIDUAL) pa[it] = *reinterpret_cast<const u32x4*>(res + (long)m * ldr + n);
                    if (flags & F8_EPI_ACCUM) pb[it] = *reinterpret_cast<const u32x4*>(C + (long)m * ldc + n);
                }
                if (flags & F8_EPI_VECSCALE) psm[it] = sa_vec[m];
            }
            if (flags & F8_EPI_VECSCALE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sbv[e] = sb_vec[n + e];
            }
            if (flags & F8_EPI_BIAS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bsv[e] = bf2f(bias[n + e]);
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + rr;
            const int m = mw0 + pass * 64 + row;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(strip + row * F8_EPI_PITCH + cc * 32);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(strip + row * F8_EPI_PITCH + cc * 32 + 16);
            if (m >= M || n >= N) continue;
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const bool full = vec_ok && (n + 8 <= N);
            if (flags & F8_EPI_VECSCALE) {          // per-row scales of both operands (`scale` is 1 then): out[m, n] = acc * (sa[m] * sb[n])
                if (fastv) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= psm[it] * sbv[e];
                } else {
                    const float sm = sa_vec[m];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < N) v[e] *= sm * sb_vec[n + e];
                }
            }
            if (flags & F8_EPI_BIAS) {
                if (fastv) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bsv[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < N) v[e] += bf2f(bias[n + e]);
                }
            }
            bf16_t* cp = C + (long)m * ldc + n;
            if constexpr (SWIGLU) {
                if (full) {
                    const u32x4 g = fastv ? pa[it] : *reinterpret_cast<const u32x4*>(res + (long)m * ldr + n);
                    const u32x4 u = fastv ? pb[it] : *reinterpret_cast<const u32x4*>(res + (long)m * ldr + N + n);
                    u32x4 og, ou;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gv[2] = {bf2f_lo(g[e]), bf2f_hi(g[e])};
                        const float uv[2] = {bf2f_lo(u[e]), bf2f_hi(u[e])};
                        float rg[2], ru[2];
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const float dv = bf2f(f2bf(v[2 * e + h2]));
                            const float sg = 1.f / (1.f + __expf(-gv[h2]));
                            const float silu = gv[h2] * sg;
                            rg[h2] = dv * uv[h2] * (sg + silu * (1.f - sg));
                            ru[h2] = dv * silu;
                        }
                        og[e] = pack_bf2(rg[0], rg[1]);
                        ou[e] = pack_bf2(ru[0], ru[1]);
                        const unsigned int a0 = (og[e] << 16) & 0x7fff0000u, a1 = og[e] & 0x7fff0000u;
                        const unsigned int a2 = (ou[e] << 16) & 0x7fff0000u, a3 = ou[e] & 0x7fff0000u;
                        amax = amax > a0 ? amax : a0;
                        amax = amax > a1 ? amax : a1;
                        amax = amax > a2 ? amax : a2;
                        amax = amax > a3 ? amax : a3;
                    }
                    *reinterpret_cast<u32x4*>(cp) = og;
                    *reinterpret_cast<u32x4*>(cp + N) = ou;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (n + e < N) {
                            const float gv = bf2f(res[(long)m * ldr + n + e]), uv = bf2f(res[(long)m * ldr + N + n + e]);
                            const float dv = bf2f(f2bf(v[e]));
                            const float sg = 1.f / (1.f + __expf(-gv));
                            const float silu = gv * sg;
                            const bf16_t o0 = f2bf(dv * uv * (sg + silu * (1.f - sg))), o1 = f2bf(dv * silu);
                            cp[e] = o0;
                            cp[N + e] = o1;
                            const unsigned int a0 = ((unsigned int)o0 << 16) & 0x7fff0000u, a1 = ((unsigned int)o1 << 16) & 0x7fff0000u;
                            amax = amax > a0 ? amax : a0;
                            amax = amax > a1 ? amax : a1;
                        }
                    }
                }
                continue;
            }
            if (SWIGLU) continue;
            if (full) {
                if (flags & F8_EPI_RESIDUAL) {
                    const u32x4 rv = fastv ? pa[it] : *reinterpret_cast<const u32x4*>(res + (long)m * ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] = bf2f(f2bf(v[2 * e])) + bf2f_lo(rv[e]);
                        v[2 * e + 1] = bf2f(f2bf(v[2 * e + 1])) + bf2f_hi(rv[e]);
                    }
                }
                if (flags & F8_EPI_ACCUM) {
                    const u32x4 cv = fastv ? pb[it] : *reinterpret_cast<const u32x4*>(cp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += bf2f_lo(cv[e]);
                        v[2 * e + 1] += bf2f_hi(cv[e]);
                    }
                }
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<u32x4*>(cp) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (n + e < N) {
                        float x = v[e];
                        if (flags & F8_EPI_RESIDUAL) x = bf2f(f2bf(x)) + bf2f(res[(long)m * ldr + n + e]);
                        if (flags & F8_EPI_ACCUM) x += bf2f(cp[e]);
                        cp[e] = f2bf(x);
                    }
                }
            }
        }
    }
    if constexpr (SWIGLU) {
        if (amax_out != nullptr) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned int y = (unsigned int)__shfl_xor((int)amax, o);
                amax = amax > y ? amax : y;
            }
            if (lane == 0 && amax != 0) atomicMax(amax_out, amax);
        }
    }
}

// FA: format of the A matrix (0 e4m3, 1 e5m2); B is always e4m3 (weights / activations).
template <int BM, int BN, int WM, int WN, int FA>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_fp8_nt_kernel(
    const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, bf16_t* __restrict__ C, int M, int N, int K, long lda,
    long ldb, long ldc, const float* __restrict__ inv_scale_a, const float* __restrict__ inv_scale_b, const bf16_t* __restrict__ bias,
    const bf16_t* __restrict__ res, long ldr, int flags, int tiles_m, int tiles_n) {
    constexpr int NWN = BN / WN, NW = (BM / WM) * NWN, TM = WM / 32, TN = WN / 32;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    constexpr int SMEM = 2 * STAGE > NW * F8_EPI_STRIP ? 2 * STAGE : NW * F8_EPI_STRIP;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    // XCD-aware tile assignment (workgroup b runs on XCD b % 8): every XCD gets a contiguous range of tile ids, 8-row groups
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
    const int m0 = (first_m + in_g % gsz) * BM, n0 = (in_g / gsz) * BN;

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (K + 127) / 128;
    f8_stage_tile<BM, NW>(A, lda, m0, M, 0, K, smem, wave, lane);
    f8_stage_tile<BN, NW>(B, ldb, n0, N, 0, K, smem + A_BYTES, wave, lane);
    for (int t = 0; t < nk; ++t) {
        char* cur = smem + (t & 1) * STAGE;
        char* nxt = smem + ((t + 1) & 1) * STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nk) {
            f8_stage_tile<BM, NW>(A, lda, m0, M, (t + 1) * 128, K, nxt, wave, lane);
            f8_stage_tile<BN, NW>(B, ldb, n0, N, (t + 1) * 128, K, nxt + A_BYTES, wave, lane);
        }
        const char* At = cur;
        const char* Bt = cur + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c0 = ks * 4 + (lane >> 5) * 2;
            i32x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[i] = f8_read_frag(Bt, wn * WN + i * 32 + (lane & 31), c0);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = f8_read_frag(At, wm * WM + i * 32 + (lane & 31), c0);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[tn], fa[tm], acc[tn][tm], F8_FMT_E4M3, FA, 0,
                                                                                  0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    const float scale = (flags & F8_EPI_VECSCALE) ? 1.f : inv_scale_a[0] * inv_scale_b[0];
    __syncthreads();                   // every wave is done with the last operand stage: the LDS becomes the epilogue strips
    f8_epilogue_lds<TM, TN>(acc, smem + wave * F8_EPI_STRIP, C, M, N, ldc, scale, bias, res, ldr, flags, m0 + wm * WM, n0 + wn * WN, lane,
                            nullptr, inv_scale_a, inv_scale_b);
}

// ---------------------------------------------------------------------------------------------------------------------
// 256x256 "ring" kernel (the schedule of csrc/gemm.hip's bf16 ring kernel on fp8 operands): the whole 160 KiB LDS is a ring of ten
// 16-KiB slabs (one slab = 128 rows x 128 K-bytes of A or B), i.e. 2.5 K-steps.  Slab (t, p) -- K-step t, part p in {A rows 0-127,
// B rows 0-127, A rows 128-255, B rows 128-255} -- lives in slot (4t + p) % 10.  DMA = buffer_load_dwordx4 ... lds with a descriptor,
// per-lane offsets fixed per tile and a scalar K-step offset (no address arithmetic in the loop, out-of-range lanes read zeros).  A
// K-step is two clusters of eight 64-cycle MFMAs; every memory instruction sits in the shadow of an MFMA (hand-placed asm reads,
// sched_barrier pins, counted vmcnt / lgkmcnt); the K-step barrier stands in front of the LAST cluster of a step, so the next step's
// first fragments and the slab refill ride behind it and the data of a step is requested a full K-step before it is needed.
template <int OFF>
__device__ __forceinline__ void f8_lds_read_b128(i32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

__device__ __forceinline__ i32x8 f8_cat(const i32x4& a, const i32x4& b) {
    i32x8 r;
    r[0] = a[0], r[1] = a[1], r[2] = a[2], r[3] = a[3], r[4] = b[0], r[5] = b[1], r[6] = b[2], r[7] = b[3];
    return r;
}

template <int FA, bool SWIGLU = false>
__global__ __launch_bounds__(512) void gemm_fp8_ring_kernel(
    const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, bf16_t* __restrict__ C, int M, int N, int K, long lda,
    long ldb, long ldc, const float* __restrict__ inv_scale_a, const float* __restrict__ inv_scale_b, const bf16_t* __restrict__ bias,
    const bf16_t* __restrict__ res, long ldr, int flags, int tiles_m, int tiles_n, unsigned int* __restrict__ amax_out) {
    constexpr int TM = 4, TN = 2, SLAB = 16384, PPW = 2;   // 8 waves: 2 (M) x 4 (N), each 128 x 64
    __shared__ __attribute__((aligned(16))) char smem[10 * SLAB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nk = (K + 127) / 128;
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
    const int m0 = (first_m + in_g % gsz) * 256, n0 = (in_g / gsz) * 256;

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(unsigned)((long)M * lda), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(unsigned)((long)N * ldb), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned voA[2][PPW], voB[2][PPW];
    int kc[PPW];                      // K byte (within a K-step) this lane's 16-B chunk starts at, per piece
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int piece = PPW * wave + j;
        const int rl = lane >> 3, fz = (piece * 4 + (rl >> 1)) & 7, chunk = (lane & 7) ^ fz;
        kc[j] = chunk * 16;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            int grow = m0 + half * 128 + piece * 8 + rl;
            grow = grow < M ? grow : M - 1;
            voA[half][j] = (unsigned)((long)grow * lda + chunk * 16);
            grow = n0 + half * 128 + piece * 8 + rl;
            grow = grow < N ? grow : N - 1;
            voB[half][j] = (unsigned)((long)grow * ldb + chunk * 16);
        }
    }
    auto issue1 = [&](int t, int p, int j) {
        char* dst = smem + ((4 * t + p) % 10) * SLAB;
        const int half = p >> 1;
        const int krem = K - t * 128;              // K bytes of this K-step inside the range (<= 0: the whole step reads zeros)
        f8_lds_void* d = (f8_lds_void*)(dst + (PPW * wave + j) * 1024);
        const unsigned so = (unsigned)t * 128u;
        if (p & 1) {
            const unsigned vo = (kc[j] < krem) ? voB[half][j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, d, 16, vo, so, 0, 0);
        } else {
            const unsigned vo = (kc[j] < krem) ? voA[half][j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, d, 16, vo, so, 0, 0);
        }
    };
    auto issue = [&](int t, int p) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) issue1(t, p, j);
    };
    issue(0, 0); issue(0, 1); issue(0, 2); issue(0, 3);
    issue(1, 0); issue(1, 1); issue(1, 2); issue(1, 3);

    const unsigned rowoff = (unsigned)(lane & 31) * 128u;
    const unsigned f = ((unsigned)(lane & 31) >> 1) & 7u;
    unsigned xo[2][2];                // [cluster][half]: 32 consecutive K bytes = chunks 4*ks + 2*kg + {0, 1}
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) xo[ks][h] = rowoff + ((((unsigned)(ks * 4 + (lane >> 5) * 2 + h)) ^ f) << 4);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    i32x4 fa[2][TM][2], fb[2][TN][2];

    auto slab_base = [&](int t, unsigned& a_base, unsigned& b_base) {
        a_base = lds0 + (unsigned)((4 * t + 2 * wm) % 10) * SLAB;                                              // A half wm
        b_base = lds0 + (unsigned)((4 * t + 1 + 2 * (wn >> 1)) % 10) * SLAB + (unsigned)(wn & 1) * 8192u;     // B half wn >> 1, rows (wn & 1) * 64
    };
    // the 12 fragment reads of a cluster (6 fragments x 2 halves); op 0-3: B, op 4-11: A
    auto frag_op = [&](int op, int ks, int buf, unsigned a_base, unsigned b_base) {
        const int h = op & 1, i = op >> 1;
        if (i == 0) f8_lds_read_b128<0>(fb[buf][0][h], b_base + xo[ks][h]);
        else if (i == 1) f8_lds_read_b128<4096>(fb[buf][1][h], b_base + xo[ks][h]);
        else if (i == 2) f8_lds_read_b128<0>(fa[buf][0][h], a_base + xo[ks][h]);
        else if (i == 3) f8_lds_read_b128<4096>(fa[buf][1][h], a_base + xo[ks][h]);
        else if (i == 4) f8_lds_read_b128<8192>(fa[buf][2][h], a_base + xo[ks][h]);
        else f8_lds_read_b128<12288>(fa[buf][3][h], a_base + xo[ks][h]);
    };
    // 12 ops over the 8 MFMA shadows of a cluster: two per slot in the first six, so the last request is two MFMAs (128+ cycles) old
    // when the next cluster asks for it (spreading them 2,2,2,2,1,1,1,1 measured the same: LDS latency is not what bounds the loop)
    auto slot_ops = [&](int i, int ks, int buf, unsigned a_base, unsigned b_base) {
        if (i < 6) { frag_op(2 * i, ks, buf, a_base, b_base); frag_op(2 * i + 1, ks, buf, a_base, b_base); }
    };

    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // step 0 landed (this wave's pieces); step 1 may be in flight
    __builtin_amdgcn_s_barrier();
    {
        unsigned a0, b0;
        slab_base(0, a0, b0);
#pragma unroll
        for (int op = 0; op < 12; ++op) frag_op(op, 0, 0, a0, b0);
    }
    for (int t = 0; t < nk; ++t) {
        unsigned a_base, b_base, a_next, b_next;
        slab_base(t, a_base, b_base);
        slab_base(t + 1, a_next, b_next);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cb = ks, nb = ks ^ 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // fragments of cluster ks (requested a cluster ago)
            if (ks == 1) {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // all but step t+2's parts 0,1 (this wave's 4 newest pieces) landed
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TN * TM; ++i) {
                const int tn = i / TM, tm = i % TM;
                acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f8_cat(fb[cb][tn][0], fb[cb][tn][1]),
                                                                              f8_cat(fa[cb][tm][0], fa[cb][tm][1]), acc[tn][tm], F8_FMT_E4M3,
                                                                              FA, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                if (ks == 0) slot_ops(i, 1, nb, a_base, b_base);
                else slot_ops(i, 0, nb, a_next, b_next);                 // first fragments of step t + 1, behind the barrier
                // refill: (t+2: 0,1) go to the slabs released one barrier earlier (cluster 0); (t+2: 2,3) right behind the barrier
                // that released the slabs of step t (needed one step later: a full K-step of lead)
                if (i >= 4) { const int qq = i - 4; issue1(t + 2, (ks == 0 ? 0 : 2) + (qq >> 1), qq & 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // retire the trailing DMA pieces and the last cluster's fragment reads ...
    __syncthreads();                                   // ... and every wave's fragment reads: the LDS becomes epilogue scratch
    const float scale = (flags & F8_EPI_VECSCALE) ? 1.f : inv_scale_a[0] * inv_scale_b[0];
    f8_epilogue_lds<TM, TN, SWIGLU>(acc, smem + wave * F8_EPI_STRIP, C, M, N, ldc, scale, bias, res, ldr, flags, m0 + wm * 128, n0 + wn * 64,
                                    lane, amax_out, inv_scale_a, inv_scale_b);
}

template <int FA, bool SWIGLU = false>
static int launch_fp8_ring(hipStream_t s, const unsigned char* A, const unsigned char* B, bf16_t* C, int M, int N, int K, long lda,
                           long ldb, long ldc, const float* sa, const float* sb, const bf16_t* bias, const bf16_t* res, long ldr,
                           int flags, unsigned int* amax_out = nullptr) {
    const int tiles_m = cdiv(M, 256), tiles_n = cdiv(N, 256);
    MANTIS_LAUNCH((gemm_fp8_ring_kernel<FA, SWIGLU>), dim3(tiles_m * tiles_n), dim3(512), 0, s, A, B, C, M, N, K, lda, ldb, ldc, sa, sb,
                       bias, res, ldr, flags, tiles_m, tiles_n, amax_out);
    return mantis_check_launch();
}

template <int BM, int BN, int WM, int WN, int FA>
static int launch_fp8(hipStream_t s, const unsigned char* A, const unsigned char* B, bf16_t* C, int M, int N, int K, long lda, long ldb,
                      long ldc, const float* sa, const float* sb, const bf16_t* bias, const bf16_t* res, long ldr, int flags) {
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    MANTIS_LAUNCH((gemm_fp8_nt_kernel<BM, BN, WM, WN, FA>), dim3(tiles_m * tiles_n), dim3((BM / WM) * (BN / WN) * 64), 0, s, A, B, C,
                       M, N, K, lda, ldb, ldc, sa, sb, bias, res, ldr, flags, tiles_m, tiles_n);
    return mantis_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------- quantisers
#define Q_PARTS MANTIS_AMAX_PARTS
// pass 1: per-workgroup maxima of |x| (bf16 bit patterns compare like unsigned integers once the sign is cleared)
__global__ __launch_bounds__(256) void fp8_amax_kernel(const bf16_t* __restrict__ x, long rows, int cols, long ld, float* __restrict__ parts) {
    const int cpr = cols >> 3;
    const long total = rows * cpr;
    unsigned int mx = 0;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    // four independent 16-B loads in flight per lane (HBM-bound: 2048 workgroups x 4 waves x 4 loads cover the latency)
    for (; i + 3 * stride < total; i += 4 * stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long j = i + u * stride;
            const long r = j / cpr;
            v[u] = *reinterpret_cast<const u32x4*>(x + r * ld + (j - r * cpr) * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned int lo = (v[u][e] << 16) & 0x7fff0000u, hi = v[u][e] & 0x7fff0000u;
                mx = mx > lo ? mx : lo;
                mx = mx > hi ? mx : hi;
            }
    }
    for (; i < total; i += stride) {
        const long r = i / cpr;
        const int c = (int)(i - r * cpr);
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + r * ld + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int lo = (v[e] << 16) & 0x7fff0000u, hi = v[e] & 0x7fff0000u;
            mx = mx > lo ? mx : lo;
            mx = mx > hi ? mx : hi;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int y = (unsigned int)__shfl_xor((int)mx, o);
        mx = mx > y ? mx : y;
    }
    __shared__ unsigned int sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) mx = mx > sm[w] ? mx : sm[w];
        parts[blockIdx.x] = __uint_as_float(mx);
    }
}

template <int FMT>
__device__ __forceinline__ unsigned int cvt4(float a, float b, float c, float d) {
    int w = 0;
    if (FMT == F8_FMT_E4M3) {
        w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    } else {
        w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
    }
    return (unsigned int)w;
}

// pass 2: scale = FMAX / amax; q = cvt(clamp(x * scale)); row-major copy and (optionally) the transposed copy through a 128 x 128 LDS
// tile, so both outputs are written in whole 128-B row segments.  state[0] = amax, state[1] = scale, state[2] = amax / FMAX (the
// dequant factor the GEMM epilogue reads).
#define QT_TILE 128
#define QT_PITCH 132
template <int FMT>
__global__ __launch_bounds__(256) void fp8_cast_kernel(const bf16_t* __restrict__ x, long rows, int cols, long ld,
                                                       const float* __restrict__ parts, int nparts, unsigned char* __restrict__ q, long ldq,
                                                       unsigned char* __restrict__ qt, long ldt, long rows_pad, float* __restrict__ state) {
    __shared__ float s_scale;
    __shared__ unsigned int sm[4];
    __shared__ __attribute__((aligned(16))) unsigned char tile[QT_TILE * QT_PITCH];
    const float FMAX = FMT == F8_FMT_E4M3 ? 448.f : 57344.f;
    {
        unsigned int mx = 0;       // nparts = Q_PARTS per-workgroup maxima of pass 1, or 1 = an amax a producer kernel already took
        for (int u = threadIdx.x; u < nparts; u += 256) {
            const unsigned int y = __float_as_uint(parts[u]);
            mx = mx > y ? mx : y;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned int y = (unsigned int)__shfl_xor((int)mx, o);
            mx = mx > y ? mx : y;
        }
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) mx = mx > sm[w] ? mx : sm[w];
            const float amax = __uint_as_float(mx);
            const float sc = amax > 0.f ? FMAX / amax : 1.f;
            s_scale = sc;
            if (blockIdx.x == 0 && blockIdx.y == 0) {
                state[0] = amax;
                state[1] = sc;
                state[2] = amax > 0.f ? amax / FMAX : 1.f;
            }
        }
        __syncthreads();
    }
    const float sc = s_scale;
    const long r0 = (long)blockIdx.y * QT_TILE;
    const int c0 = blockIdx.x * QT_TILE;
    const int t = threadIdx.x;
    {
        const int cc = (t & 7) * 16;               // 8 lanes x 16 columns = one 128-B row segment of the fp8 output
        const int c = c0 + cc;
        u32x4 v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = r0 + (t >> 3) + 32 * u;
            if (r < rows && c < cols) {            // cols % 16 == 0: a 16-column group is either fully inside or fully outside
                v0[u] = *reinterpret_cast<const u32x4*>(x + r * ld + c);
                v1[u] = *reinterpret_cast<const u32x4*>(x + r * ld + c + 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rl = (t >> 3) + 32 * u;
            const long r = r0 + rl;
            unsigned int w[4] = {0u, 0u, 0u, 0u};
            if (r < rows && c < cols) {
                float f[16];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f[2 * e] = bf2f_lo(v0[u][e]), f[2 * e + 1] = bf2f_hi(v0[u][e]);
                    f[8 + 2 * e] = bf2f_lo(v1[u][e]), f[8 + 2 * e + 1] = bf2f_hi(v1[u][e]);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) f[e] = fminf(fmaxf(f[e] * sc, -FMAX), FMAX);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = cvt4<FMT>(f[4 * e], f[4 * e + 1], f[4 * e + 2], f[4 * e + 3]);
                u32x4 o;
                o[0] = w[0], o[1] = w[1], o[2] = w[2], o[3] = w[3];
                *reinterpret_cast<u32x4*>(q + r * ldq + c) = o;
            }
            if (qt != nullptr) {
                unsigned int* tp = reinterpret_cast<unsigned int*>(tile + rl * QT_PITCH + cc);
                tp[0] = w[0], tp[1] = w[1], tp[2] = w[2], tp[3] = w[3];
            }
        }
    }
    if (qt == nullptr) return;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int cl = (t >> 3) + 32 * u, rr = (t & 7) * 16;     // output row = source column c0 + cl; 16 source rows r0 + rr ..
        const int c = c0 + cl;
        const long r = r0 + rr;
        if (c < cols && r < rows_pad) {                          // rows_pad % 16 == 0; source rows >= rows were staged as zeros
            unsigned int w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned int acc = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) acc |= (unsigned int)tile[(rr + 4 * e + b) * QT_PITCH + cl] << (8 * b);
                w[e] = acc;
            }
            u32x4 o;
            o[0] = w[0], o[1] = w[1], o[2] = w[2], o[3] = w[3];
            *reinterpret_cast<u32x4*>(qt + (long)c * ldt + r) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ row / column scaled quantiser ("2-D")
// One scale per ROW for the row-major copy (tokens of an activation, output features of a weight) and one per COLUMN for the transposed
// copy, so each of the three GEMMs of a linear sees per-row scales on both of its "NT" operands:
//   forward  X8.q  (per token)        . W8.q  (per output feature)
//   dX       dY8.q (per token)        . W8.qt (per input feature = column of W)
//   dW       dY8.qt (per out feature) . X8.qt (per input feature)
// pass 1: |x| maxima per row and per column (bf16 magnitudes compare as unsigned integers), atomically into rowmax[rows] / colmax[cols]
__global__ __launch_bounds__(256) void fp8_amax2d_kernel(const bf16_t* __restrict__ x, long rows, int cols, long ld,
                                                         unsigned int* __restrict__ rowmax, unsigned int* __restrict__ colmax) {
    __shared__ unsigned int cm[QT_TILE];
    const int t = threadIdx.x;
    if (t < QT_TILE) cm[t] = 0u;
    __syncthreads();
    const long r0 = (long)blockIdx.y * QT_TILE;
    const int c0 = blockIdx.x * QT_TILE;
    const int cc = (t & 7) * 16;
    const int c = c0 + cc;
    unsigned int colm[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) colm[e] = 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long r = r0 + (t >> 3) + 32 * u;
        unsigned int rm = 0u;
        if (r < rows && c < cols) {
            const u32x4 v0 = *reinterpret_cast<const u32x4*>(x + r * ld + c);
            const u32x4 v1 = *reinterpret_cast<const u32x4*>(x + r * ld + c + 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned int a0 = (v0[e] << 16) & 0x7fff0000u, a1 = v0[e] & 0x7fff0000u;
                const unsigned int b0 = (v1[e] << 16) & 0x7fff0000u, b1 = v1[e] & 0x7fff0000u;
                colm[2 * e] = colm[2 * e] > a0 ? colm[2 * e] : a0;
                colm[2 * e + 1] = colm[2 * e + 1] > a1 ? colm[2 * e + 1] : a1;
                colm[8 + 2 * e] = colm[8 + 2 * e] > b0 ? colm[8 + 2 * e] : b0;
                colm[8 + 2 * e + 1] = colm[8 + 2 * e + 1] > b1 ? colm[8 + 2 * e + 1] : b1;
                const unsigned int m01 = a0 > a1 ? a0 : a1, m23 = b0 > b1 ? b0 : b1;
                const unsigned int mm = m01 > m23 ? m01 : m23;
                rm = rm > mm ? rm : mm;
            }
        }
        // the 8 lanes of one row are neighbours in the wave
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const unsigned int y = (unsigned int)__shfl_xor((int)rm, o);
            rm = rm > y ? rm : y;
        }
        if ((t & 7) == 0 && r < rows && rm != 0u) atomicMax(rowmax + r, rm);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (colm[e] != 0u) atomicMax(&cm[cc + e], colm[e]);
    __syncthreads();
    if (t < QT_TILE && c0 + t < cols && cm[t] != 0u) atomicMax(colmax + c0 + t, cm[t]);
}

// pass 2: q[r, c] = cvt(clamp(x * FMAX / rowmax[r])), qt[c, r] = cvt(clamp(x * FMAX / colmax[c])); row_dequant[r] = rowmax[r] / FMAX and
// col_dequant[c] = colmax[c] / FMAX (1 where the maximum is 0) are what the GEMM epilogue multiplies by.  q or qt may be NULL.
template <int FMT>
__global__ __launch_bounds__(256) void fp8_cast2d_kernel(const bf16_t* __restrict__ x, long rows, int cols, long ld,
                                                         const unsigned int* __restrict__ rowmax, const unsigned int* __restrict__ colmax,
                                                         unsigned char* __restrict__ q, long ldq, float* __restrict__ row_dequant,
                                                         unsigned char* __restrict__ qt, long ldt, long rows_pad,
                                                         float* __restrict__ col_dequant) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[QT_TILE * QT_PITCH];
    const float FMAX = FMT == F8_FMT_E4M3 ? 448.f : 57344.f;
    const long r0 = (long)blockIdx.y * QT_TILE;
    const int c0 = blockIdx.x * QT_TILE;
    const int t = threadIdx.x;
    const int cc = (t & 7) * 16;
    const int c = c0 + cc;
    float csc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) csc[e] = 1.f;
    if (qt != nullptr && c < cols) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float am = __uint_as_float(colmax[c + e]);
            csc[e] = am > 0.f ? FMAX / am : 1.f;
        }
        if (blockIdx.y == 0 && t < 8) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float am = __uint_as_float(colmax[c + e]);
                col_dequant[c + e] = am > 0.f ? am / FMAX : 1.f;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rl = (t >> 3) + 32 * u;
        const long r = r0 + rl;
        unsigned int w[4] = {0u, 0u, 0u, 0u}, wt[4] = {0u, 0u, 0u, 0u};
        if (r < rows && c < cols) {
            const u32x4 v0 = *reinterpret_cast<const u32x4*>(x + r * ld + c);
            const u32x4 v1 = *reinterpret_cast<const u32x4*>(x + r * ld + c + 8);
            float f[16];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = bf2f_lo(v0[e]), f[2 * e + 1] = bf2f_hi(v0[e]);
                f[8 + 2 * e] = bf2f_lo(v1[e]), f[8 + 2 * e + 1] = bf2f_hi(v1[e]);
            }
            if (q != nullptr) {
                const float am = __uint_as_float(rowmax[r]);
                const float rsc = am > 0.f ? FMAX / am : 1.f;
                if (blockIdx.x == 0 && (t & 7) == 0) row_dequant[r] = am > 0.f ? am / FMAX : 1.f;
                float g[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) g[e] = fminf(fmaxf(f[e] * rsc, -FMAX), FMAX);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = cvt4<FMT>(g[4 * e], g[4 * e + 1], g[4 * e + 2], g[4 * e + 3]);
                u32x4 o;
                o[0] = w[0], o[1] = w[1], o[2] = w[2], o[3] = w[3];
                *reinterpret_cast<u32x4*>(q + r * ldq + c) = o;
            }
            if (qt != nullptr) {
#pragma unroll
                for (int e = 0; e < 16; ++e) f[e] = fminf(fmaxf(f[e] * csc[e], -FMAX), FMAX);
#pragma unroll
                for (int e = 0; e < 4; ++e) wt[e] = cvt4<FMT>(f[4 * e], f[4 * e + 1], f[4 * e + 2], f[4 * e + 3]);
            }
        }
        if (qt != nullptr) {
            unsigned int* tp = reinterpret_cast<unsigned int*>(tile + rl * QT_PITCH + cc);
            tp[0] = wt[0], tp[1] = wt[1], tp[2] = wt[2], tp[3] = wt[3];
        }
    }
    if (qt == nullptr) return;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int cl = (t >> 3) + 32 * u, rr = (t & 7) * 16;
        const int co = c0 + cl;
        const long r = r0 + rr;
        if (co < cols && r < rows_pad) {
            unsigned int w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned int acc = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) acc |= (unsigned int)tile[(rr + 4 * e + b) * QT_PITCH + cl] << (8 * b);
                w[e] = acc;
            }
            u32x4 o;
            o[0] = w[0], o[1] = w[1], o[2] = w[2], o[3] = w[3];
            *reinterpret_cast<u32x4*>(qt + (long)co * ldt + r) = o;
        }
    }
}

// one launch of the chosen kernel on a (sub-)problem; f = epilogue flag bits
static int fp8_dispatch(hipStream_t s, int variant, const void* A8, long lda, const void* B8, long ldb, void* C, long ldc, int M, int N, int K,
                        const float* dequant_a, const float* dequant_b, int fmt_a, const void* bias, const void* residual, long ldr, int f) {
    const unsigned char* A = (const unsigned char*)A8;
    const unsigned char* B = (const unsigned char*)B8;
    bf16_t* Cc = (bf16_t*)C;
    const bf16_t* bi = (const bf16_t*)bias;
    const bf16_t* re = (const bf16_t*)residual;
    if (variant == 3)
        return fmt_a == 0 ? launch_fp8_ring<0>(s, A, B, Cc, M, N, K, lda, ldb, ldc, dequant_a, dequant_b, bi, re, ldr, f)
                          : launch_fp8_ring<1>(s, A, B, Cc, M, N, K, lda, ldb, ldc, dequant_a, dequant_b, bi, re, ldr, f);
    if (variant == 2)
        return fmt_a == 0 ? launch_fp8<256, 256, 128, 64, 0>(s, A, B, Cc, M, N, K, lda, ldb, ldc, dequant_a, dequant_b, bi, re, ldr, f)
                          : launch_fp8<256, 256, 128, 64, 1>(s, A, B, Cc, M, N, K, lda, ldb, ldc, dequant_a, dequant_b, bi, re, ldr, f);
    return fmt_a == 0 ? launch_fp8<128, 128, 64, 64, 0>(s, A, B, Cc, M, N, K, lda, ldb, ldc, dequant_a, dequant_b, bi, re, ldr, f)
                      : launch_fp8<128, 128, 64, 64, 1>(s, A, B, Cc, M, N, K, lda, ldb, ldc, dequant_a, dequant_b, bi, re, ldr, f);
}

extern "C" {

// Workspace floats the quantiser needs (per-workgroup maxima of pass 1).
int mantis_fp8_quantize_ws_floats(void) { return Q_PARTS; }

// x bf16 [rows, cols] (row stride ld, elements) -> q fp8 [rows, cols] (row stride ldq bytes) and, if qt != NULL, the transposed copy
// qt [cols, rows_pad] (row stride ldt bytes; rows_pad = rows rounded up to 16, zero tail).  fmt: 0 = e4m3 (max 448), 1 = e5m2 (max
// 57344).  state float[3] <- {amax, scale = FMAX / amax, dequant = amax / FMAX}.  workspace: mantis_fp8_quantize_ws_floats() floats.
// amax_in (nullable): max |x| already taken by x's producer -- amax_in_count = 1: one float (mantis_gemm_fp8_dx_swiglu);
// = mantis_fp8_quantize_ws_floats(): per-workgroup maxima (mantis_rmsnorm_fwd / _bwd, mantis_swiglu_fwd) -- the amax pass is then skipped.
int mantis_fp8_quantize(const void* x, int64_t rows, int cols, int64_t ld, int fmt, void* q, int64_t ldq, void* qt, int64_t ldt,
                        float* state, float* workspace, const float* amax_in, int amax_in_count, void* stream) {
    if (rows <= 0 || cols <= 0 || cols % 16 || ld % 8 || ldq % 16 || ldq < cols || (fmt != 0 && fmt != 1) || !state || (!workspace && !amax_in))
        return MANTIS_EINVAL;
    if (amax_in && amax_in_count != 1 && amax_in_count != Q_PARTS) return MANTIS_EINVAL;
    const long rows_pad = (rows + 15) / 16 * 16;
    if (qt != nullptr && (ldt % 16 || ldt < rows_pad)) return MANTIS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(cols, QT_TILE), cdiv(rows_pad, QT_TILE));
    if (grid.y > 65535) return MANTIS_EUNSUPPORTED;
    // amax_in: the maximum |x| was already taken by the kernel that produced x (mantis_gemm_fp8_dx_swiglu) -- pass 1 is skipped
    if (amax_in == nullptr)
        MANTIS_LAUNCH(fp8_amax_kernel, dim3(Q_PARTS), dim3(256), 0, s, (const bf16_t*)x, (long)rows, cols, (long)ld, workspace);
    const float* parts = amax_in ? amax_in : workspace;
    const int nparts = amax_in ? amax_in_count : Q_PARTS;
    if (fmt == 0)
        MANTIS_LAUNCH(fp8_cast_kernel<F8_FMT_E4M3>, grid, dim3(256), 0, s, (const bf16_t*)x, (long)rows, cols, (long)ld, parts, nparts,
                           (unsigned char*)q, (long)ldq, (unsigned char*)qt, (long)ldt, rows_pad, state);
    else
        MANTIS_LAUNCH(fp8_cast_kernel<F8_FMT_E5M2>, grid, dim3(256), 0, s, (const bf16_t*)x, (long)rows, cols, (long)ld, parts, nparts,
                           (unsigned char*)q, (long)ldq, (unsigned char*)qt, (long)ldt, rows_pad, state);
    return mantis_check_launch();
}

// Row / column scaled variant: q fp8 [rows, cols] (nullable) quantised with ONE SCALE PER ROW, row_dequant float[rows] <- rowmax / FMAX;
// qt fp8 [cols, rows_pad] (nullable, zero tail) quantised with ONE SCALE PER COLUMN of x, col_dequant float[cols] <- colmax / FMAX.
// workspace: rows + cols floats (set to 0 here).  The vectors are what mantis_gemm_fp8_nt takes with flag 128.
int mantis_fp8_quantize_2d(const void* x, int64_t rows, int cols, int64_t ld, int fmt, void* q, int64_t ldq, float* row_dequant, void* qt,
                           int64_t ldt, float* col_dequant, float* workspace, void* stream) {
    if (rows <= 0 || cols <= 0 || cols % 16 || ld % 8 || (fmt != 0 && fmt != 1) || !workspace || (!q && !qt)) return MANTIS_EINVAL;
    if (q != nullptr && (ldq % 16 || ldq < cols || !row_dequant)) return MANTIS_EINVAL;
    const long rows_pad = (rows + 15) / 16 * 16;
    if (qt != nullptr && (ldt % 16 || ldt < rows_pad || !col_dequant)) return MANTIS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(cols, QT_TILE), cdiv(rows_pad, QT_TILE));
    if (grid.y > 65535) return MANTIS_EUNSUPPORTED;
    unsigned int* rowmax = (unsigned int*)workspace;
    unsigned int* colmax = rowmax + rows;
    if (hipMemsetAsync(workspace, 0, sizeof(float) * (size_t)(rows + cols), s) != hipSuccess) return MANTIS_ELAUNCH;
    MANTIS_LAUNCH(fp8_amax2d_kernel, grid, dim3(256), 0, s, (const bf16_t*)x, (long)rows, cols, (long)ld, rowmax, colmax);
    if (fmt == 0)
        MANTIS_LAUNCH(fp8_cast2d_kernel<F8_FMT_E4M3>, grid, dim3(256), 0, s, (const bf16_t*)x, (long)rows, cols, (long)ld, rowmax, colmax,
                      (unsigned char*)q, (long)ldq, row_dequant, (unsigned char*)qt, (long)ldt, rows_pad, col_dequant);
    else
        MANTIS_LAUNCH(fp8_cast2d_kernel<F8_FMT_E5M2>, grid, dim3(256), 0, s, (const bf16_t*)x, (long)rows, cols, (long)ld, rowmax, colmax,
                      (unsigned char*)q, (long)ldq, row_dequant, (unsigned char*)qt, (long)ldt, rows_pad, col_dequant);
    return mantis_check_launch();
}

// dgu[M, 2N] = swiglu_backward(dequant * A8[M,K] . B8[N,K]^T, gate_up[M, 2N]) in one launch (ring kernel; the [M, N] activation gradient
// never goes to HBM), A8 = dY in e5m2 (fmt_a 1) or e4m3, B8 = the transposed e4m3 copy of W_down.  amax_out (nullable): float[1] <- max
// |dgu| (set to 0 here first), ready to be handed to mantis_fp8_quantize as amax_in.
int mantis_gemm_fp8_dx_swiglu(const void* A8, int64_t lda, const void* B8, int64_t ldb, void* dgu, int64_t ld_dgu, int M, int N, int K,
                              const float* dequant_a, const float* dequant_b, int fmt_a, const void* gate_up, int64_t ld_gu, float* amax_out,
                              void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 16 || lda % 16 || ldb % 16 || lda < K || ldb < K || ld_dgu < 2 * (long)N || ld_gu < 2 * (long)N ||
        !dequant_a || !dequant_b || !gate_up || !dgu)
        return MANTIS_EINVAL;
    if (fmt_a != 0 && fmt_a != 1) return MANTIS_EINVAL;
    if ((long)M * lda >= (1L << 32) || (long)N * ldb >= (1L << 32)) return MANTIS_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (amax_out != nullptr && hipMemsetAsync(amax_out, 0, sizeof(float), s) != hipSuccess) return MANTIS_ELAUNCH;
    return fmt_a == 0 ? launch_fp8_ring<0, true>(s, (const unsigned char*)A8, (const unsigned char*)B8, (bf16_t*)dgu, M, N, K, (long)lda, (long)ldb,
                                                 (long)ld_dgu, dequant_a, dequant_b, nullptr, (const bf16_t*)gate_up, (long)ld_gu, 0,
                                                 (unsigned int*)amax_out)
                      : launch_fp8_ring<1, true>(s, (const unsigned char*)A8, (const unsigned char*)B8, (bf16_t*)dgu, M, N, K, (long)lda, (long)ldb,
                                                 (long)ld_dgu, dequant_a, dequant_b, nullptr, (const bf16_t*)gate_up, (long)ld_gu, 0,
                                                 (unsigned int*)amax_out);
}

// C[M,N] bf16 (row stride ldc elements) = epi(dequant_a * dequant_b * A8[M,K] . B8[N,K]^T); lda / ldb in bytes, K % 16 == 0.
// fmt_a: 0 e4m3 | 1 e5m2; B is e4m3.  flags: 1 bias[n] | 16 + residual[m,n] (stride ldr) | 32 accumulate into C | 128 dequant_a / dequant_b
// are vectors float[M] / float[N] (one factor per row of A / of B: mantis_fp8_quantize_2d) instead of one float each | variant << 8
// (0 auto, 1 = 128x128 tiles, 2 = 256x256 tiles, 3 = 256x256 ring kernel).  dequant_a / dequant_b: device pointers to state[2] of the quantiser.
int mantis_gemm_fp8_nt(const void* A8, int64_t lda, const void* B8, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                       const float* dequant_a, const float* dequant_b, int fmt_a, const void* bias, const void* residual, int64_t ldr,
                       int flags, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 16 || lda % 16 || ldb % 16 || lda < K || ldb < K || ldc < N || !dequant_a || !dequant_b)
        return MANTIS_EINVAL;
    if ((flags & F8_EPI_BIAS) && !bias) return MANTIS_EINVAL;
    if ((flags & F8_EPI_RESIDUAL) && !residual) return MANTIS_EINVAL;
    if (fmt_a != 0 && fmt_a != 1) return MANTIS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int variant = (flags >> 8) & 15;
    const int f = flags & 0xff;
    if (variant == 0) {
        // Wave quantisation.  256x256 tiles run one per CU (256 slots per round), 128x128 tiles two per CU.  When the big tiles leave an
        // incomplete last round (q|k|v forward: 288 tiles = 1.125 rounds), the strip of tile columns (or rows) beyond the last FULL round
        // goes to the 128x128 kernel instead: 2 rounds become 1 + a quarter-length one.  Both launches write disjoint parts of C.
        const int tm = cdiv(M, 256), tn = cdiv(N, 256);
        const long t2 = (long)tm * tn;
        const bool ring_ok = (long)M * lda < (1L << 32) && (long)N * ldb < (1L << 32);
        if (ring_ok && t2 > 256 && t2 % 256 != 0) {
            const long full = t2 / 256;
            if (tn >= tm) {
                const int cols_ring = (int)(full * 256 / tm);
                if (cols_ring > 0 && cols_ring < tn && (long)(tn - cols_ring) * tm <= 128) {      // the strip fits one round of small tiles
                    const int ns = cols_ring * 256;
                    int rc = fp8_dispatch(s, 3, A8, lda, B8, ldb, C, ldc, M, ns, K, dequant_a, dequant_b, fmt_a, bias, residual, ldr, f);
                    if (rc != MANTIS_OK) return rc;
                    return fp8_dispatch(s, 1, A8, lda, (const unsigned char*)B8 + (long)ns * ldb, ldb, (bf16_t*)C + ns, ldc, M, N - ns, K, dequant_a,
                                        (f & F8_EPI_VECSCALE) ? dequant_b + ns : dequant_b, fmt_a, bias ? (const bf16_t*)bias + ns : nullptr,
                                        residual ? (const bf16_t*)residual + ns : nullptr, ldr, f);
                }
            } else {
                const int rows_ring = (int)(full * 256 / tn);
                if (rows_ring > 0 && rows_ring < tm && (long)(tm - rows_ring) * tn <= 128) {
                    const int ms = rows_ring * 256;
                    int rc = fp8_dispatch(s, 3, A8, lda, B8, ldb, C, ldc, ms, N, K, dequant_a, dequant_b, fmt_a, bias, residual, ldr, f);
                    if (rc != MANTIS_OK) return rc;
                    return fp8_dispatch(s, 1, (const unsigned char*)A8 + (long)ms * lda, lda, B8, ldb, (bf16_t*)C + (long)ms * ldc, ldc, M - ms, N, K,
                                        (f & F8_EPI_VECSCALE) ? dequant_a + ms : dequant_a, dequant_b, fmt_a, bias,
                                        residual ? (const bf16_t*)residual + (long)ms * ldr : nullptr, ldr, f);
                }
            }
        }
        // otherwise: the ring kernel unless the 128x128 kernel fills its rounds much better (the ring is ~20 % faster per flop at equal fill)
        const long t1 = (long)cdiv(M, 128) * cdiv(N, 128);
        const double e2 = (double)t2 / (double)(((t2 + 255) / 256) * 256), e1 = (double)t1 / (double)(((t1 + 511) / 512) * 512);
        variant = (e2 * 1.20 >= e1) ? (ring_ok ? 3 : 2) : 1;
    }
    if (variant == 3 && ((long)M * lda >= (1L << 32) || (long)N * ldb >= (1L << 32))) return MANTIS_EUNSUPPORTED;     // 32-bit buffer offsets
    if (variant < 1 || variant > 3) return MANTIS_EINVAL;
    return fp8_dispatch(s, variant, A8, lda, B8, ldb, C, ldc, M, N, K, dequant_a, dequant_b, fmt_a, bias, residual, ldr, f);
}

}  // extern "C"
