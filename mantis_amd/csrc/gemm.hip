// bf16 "NT" GEMM with fused epilogues for gfx950:   C[M,N] = epi( A[M,K] . B[N,K]^T )
//
// Replaces (reference path): every nn.Linear on the hot path -- ViT q/k/v/out/fc1/fc2 and the patch-embedding conv
// (transformers/models/siglip/modeling_siglip.py:124-130,267-322), the projector
// (/root/reference/mantis/models/mllava/modeling_llava.py:106-118), Llama q/k/v/o/gate/up/down and lm_head
// (transformers/models/llama/modeling_llama.py:163-176,229-280,438-492) -- which today run in cuBLAS/hipBLASLt.
//
// Structure (MFMA-bound, fp32 accumulate):
//   * BM x BN output tile per workgroup: 256x256 with 8 waves of 128x64 (ring kernel, 1 workgroup/CU), or 128x128 with 4 waves
//     of 64x64 (generic kernel, 2 workgroups/CU) for small shapes; every wave tile is built from v_mfma_f32_32x32x16_bf16 blocks
//   * K step 64; A/B tiles go HBM -> LDS by LDS-DMA (16 B per lane, no VGPR round trip).  LDS image [rows][64 k]: 16-B chunk c
//     of row r sits at slot c ^ ((r >> 1) & 7) (conflict-free ds_read_b128); the DMA writes lane-linearly, so the swizzle is
//     applied on the SOURCE address and again on the fragment read (cdna guide rule 21)
//   * ring kernel: the whole 160 KiB LDS is a ring of ten 16-KiB slabs.  DMA = buffer_load_dwordx4 ... lds with a descriptor,
//     per-lane offsets fixed per tile and a scalar K-step offset (no address arithmetic in the loop, out-of-range lanes read
//     zeros).  Every memory instruction of the loop sits in the shadow of an MFMA (hand-placed asm reads, sched_barrier pins,
//     counted vmcnt / lgkmcnt), the K-step barrier stands in front of the LAST MFMA cluster of a step so the next step's first
//     fragments and the slab refill ride behind it, and the data of a step is requested a full K-step before it is needed
//   * operands may be K-major ([K, rows]): dX = dY.W reads the weight as stored, dW = dY^T.X reads both activations as
//     stored; fragments then come from a [k][rows] LDS image through the hardware-transposing ds_read_b64_tr_b16
//   * tiles of an incomplete last round are split along K (deterministic slab reduction); epilogue through wave-private LDS
//     strips (bias, GELUs, residual, grad accumulation, fused SwiGLU backward) with 16-B coalesced global accesses
//   * measured dead ends (profiles/r01_gemm_experiments.md): BK=32 ping-pong / strictly alternating wave groups, one 128x128
//     wave tile per SIMD, 256x128 / 128x256 tiles, persistent workgroups, stream-K, explicit L2 prefetch, sc1 slabs
//   * operands are fed swapped (mfma(a = B rows, b = A rows)) so each lane owns ONE output row m and 4 consecutive n
//   * edges: rows beyond M/N are clamped on load and predicated on store; K tails read zeros (K % 8 == 0)
//   * workgroup -> tile map is XCD-aware (contiguous tile range per XCD, 8-row groups) so the tiles resident on one XCD
//     share A/B panels in that XCD's private 4 MiB L2
// Algorithmic FLOPs per launch: 2*M*N*K.
#include "gemm_ring.h"

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[16];

// One global_load_lds: 8 rows x 128 B of an operand tile (row block rb) -> LDS.
// ROT: slot = (chunk + f(row)) mod 8 (a rotation keeps the 8 lanes of a row ascending apart from one wrap, which coalesces better
// in the vector memory path: +3 % on the 128x128 kernel, measured); otherwise slot = chunk ^ f(row) (1-2 % better in the ring kernel).
template <bool ROT = false>
__device__ __forceinline__ void stage_piece(const bf16_t* __restrict__ G, long ld, int row0, int rows_total, int k0, int K,
                                            char* lds_tile, int rb, int lane) {
    const int rl = lane >> 3;
    const int f = (rb * 4 + (rl >> 1)) & 7;
    const int chunk = ROT ? (((lane & 7) - f) & 7) : ((lane & 7) ^ f);
    const int k = k0 + chunk * 8;
    int grow = row0 + rb * 8 + rl;
    grow = grow < rows_total ? grow : rows_total - 1;
    const bf16_t* src = (k < K) ? (G + (long)grow * ld + k) : reinterpret_cast<const bf16_t*>(g_zero_page);
    __builtin_amdgcn_global_load_lds(src, (lds_void*)(lds_tile + rb * 1024), 16, 0, DMA_AUX);
}

template <int ROWS, int NW>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, long ld, int row0, int rows_total, int k0, int K,
                                           char* lds_tile, int wave, int lane) {
    constexpr int PER = ROWS / 8 / NW;
#pragma unroll
    for (int j = 0; j < PER; ++j) stage_piece<true>(G, ld, row0, rows_total, k0, K, lds_tile, wave * PER + j, lane);
}

__device__ __forceinline__ bf16x8 read_frag(const char* tile, int row, int chunk) {   // generic kernel: rotation swizzle
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (((chunk + ((row >> 1) & 7)) & 7) << 4));
}

// hand-placed LDS fragment read: the compiler neither counts it nor moves it (cdna guide 5.7 form iii)
template <int OFF>
__device__ __forceinline__ void lds_read_b128(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

template <int N_>
__device__ __forceinline__ void read_frags(bf16x8 (&dst)[N_], unsigned addr) {
    lds_read_b128<0>(dst[0], addr);
    if constexpr (N_ > 1) lds_read_b128<4096>(dst[1], addr);
    if constexpr (N_ > 2) lds_read_b128<8192>(dst[2], addr);
    if constexpr (N_ > 3) lds_read_b128<12288>(dst[3], addr);
}


// ---- K-major operands (element (row, k) at G[k*ld + row]): what dX = dY.W (B = W) and dW = dY^T.X (A = dY, B = X) need.
// LDS image of a ROWS x 64 tile: [64 k][ROWS] bf16; a 64-B segment (one 32-row MFMA block of one k-row) sits at block index
// blk ^ (k & 3), so the four k-rows a ds_read_b64_tr_b16 lane group touches land on the four 64-B quarters of a bank row.
template <int ROWS>
__device__ __forceinline__ void stage_piece_km(const bf16_t* __restrict__ G, long ld, int row0, int k0, int K, char* lds_tile,
                                               int piece, int lane) {
    constexpr int CPR = ROWS / 8;        // 16-B chunks per k-row
    constexpr int RPP = 64 / CPR;        // k-rows per 1-KiB global_load_lds
    const int kk = piece * RPP + lane / CPR;
    const int slot = lane % CPR;
    const int c = slot ^ ((kk & 3) << 2);
    const int k = k0 + kk;
    const long col = (long)row0 + c * 8;
    const bf16_t* src = (k < K && col + 8 <= ld) ? (G + (long)k * ld + col) : reinterpret_cast<const bf16_t*>(g_zero_page);
    __builtin_amdgcn_global_load_lds(src, (lds_void*)(lds_tile + piece * 1024), 16, 0, DMA_AUX);
}

template <int ROWS, int NW>
__device__ __forceinline__ void stage_tile_km(const bf16_t* __restrict__ G, long ld, int row0, int k0, int K, char* lds_tile,
                                              int wave, int lane) {
    constexpr int PER = ROWS / 8 / NW;
#pragma unroll
    for (int j = 0; j < PER; ++j) stage_piece_km<ROWS>(G, ld, row0, k0, K, lds_tile, wave * PER + j, lane);
}

// fragment for MFMA block `blk` (32 rows) and k-step ks from a K-major tile: lane (i = lane & 31, kg = lane >> 5) receives
// k = ks*16 + kg*8 + 0..7 of row blk*32 + i  (two hardware-transposing reads of 4 k-rows x 16 rows each)
template <int ROWS>
__device__ __forceinline__ bf16x8 read_frag_km(const char* tile, int blk, int ks, int lane) {
    const int s = lane & 15, g16 = (lane >> 4) & 1, kg = lane >> 5;
    const int kk = ks * 16 + kg * 8 + (s >> 2);      // (kk & 3) == (s >> 2) & 3 for both halves
    const char* p = tile + kk * (ROWS * 2) + ((blk ^ ((s >> 2) & 3)) << 6) + g16 * 32 + (s & 3) * 8;
    union { gs16x4 h[2]; bf16x8 f; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_gs16x4*)p);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_gs16x4*)(p + 4 * ROWS * 2));
    return u.f;
}

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[TN][TM], bf16_t* __restrict__ C, int M, int N, long ldc,
                                              const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res, long ldr,
                                              int flags, int mw0, int nw0, int lane) {
    const int act = (flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT;
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

// Epilogue of the 8-wave 256x256 kernel (32x32x16 accumulators): the MFMA layout gives every lane ONE output row, so storing from
// registers touches 32 rows x 16 B per instruction (measured: 17 us of a 146 us K = 4096 tile, 34 us with a residual).  Instead every
// wave transposes its 128 x 64 tile through a wave-private LDS strip, 64 rows x 64 fp32 per pass (conflict-free ds_write_b128).
template <int TM, int TN, bool SWIGLU>
__device__ __forceinline__ void gemm_epilogue_lds(f32x16 (&acc)[TN][TM], char* __restrict__ strip, bf16_t* __restrict__ C, int M,
                                                  int N, long ldc, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                  long ldr, int flags, int mw0, int nw0, int lane) {
    static_assert(TN == 2 && (TM % 2) == 0, "strip is 64 columns wide, two 32-row blocks per pass");
#pragma unroll
    for (int pass = 0; pass < TM / 2; ++pass) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 v = {acc[tn][pass * 2 + t2][4 * g4], acc[tn][pass * 2 + t2][4 * g4 + 1], acc[tn][pass * 2 + t2][4 * g4 + 2],
                                     acc[tn][pass * 2 + t2][4 * g4 + 3]};
                    *reinterpret_cast<f32x4*>(strip + (t2 * 32 + (lane & 31)) * EPI_PITCH + (tn * 32 + 8 * g4 + 4 * (lane >> 5)) * 4) = v;
                }
        epi_readback64<SWIGLU>(strip, C, M, N, ldc, bias, res, ldr, flags, mw0 + pass * 64, nw0, lane);
    }
}

// Generic kernel (compiler-scheduled inner loop): 128x128 tiles for small / badly quantised shapes, any operand layout.
template <int BM, int BN, int WM, int WN, bool AKM = false, bool BKM = false>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_nt_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M, int N, int K, long lda, long ldb,
    long ldc, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res, long ldr, int flags, int tiles_m, int tiles_n) {
    constexpr int NWN = BN / WN, NW = (BM / WM) * NWN, TM = WM / 32, TN = WN / 32;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

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
    const int m0 = (first_m + in_g % gsz) * BM, n0 = (in_g / gsz) * BN;

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (K + BK - 1) / BK;
    if constexpr (AKM) stage_tile_km<BM, NW>(A, lda, m0, 0, K, smem, wave, lane);
    else stage_tile<BM, NW>(A, lda, m0, M, 0, K, smem, wave, lane);
    if constexpr (BKM) stage_tile_km<BN, NW>(B, ldb, n0, 0, K, smem + A_BYTES, wave, lane);
    else stage_tile<BN, NW>(B, ldb, n0, N, 0, K, smem + A_BYTES, wave, lane);

    {
        for (int t = 0; t < nk; ++t) {
            char* cur = smem + (t & 1) * STAGE;
            char* nxt = smem + ((t + 1) & 1) * STAGE;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 1 < nk) {
                if constexpr (AKM) stage_tile_km<BM, NW>(A, lda, m0, (t + 1) * BK, K, nxt, wave, lane);
                else stage_tile<BM, NW>(A, lda, m0, M, (t + 1) * BK, K, nxt, wave, lane);
                if constexpr (BKM) stage_tile_km<BN, NW>(B, ldb, n0, (t + 1) * BK, K, nxt + A_BYTES, wave, lane);
                else stage_tile<BN, NW>(B, ldb, n0, N, (t + 1) * BK, K, nxt + A_BYTES, wave, lane);
            }
            const char* At = cur;
            const char* Bt = cur + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int chunk = ks * 2 + (lane >> 5);
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    if constexpr (BKM) fb[i] = read_frag_km<BN>(Bt, wn * (WN / 32) + i, ks, lane);
                    else fb[i] = read_frag(Bt, wn * WN + i * 32 + (lane & 31), chunk);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (AKM) fa[i] = read_frag_km<BM>(At, wm * (WM / 32) + i, ks, lane);
                    else fa[i] = read_frag(At, wm * WM + i * 32 + (lane & 31), chunk);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tn], fa[tm], acc[tn][tm], 0, 0, 0);
            }
        }
    }

    gemm_epilogue<TM, TN>(acc, C, M, N, ldc, bias, res, ldr, flags, m0 + wm * WM, n0 + wn * WN, lane);
}

template <int BM, int BN, int WM, int WN, bool AKM = false, bool BKM = false>
static int launch_gemm(hipStream_t s, const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, long lda, long ldb,
                       long ldc, const bf16_t* bias, const bf16_t* res, long ldr, int flags) {
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    MANTIS_LAUNCH((gemm_nt_kernel<BM, BN, WM, WN, AKM, BKM>), dim3(tiles_m * tiles_n), dim3((BM / WM) * (BN / WN) * 64), 0, s, A,
                       B, C, M, N, K, lda, ldb, ldc, bias, res, ldr, flags, tiles_m, tiles_n);
    return mantis_check_launch();
}


// hand-placed transposing fragment reads for K-major operands in the ring kernel: two ds_read_b64_tr_b16 per fragment
template <int OFF>
__device__ __forceinline__ void lds_read_tr64(gs16x4& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int KS>
__device__ __forceinline__ void read_frag_km_asm(bf16x8& dst, unsigned addr) {
    union { gs16x4 h[2]; bf16x8 f; } u;
    lds_read_tr64<KS * 4096>(u.h[0], addr);
    lds_read_tr64<KS * 4096 + 1024>(u.h[1], addr);
    dst = u.f;
}
template <int KS, int H>
__device__ __forceinline__ void read_half_km_asm(bf16x8& dst, unsigned addr) {
    lds_read_tr64<KS * 4096 + H * 1024>(reinterpret_cast<gs16x4*>(&dst)[H], addr);
}
template <int N_>
__device__ __forceinline__ void read_frags_km(bf16x8 (&dst)[N_], unsigned slab, const unsigned (&xb)[N_], int ks) {
#pragma unroll
    for (int i = 0; i < N_; ++i) {
        if (ks == 0) read_frag_km_asm<0>(dst[i], slab + xb[i]);
        else if (ks == 1) read_frag_km_asm<1>(dst[i], slab + xb[i]);
        else if (ks == 2) read_frag_km_asm<2>(dst[i], slab + xb[i]);
        else read_frag_km_asm<3>(dst[i], slab + xb[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256x256 "ring" kernel: the whole 160 KiB LDS is a ring of ten 16-KiB slabs (one slab = 128 rows x 64 k of A or B), i.e.
// 2.5 K-steps.  Slab (t, p) -- K-step t, part p in {A rows 0-127, B rows 0-127, A rows 128-255, B rows 128-255} -- lives
// in slot (4t + p) % 10.  At the start of K-step t (after ONE raw s_barrier, which also retires K-step t-1's readers) the
// four freed slots are refilled with parts 2,3 of step t+1 and parts 0,1 of step t+2, so 64-96 KiB of global_load_lds are
// always in flight per CU and the loads get 1-2 K-steps of lead; the only vector-memory wait is a COUNTED
// s_waitcnt vmcnt(4) (this wave's newest two slabs may still be in flight) -- the queue is never drained in the loop.
//
// Work split: one workgroup per tile in XCD-grouped order, dispatched in rounds of #CU.  The tiles of an incomplete last round
// (M = 5624 x N = 4096 is 352 tiles = 1.375 rounds, 69 % efficient as whole tiles) are split S ways along K, S = #CU / #remainder
// tiles, so that round lasts 1/S of a tile time (split-K confined to the remainder; a full stream-K schedule was measured
// slower: its skewed K phases stop the workgroups of an XCD from sharing A/B panels in L2).  The S workgroups of a split tile
// write fp32 partials to slabs, publish (agent-scope release) and take a ticket; the last arriver acquires and sums the slabs in
// part order -- the result does not depend on who is last (bitwise reproducible), nobody waits (no co-residency assumption),
// and the counter is left at zero for the next launch.
#define SK_SLAB_FLOATS (256 * 256)
#define SK_MAX_ROUNDS 1                      // sub-rounds the remainder tiles' K parts may run as (workspace = SK_MAX_ROUNDS * #CU slabs)

template <int TM, int TN>
__device__ __forceinline__ void sk_store(const f32x16 (&acc)[TN][TM], float* __restrict__ slab, int tid) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 v = {acc[tn][tm][4 * g4], acc[tn][tm][4 * g4 + 1], acc[tn][tm][4 * g4 + 2], acc[tn][tm][4 * g4 + 3]};
                reinterpret_cast<f32x4*>(slab)[((tn * TM + tm) * 4 + g4) * 512 + tid] = v;
            }
}
template <int TM, int TN>
__device__ __forceinline__ void sk_add(f32x16 (&acc)[TN][TM], const float* __restrict__ slab, int tid) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 v = reinterpret_cast<const f32x4*>(slab)[((tn * TM + tm) * 4 + g4) * 512 + tid];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[tn][tm][4 * g4 + e] += v[e];
            }
}

template <bool AKM, bool BKM, bool SWIGLU = false>
__global__ __launch_bounds__(512) void gemm_nt_ring_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M, int N, int K, long lda, long ldb,
    long ldc, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res, long ldr, int flags, int tiles_m, int tiles_n,
    int full, int S, float* __restrict__ sk_slabs, unsigned int* __restrict__ sk_cnt) {
    constexpr int TM = 4, TN = 2, SLAB = 16384, PPW = 2;   // 8 waves: 2 (M) x 4 (N), each 128 x 64
    __shared__ __attribute__((aligned(16))) char smem[10 * SLAB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // workgroup -> unit: the workgroups of one XCD (blockIdx % 8) take a contiguous range of tile ids, so the tiles resident on
    // an XCD share A/B panels in its private L2; units [0, full) are whole tiles, the rest are (remainder tile, K part) pairs
    // with all tiles of one K part adjacent
    const int nk = (K + BK - 1) / BK;
    const int bid = blockIdx.x;
    int tile_id, part = 0;
    if (bid < full) {
        const int q = full >> 3, r = full & 7, xcd = bid & 7;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    } else {
        const int j = bid - full, nu = gridDim.x - full, rem = nu / S;
        const int q = nu >> 3, r = nu & 7, xcd = j & 7;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (j >> 3);
        part = lin / rem;
        tile_id = full + lin - part * rem;
    }
    // 32-bit on purpose (nk <= 2^15, part < S <= 8): the 64-bit divisions cost the K-split units ~300 scalar instructions before their first DMA
    const int t0 = (bid < full) ? 0 : (int)((unsigned)nk * (unsigned)part / (unsigned)S);
    const int t1 = (bid < full) ? nk : (int)((unsigned)nk * (unsigned)(part + 1) / (unsigned)S);
    // tile id -> (m0, n0): groups of 8 tile rows, column-major inside a group
    const int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int g = tile_id / per_group;
    const int first_m = g * GROUP;
    const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int in_g = tile_id - g * per_group;
    const int m0 = (first_m + in_g % gsz) * 256, n0 = (in_g / gsz) * 256;
    const int Kseg = (t1 * BK < K) ? t1 * BK : K;     // K-steps past this unit's range read the zero page

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // slab (t, p): two LDS-DMA pieces per wave (pieces 2w, 2w+1 of the 16 eight-row pieces), issued as
    // buffer_load_dwordx4 ... lds: descriptor + per-lane 32-bit offset fixed for the whole tile + scalar K-step offset, so the
    // K loop carries no address arithmetic; lanes whose chunk lies beyond the K range get an out-of-range offset (-> zeros)
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)A, 0, (int)(unsigned)(((long)(AKM ? K : M) * lda) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)B, 0, (int)(unsigned)(((long)(BKM ? K : N) * ldb) * 2), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned voA[2][PPW], voB[2][PPW];
    int kcA[PPW], kcB[PPW];          // k index (within a K-step) this lane's chunk starts at, per piece
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int piece = PPW * wave + j;
        {   // row-major piece geometry (stage_piece): 8 rows x 128 B, chunk swizzled with the row pair
            const int rl = lane >> 3, fz = (piece * 4 + (rl >> 1)) & 7, chunk = (lane & 7) ^ fz;
            if constexpr (!AKM) {
                kcA[j] = chunk * 8;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    int grow = m0 + half * 128 + piece * 8 + rl;
                    grow = grow < M ? grow : M - 1;
                    voA[half][j] = (unsigned)(((long)grow * lda + chunk * 8) * 2);
                }
            }
            if constexpr (!BKM) {
                kcB[j] = chunk * 8;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    int grow = n0 + half * 128 + piece * 8 + rl;
                    grow = grow < N ? grow : N - 1;
                    voB[half][j] = (unsigned)(((long)grow * ldb + chunk * 8) * 2);
                }
            }
        }
        {   // K-major piece geometry (stage_piece_km<128>): 4 k-rows x 256 B, column chunk swizzled with k & 3
            const int kk = piece * 4 + (lane >> 4), slot = lane & 15, cc = slot ^ ((kk & 3) << 2);
            if constexpr (AKM) {
                kcA[j] = kk;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const long col = (long)m0 + half * 128 + cc * 8;
                    voA[half][j] = (col + 8 <= lda) ? (unsigned)(((long)kk * lda + col) * 2) : OOB;
                }
            }
            if constexpr (BKM) {
                kcB[j] = kk;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const long col = (long)n0 + half * 128 + cc * 8;
                    voB[half][j] = (col + 8 <= ldb) ? (unsigned)(((long)kk * ldb + col) * 2) : OOB;
                }
            }
        }
    }
    auto issue1 = [&](int t, int p, int j) {
        char* dst = smem + ((4 * t + p) % 10) * SLAB;
        const int half = p >> 1;
        const int krem = Kseg - t * BK;            // K elements of this K-step inside the range (<= 0: the whole step reads zeros)
        lds_void* d = (lds_void*)(dst + (PPW * wave + j) * 1024);
        if (p & 1) {
            const unsigned vo = (kcB[j] < krem) ? voB[half][j] : OOB;
            const unsigned so = BKM ? (unsigned)t * (unsigned)(BK * 2) * (unsigned)ldb : (unsigned)t * (BK * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, d, 16, vo, so, 0, DMA_AUX_B);
        } else {
            const unsigned vo = (kcA[j] < krem) ? voA[half][j] : OOB;
            const unsigned so = AKM ? (unsigned)t * (unsigned)(BK * 2) * (unsigned)lda : (unsigned)t * (BK * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, d, 16, vo, so, 0, DMA_AUX_A);
        }
    };
    auto issue = [&](int t, int p) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) issue1(t, p, j);
    };
    issue(t0, 0); issue(t0, 1); issue(t0, 2); issue(t0, 3);
    issue(t0 + 1, 0); issue(t0 + 1, 1); issue(t0 + 1, 2); issue(t0 + 1, 3);

    const unsigned rowoff = (unsigned)(lane & 31) * 128u;
    const unsigned f = ((unsigned)(lane & 31) >> 1) & 7u;
    unsigned xo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xo[ks] = rowoff + ((((unsigned)(ks * 2 + (lane >> 5))) ^ f) << 4);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    bf16x8 fa[2][TM], fb[2][TN];
    // K-major operands: transposing reads, address = slab + lane part + (blk ^ j) * 64 [+ immediates ks*4096, half*1024]
    const unsigned kj = (unsigned)(lane & 15) >> 2;
    const unsigned klane = kj * 256u + (((unsigned)lane >> 4) & 1u) * 32u + ((unsigned)lane & 3u) * 8u + ((unsigned)lane >> 5) * 2048u;
    unsigned kxa[TM], kxb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) kxa[i] = klane + ((((unsigned)i) ^ kj) << 6);
#pragma unroll
    for (int i = 0; i < TN; ++i) kxb[i] = klane + ((((unsigned)((wn & 1) * 2 + i)) ^ kj) << 6);

    // K loop.  The K-step barrier sits in front of the LAST MFMA cluster of a step, not after it: by then every wave has issued (and,
    // with the lgkmcnt(0) in front of it, received) all fragment reads of step t, so (a) the slabs of step t are free for refill at
    // once, (b) the data of step t + 1 -- requested a full K-step earlier -- is visible, and (c) the first fragments of step t + 1
    // are read in the shadows of cluster 3, so no wave ever waits on LDS latency or barrier skew with an empty MFMA pipe.
    // Ring state in front of cluster 3 of step t: step t + 1 complete (4 slabs), step t + 2 parts 0,1 in flight (2), step t's 4
    // slabs released -> refilled with (t+2: 2,3) in cluster 3 and (t+3: 0,1) in the next cluster 0.
    auto slab_base = [&](int t, unsigned& a_base, unsigned& b_slab, unsigned& b_base) {
        a_base = lds0 + (unsigned)((4 * t + 2 * wm) % 10) * SLAB;                                  // A half wm
        b_slab = lds0 + (unsigned)((4 * t + 1 + 2 * (wn >> 1)) % 10) * SLAB;                     // B half wn >> 1
        b_base = b_slab + (unsigned)(wn & 1) * 8192u;
    };
    // memory instruction `op` of the fragment set of k-chunk ks (6 row-major fragments, or up to 12 transposing half reads)
    constexpr int NOPS = (AKM ? 2 * TM : TM) + (BKM ? 2 * TN : TN);
    auto frag_op = [&](int op, int ks, int buf, unsigned a_base, unsigned b_slab, unsigned b_base) {
        constexpr int NB_OPS = BKM ? 2 * TN : TN;
        if (op < NB_OPS) {
            if constexpr (BKM) {
                bf16x8& dst = fb[buf][op >> 1];
                const unsigned addr = b_slab + kxb[op >> 1];
                const int h = op & 1;
                if (ks == 0) { if (h) read_half_km_asm<0, 1>(dst, addr); else read_half_km_asm<0, 0>(dst, addr); }
                else if (ks == 1) { if (h) read_half_km_asm<1, 1>(dst, addr); else read_half_km_asm<1, 0>(dst, addr); }
                else if (ks == 2) { if (h) read_half_km_asm<2, 1>(dst, addr); else read_half_km_asm<2, 0>(dst, addr); }
                else { if (h) read_half_km_asm<3, 1>(dst, addr); else read_half_km_asm<3, 0>(dst, addr); }
            } else {
                if (op == 0) lds_read_b128<0>(fb[buf][0], b_base + xo[ks]);
                else lds_read_b128<4096>(fb[buf][1], b_base + xo[ks]);
            }
        } else {
            const int o = op - NB_OPS;
            if constexpr (AKM) {
                bf16x8& dst = fa[buf][o >> 1];
                const unsigned addr = a_base + kxa[o >> 1];
                const int h = o & 1;
                if (ks == 0) { if (h) read_half_km_asm<0, 1>(dst, addr); else read_half_km_asm<0, 0>(dst, addr); }
                else if (ks == 1) { if (h) read_half_km_asm<1, 1>(dst, addr); else read_half_km_asm<1, 0>(dst, addr); }
                else if (ks == 2) { if (h) read_half_km_asm<2, 1>(dst, addr); else read_half_km_asm<2, 0>(dst, addr); }
                else { if (h) read_half_km_asm<3, 1>(dst, addr); else read_half_km_asm<3, 0>(dst, addr); }
            } else {
                if (o == 0) lds_read_b128<0>(fa[buf][0], a_base + xo[ks]);
                else if (o == 1) lds_read_b128<4096>(fa[buf][1], a_base + xo[ks]);
                else if (o == 2) lds_read_b128<8192>(fa[buf][2], a_base + xo[ks]);
                else lds_read_b128<12288>(fa[buf][3], a_base + xo[ks]);
            }
        }
    };
    // ops spread over the 8 MFMA shadows of a cluster: two per slot first if there are more than 8, then one per slot
    auto slot_ops = [&](int i, int ks, int buf, unsigned a_base, unsigned b_slab, unsigned b_base) {
        constexpr int DBL = NOPS > 8 ? NOPS - 8 : 0;     // slots that carry two ops
        if (i < DBL) { frag_op(2 * i, ks, buf, a_base, b_slab, b_base); frag_op(2 * i + 1, ks, buf, a_base, b_slab, b_base); }
        else if (DBL + i < NOPS) frag_op(DBL + i, ks, buf, a_base, b_slab, b_base);
    };

    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // step t0 landed (this wave's pieces); step t0 + 1 may be in flight
    __builtin_amdgcn_s_barrier();
    {
        unsigned a0, bs0, bb0;
        slab_base(t0, a0, bs0, bb0);
#pragma unroll
        for (int op = 0; op < NOPS; ++op) frag_op(op, 0, 0, a0, bs0, bb0);
    }
    for (int t = t0; t < t1; ++t) {
        unsigned a_base, b_slab, b_base, a_next, bs_next, bb_next;
        slab_base(t, a_base, b_slab, b_base);
        slab_base(t + 1, a_next, bs_next, bb_next);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cb = ks & 1, nb = cb ^ 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // fragments of k-chunk ks (requested a cluster ago)
            if (ks == 3) {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // all but step t+2's parts 0,1 (this wave's 4 newest pieces) landed
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TN * TM; ++i) {
                const int tn = i / TM, tm = i % TM;
                acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cb][tn], fa[cb][tm], acc[tn][tm], 0, 0, 0);
                if (ks < 3) slot_ops(i, ks + 1, nb, a_base, b_slab, b_base);
                else slot_ops(i, 0, nb, a_next, bs_next, bb_next);            // first fragments of step t + 1, behind the barrier
                // refill: (t+2: 2,3) right behind the barrier that released the slabs of step t (needed one step later: a full K-step
                // of lead); (t+2: 0,1), which go to slabs released one barrier earlier, two pieces each in clusters 1 and 2
                if (i >= 4 && ks == 3) { const int q = i - 4; issue1(t + 2, 2 + (q >> 1), q & 1); }
                if (i >= 6 && ks == 1) issue1(t + 2, 0, i - 6);
                if (i >= 6 && ks == 2) issue1(t + 2, 1, i - 6);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // retire the trailing DMA pieces and the last cluster's fragment reads ...
    __syncthreads();                                   // ... and every wave's fragment reads: the LDS becomes epilogue scratch

    if (bid >= full) {
        // one of the S K-parts of a remainder tile: publish the partial, take a ticket, the last arriver sums in part order
        const int rt = tile_id - full;
        float* slabs = sk_slabs + (size_t)rt * S * SK_SLAB_FLOATS;
        sk_store<TM, TN>(acc, slabs + (size_t)part * SK_SLAB_FLOATS, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int ticket = __hip_atomic_fetch_add(sk_cnt + rt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == (unsigned)(S - 1)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(sk_cnt + rt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *reinterpret_cast<volatile unsigned int*>(smem) = ticket;
        }
        __syncthreads();
        const unsigned int ticket = *reinterpret_cast<volatile unsigned int*>(smem);
        if (__builtin_amdgcn_readfirstlane(ticket) != (unsigned)(S - 1)) return;
        __syncthreads();                               // the ticket word lies in wave 0's epilogue strip
        if (S == 2) {
            sk_add<TM, TN>(acc, slabs + (size_t)(part ^ 1) * SK_SLAB_FLOATS, tid);      // a + b == b + a
        } else {                                   // fixed association ((p0 + p1) + p2) + ... whoever is last: re-read every slab
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            for (int p = 0; p < S; ++p) sk_add<TM, TN>(acc, slabs + (size_t)p * SK_SLAB_FLOATS, tid);
        }
    }
    gemm_epilogue_lds<TM, TN, SWIGLU>(acc, smem + wave * EPI_STRIP, C, M, N, ldc, bias, res, ldr, flags, m0 + wm * 128, n0 + wn * 64, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// 256x256 "ring16" kernel (round 3): the ring kernel's data path -- ten 16-KiB LDS slabs, buffer_load ... lds DMA with counted waits,
// XCD-grouped tiles, K-split remainder round, LDS-transposed epilogue -- under the MFMA shape and wave layout the matrix pipe runs
// coolest at: 4 waves x (128 x 128) of v_mfma_f32_16x16x32_bf16, accumulators pinned in 256 AGPRs, ONE wave per SIMD.
// Why (profiles/r03_mfma_shape_probe.md): on random operands the chip is power limited, and with no global traffic at all the loop of
// 32x32x16 MFMAs tops out at 1.75 PF in either wave layout while 16x16x32 reaches 1.89 PF fed from LDS at 128 x 128 per wave (2.0 PF
// from registers): K = 32 per instruction halves the accumulator traffic per FLOP, and the 128 x 128 wave tile needs one third fewer
// LDS fragment bytes per FLOP than 128 x 64 (which only pays once the MFMA itself is cheaper).  It is also the shape the vendor
// library's own gfx950 kernels use.
//
// Per K-step (64 k) a wave issues 128 MFMAs in two halves (k 0-31, 32-63) of 8 x 8 blocks, 16 fragment reads per half (ds_read_b128
// from the row-major slab image: lane (r = lane & 15, kg = lane >> 4) owns row r, k = 8 kg .. 8 kg + 7 -- the same XOR swizzle is
// conflict-free for this access too; K-major operands: two ds_read_b64_tr_b16 per fragment) and 16 DMA pieces, every one of them in the
// shadow of an MFMA.  Operands swapped as in the 8-wave kernel (a = B rows, b = A rows): a lane owns ONE output row and 4 consecutive
// columns per 16 x 16 block.  Schedule of step t:
//   half 0: MFMAs on fragments (t, k-half 0); shadows: read fragments (t, k-half 1), issue DMA (t+2: parts 0,1)
//   lgkmcnt(0), vmcnt(8), s_barrier: every wave has read all of step t's slabs (free for refill) and step t+1 has landed everywhere
//   half 1: MFMAs on fragments (t, k-half 1); shadows: read fragments (t+1, k-half 0), issue DMA (t+2: parts 2,3)
// i.e. one barrier per K-step and a full K-step of lead for every slab, as before.
template <int NBN, int NBM, int NT>
__device__ __forceinline__ void sk_store16(const f32x4 (&acc)[NBN][NBM], float* __restrict__ slab, int tid) {
#pragma unroll
    for (int i = 0; i < NBN; ++i)
#pragma unroll
        for (int j = 0; j < NBM; ++j) reinterpret_cast<f32x4*>(slab)[(i * NBM + j) * NT + tid] = acc[i][j];      // (nt stores: measured equal, round 5)
}
// acc += slab, eight 16-B loads in flight per lane.  The accumulators live in AGPRs and the compiler's scheduler, minimising register
// pressure, turned `acc += slab[...]` into load -> s_waitcnt vmcnt(0) -> add, one memory round trip per 16 bytes: 32 - 64 round trips per
// slab, 27 us (S = 2) to 108 us (S = 8) per last arriver (profiles/r04_gemm_anatomy.md).  The loads are therefore issued by hand (asm
// volatile keeps them together) and one wait, which names the destinations, stands between them and the adds.
__device__ __forceinline__ void sk_load8(f32x4 (&v)[8], const f32x4* __restrict__ p, int stride) {
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(p + (long)j * stride) : "memory");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 :
                 : "memory");
}
template <int NBN, int NBM, int NT>
__device__ __forceinline__ void sk_add16(f32x4 (&acc)[NBN][NBM], const float* __restrict__ slab, int tid) {
    static_assert(NBM == 8, "one batch = the eight M blocks of an N block");
#pragma unroll
    for (int i = 0; i < NBN; ++i) {
        f32x4 v[8];
        sk_load8(v, reinterpret_cast<const f32x4*>(slab) + (long)(i * NBM) * NT + tid, NT);
#pragma unroll
        for (int j = 0; j < NBM; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] += v[j][e];
    }
}

// NW = 4: 2 x 2 waves of 128 x 128 (one wave per SIMD, the coolest loop: variant 13); NW = 8: 2 x 4 waves of 128 x 64 (two waves per
// SIMD: the 8-wave kernel's latency hiding and epilogue width under the 16x16x32 shape: variant 14)
// PAIR (forward epilogue fusions that need TWO output columns in one lane; NT only): every 256-column tile is built from 128 "features"
// phi, each a pair of output columns (first(phi), first(phi) + pair_dist); the B rows are fetched so that a wave's columns are
// [its features' first columns | the same features' second columns], and the epilogue meets both in one lane through two strips.
//   PAIR_SWIGLU (1): B = [gate | up] rows, N = 2 I; first(phi) = tile*128 + phi, pair_dist = I: C = [gate | up] as the unfused GEMM writes
//                    it AND aux0 = silu(gate) * up [M, I] (row stride aux_ld) -- swiglu_fwd's arithmetic, the 322 MB re-read gone
//   PAIR_ROPE   (2): q|k|v projection, heads of 128: first(phi) = n0 + (phi >> 6)*128 + (phi & 63), pair_dist = 64: columns below aux_n get
//                    the rotary embedding (aux0 = cos, aux1 = sin, bf16 [M, 64], row stride aux_ld) with rope_apply's bf16 rounding
//                    sequence (modeling_llama.py:157-158) before they are stored; columns from aux_n on (v) are stored as they are
#ifdef RING16_STAMPS
// timing probe (tools/gemm_anatomy.py; never in the product build): per workgroup {XCC id, HW id, unit info, s_memtime at entry / after the
// prologue's DMA issue / at the first MFMA / at the end of the K loop / at exit}
__device__ unsigned long long g_ring16_stamps[8192 * 8];
#define STAMP(i) do { if (threadIdx.x == 0) stamp_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
// ---- Balanced remainder round (round 5; finishing-kernel mode only).  The tiles of an incomplete last round used to be split into S EQUAL
// K parts with S * rem <= #CU: 352 tiles on 256 CUs = 96 remainder tiles x 2 parts -- 192 units of K/2 while 64 CUs idle; every CU that works
// walks 0.5 K where 96 / 256 = 0.375 K would do.  Now the remainder's K-steps are one sequence of T = rem * nk steps cut into `units` (= #CU)
// equal ranges; a range that crosses a tile boundary is TWO workgroups (the tail of one tile, the head of the next), so no workgroup ever
// spans tiles (the K loop, its DMA prologue and the slab store stay as they are).  Grid order: first the `units` workgroups that start each
// range (they start together; workgroup j = range j runs on XCD j % 8), then the second segments -- each on the XCD of its range's first
// segment (the hardware deals workgroup j to XCD j % 8 and hands it to the first CU of that XCD that frees up: a tail placed on another XCD
// waits behind that XCD's long first segments -- measured: +3 % on the family instead of -2 %), per XCD the one whose first segment is SHORTEST
// first, so a CU that finishes a short head picks up the longest remaining tail and every CU ends after ~T / units steps.  The tail region of
// the grid is therefore 8 x R slots (R = the largest number of crossing ranges on one XCD); slots without a tail hold SK_EMPTY and exit at
// once.  (Placement and order are assumptions for speed only: any dispatch order gives the same result.)  Slabs: first segment of range c ->
// slab c; the segment that BEGINS tile t (t >= 1) from a range started in tile t - 1 -> slab units + t - 1 (at most one per tile boundary).
// The finishing kernel adds a tile's parts in K order.  Boundaries closer than SK_MINSEG steps to a tile boundary snap onto it (no 1-step
// segments).
#define SK_MINSEG 4
#define SK_BALANCED_MIN_STEPS 16            // ranges shorter than this: the equal split (its parts are >= 8 steps)
#define SK_BALANCED_MIN_GAIN 24             // K-steps per CU the balanced round must save over the equal split to be chosen
#define SK_EMPTY 0xFFFFu
struct SkPlan {
    int units;                              // 0: equal split (S parts per tile, slab rt * S + part)
    int nspan;                              // slots of the tail region (a multiple of 8)
    unsigned short span[256];               // slot i (XCD i % 8): the range whose second segment runs there, or SK_EMPTY
};
// first step of range c (0 .. units) in the remainder's step sequence; T = rem * nk
// (32-bit on purpose: c <= 256 and T < 2^23, see sk_make_plan -- a 64-bit division is ~300 scalar instructions in a workgroup's prologue)
__host__ __device__ __forceinline__ int sk_bound(int c, int units, int T, int nk) {
    unsigned a = (unsigned)c * (unsigned)T / (unsigned)units;
    const unsigned r = a % (unsigned)nk;
    if (r < SK_MINSEG) a -= r;
    else if ((unsigned)nk - r < SK_MINSEG) a += (unsigned)nk - r;
    return (int)a;
}

// Two problems in ONE grid (round 6b): two weight-gradient GEMMs of a decoder layer that share K (the token count) and are independent of each
// other -- dW(down_proj) 4096 x 14336 = 896 tiles (3.5 rounds on 256 CUs) and dW(q|k|v) 6144 x 4096 = 384 tiles (1.5 rounds), each with a K-split
// remainder round and a finishing pass -- are 1280 tiles = 5.0 rounds together: whole tiles only, no slabs, no finishing kernels.  The grid's tile
// order is the XCD-contiguous one over the union; a workgroup whose tile lies behind the first problem's takes the second problem's operands
// (a wave-uniform switch of kernel arguments in front of everything else).  RingGroup2 carries the second problem; the plain kernels take the
// empty NoGroup.
struct NoGroup {};
struct RingGroup2 {
    const bf16_t* A; const bf16_t* B; bf16_t* C; bf16_t* aux0;
    long lda, ldb, ldc;
    int M, N, tiles_m, tiles_n, tiles1;      // tiles1: tiles of the first problem
};
template <int NW, bool AKM, bool BKM, bool SWIGLU = false, int PAIR = PAIR_NONE, class GRP = NoGroup>
__global__ __launch_bounds__(NW * 64) void gemm_nt_ring16_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M, int N, int K, long lda, long ldb,
    long ldc, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res, long ldr, int flags, int tiles_m, int tiles_n,
    int full, int S, float* __restrict__ sk_slabs, unsigned int* __restrict__ sk_cnt, bf16_t* __restrict__ aux0,
    const bf16_t* __restrict__ aux1, long aux_ld, int aux_n, SkPlan plan, GRP grp) {
    static_assert(NW == 4 || NW == 8, "4 waves of 128 x 128 or 8 waves of 128 x 64");
    static_assert(PAIR == PAIR_NONE || (!AKM && !BKM && !SWIGLU), "the pair epilogues are forward (NT) fusions");
#ifdef RING16_STAMPS
    unsigned long long stamp_[5] = {0, 0, 0, 0, 0};
    struct StampOut {
        unsigned long long* s; int bid;
        __device__ ~StampOut() {
            if (threadIdx.x == 0 && bid < 8192) {
                unsigned xcc, hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                unsigned long long* o = g_ring16_stamps + (size_t)bid * 8;
                o[0] = xcc; o[1] = hw; o[2] = s[0]; o[3] = s[1]; o[4] = s[2]; o[5] = s[3]; o[6] = __builtin_amdgcn_s_memtime(); o[7] = s[4];
            }
        }
    } stamp_out_{stamp_, (int)blockIdx.x};
    STAMP(0);
#endif
    constexpr int NBM = 8, NBN = NW == 4 ? 8 : 4;          // 16 x 16 blocks per wave along M / N
    constexpr int SLAB = 16384, RING = 10 * SLAB, PPW = 16 / NW, NMF = NBN * NBM;
    __shared__ __attribute__((aligned(16))) char smem[RING];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NW == 4 ? wave >> 1 : wave >> 2;        // A rows wm*128 .. +127  (slab part 2 wm)
    const int wn = NW == 4 ? wave & 1 : wave & 3;          // B rows wn*(NBN*16) ..  (slab part 1 + 2 wnh, block offset wno)
    const int wnh = NW == 4 ? wn : wn >> 1, wno = NW == 4 ? 0 : (wn & 1) * 4;

    // workgroup -> unit, exactly as in the 8-wave kernel (XCD-contiguous tile ranges, K parts of the remainder tiles adjacent)
    const int nk = (K + BK - 1) / BK;
    const int bid = blockIdx.x;
    int tile_id, part = 0;
    if (bid < full) {
        const int q = full >> 3, r = full & 7, xcd = bid & 7;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    } else if (plan.units == 0) {
        const int j = bid - full, nu = gridDim.x - full, rem = nu / S;
        const int q = nu >> 3, r = nu & 7, xcd = j & 7;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (j >> 3);
        part = lin / rem;
        tile_id = full + lin - part * rem;
    } else {
        tile_id = full;                      // balanced remainder round: set below
    }
    // 32-bit on purpose (nk <= 2^15, part < S <= 8): the 64-bit divisions cost the K-split units ~300 scalar instructions before their first DMA
    int t0 = (bid < full) ? 0 : (int)((unsigned)nk * (unsigned)part / (unsigned)S);
    int t1 = (bid < full) ? nk : (int)((unsigned)nk * (unsigned)(part + 1) / (unsigned)S);
    int sk_slab = 0;                         // balanced plan: this workgroup's slab
    if (bid >= full && plan.units != 0) {
        // see SkPlan: workgroups [0, units) start the ranges, the rest are the second segments of the ranges that cross a tile boundary
        const int j = bid - full, U = plan.units, rem = S, T = rem * nk;      // (S carries the number of remainder tiles in this mode)
        const bool first = j < U;
        int c;
        if (first) {
            c = j;                                   // range j on XCD j % 8: ranges of one K phase (period 8 for 96 tiles on 256 CUs) share panels
        } else {
            c = plan.span[j - U];
            if (c == (int)SK_EMPTY) return;         // a tail slot of an XCD with fewer crossing ranges than the fullest one
        }
        const int a0 = sk_bound(c, U, T, nk), a1 = sk_bound(c + 1, U, T, nk);
        int tl = a0 / nk;
        if (first) {
            const int e = (tl + 1) * nk;
            t0 = a0 - tl * nk;
            t1 = (a1 < e ? a1 : e) - tl * nk;
            sk_slab = c;
        } else {
            tl += 1;
            t0 = 0;
            t1 = a1 - tl * nk;
            sk_slab = U + tl - 1;
        }
        tile_id = full + tl;
    }
    if constexpr (std::is_same<GRP, RingGroup2>::value) {
        if (tile_id >= grp.tiles1) {         // the second problem of a grouped launch (whole tiles only: bid < full always)
            tile_id -= grp.tiles1;
            A = grp.A; B = grp.B; C = grp.C; aux0 = grp.aux0;
            lda = grp.lda; ldb = grp.ldb; ldc = grp.ldc;
            M = grp.M; N = grp.N; tiles_m = grp.tiles_m; tiles_n = grp.tiles_n;
        }
    }
    const int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int g = tile_id / per_group;
    const int first_m = g * GROUP;
    const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int in_g = tile_id - g * per_group;
    const int m0 = (first_m + in_g % gsz) * 256, n0 = (in_g / gsz) * 256;
    const int Kseg = (t1 * BK < K) ? t1 * BK : K;
    // pair modes: first output column of feature phi (0 .. 127) of this tile, and the distance to its partner column
    const int pair_dist = PAIR == PAIR_SWIGLU ? (N >> 1) : 64;
    auto pair_first = [&](int phi) { return PAIR == PAIR_SWIGLU ? (n0 >> 1) + phi : n0 + (phi >> 6) * 128 + (phi & 63); };

    f32x4 acc[NBN][NBM];       // acc[tn][tm]: block (n block tn, m block tm); zeroed BEHIND the prologue's DMA issue (below)

    // DMA pieces: slab (t, p) = 16 pieces of 1 KiB, pieces PPW*w .. PPW*w + PPW - 1 belong to wave w (descriptor + fixed lane offset +
    // scalar K-step offset)
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)A, 0, (int)(unsigned)(((long)(AKM ? K : M) * lda) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)B, 0, (int)(unsigned)(((long)(BKM ? K : N) * ldb) * 2), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned voA[2][PPW], voB[2][PPW];
    int kcA[PPW], kcB[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int piece = PPW * wave + j;
        {   // row-major piece: 8 rows x 128 B, 16-B chunk swizzled with the row pair (slot = chunk ^ ((row >> 1) & 7))
            const int rl = lane >> 3, fz = (piece * 4 + (rl >> 1)) & 7, chunk = (lane & 7) ^ fz;
            if constexpr (!AKM) {
                kcA[j] = chunk * 8;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    int grow = m0 + half * 128 + piece * 8 + rl;
                    grow = grow < M ? grow : M - 1;
                    voA[half][j] = (unsigned)(((long)grow * lda + chunk * 8) * 2);
                }
            }
            if constexpr (!BKM) {
                kcB[j] = chunk * 8;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int r = piece * 8 + rl;                  // row of this half of the tile's B operand, 0 .. 127
                    int grow = n0 + half * 128 + r;
                    if constexpr (PAIR != PAIR_NONE) {
                        // a wave's 2 G rows = [G first columns | G second columns] of its G features (G = 64 / 32 for 4 / 8 waves)
                        constexpr int G = NBN * 8;
                        const int phi = half * 64 + (r / (2 * G)) * G + (r % G), second = (r / G) & 1;
                        grow = pair_first(phi) + second * pair_dist;
                    }
                    grow = grow < N ? grow : N - 1;
                    voB[half][j] = (unsigned)(((long)grow * ldb + chunk * 8) * 2);
                }
            }
        }
        {   // K-major piece: 4 k-rows x 256 B; LDS slot s of k-row kk holds the 16-B chunk s ^ ((kk & 3) << 2) ^ (((kk >> 3) & 1) << 1):
            // the 64-B segment index is XORed with kk & 3 (the four k-rows of one transposing read land on the four quarters of a
            // bank row) and the 32-B half with bit 3 of kk (the two k-groups that share an LDS cycle land on different halves)
            const int kk = piece * 4 + (lane >> 4), slot = lane & 15, cc = slot ^ ((kk & 3) << 2) ^ (((kk >> 3) & 1) << 1);
            if constexpr (AKM) {
                kcA[j] = kk;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const long col = (long)m0 + half * 128 + cc * 8;
                    voA[half][j] = (col + 8 <= lda) ? (unsigned)(((long)kk * lda + col) * 2) : OOB;
                }
            }
            if constexpr (BKM) {
                kcB[j] = kk;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const long col = (long)n0 + half * 128 + cc * 8;
                    voB[half][j] = (col + 8 <= ldb) ? (unsigned)(((long)kk * ldb + col) * 2) : OOB;
                }
            }
        }
    }
    // ring arithmetic on byte offsets, scalar and incremental (no division in the loop): slab (t, p) lives at wrap(base(t) + p*SLAB),
    // base(t) = ((4 t) mod 10) * SLAB relative to this unit's first step (any consistent mapping will do: the ring is private)
    auto wrap = [](unsigned x) {
        x = x >= (unsigned)RING ? x - RING : x;
        return x >= (unsigned)RING ? x - RING : x;
    };
    const unsigned soA1 = AKM ? (unsigned)(BK * 2) * (unsigned)lda : (unsigned)(BK * 2);      // K-step advance of the scalar offset
    const unsigned soB1 = BKM ? (unsigned)(BK * 2) * (unsigned)ldb : (unsigned)(BK * 2);
    const unsigned wave_dst = (unsigned)(PPW * wave) * 1024u;
    // one DMA piece: part P (compile-time) of the step whose slab offset is `doff`, K-step scalar offsets soa / sob, `krem` k left
    auto issue1 = [&](auto pc, int j, unsigned doff, unsigned soa, unsigned sob, int krem) {
        constexpr int p = decltype(pc)::value, half = p >> 1;
        lds_void* d = (lds_void*)(smem + doff + wave_dst + j * 1024);
        if constexpr (p & 1) {
            const unsigned vo = (kcB[j] < krem) ? voB[half][j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, d, 16, vo, sob, 0, DMA_AUX_B);
        } else {
            const unsigned vo = (kcA[j] < krem) ? voA[half][j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, d, 16, vo, soa, 0, DMA_AUX_A);
        }
    };
    {   // prologue: steps t0 (slabs 0-3) and t0 + 1 (slabs 4-7), all four parts each
        const unsigned a0 = (unsigned)t0 * soA1, b0 = (unsigned)t0 * soB1;
        const int kr = Kseg - t0 * BK;
        static_for<0, 4>([&](auto pc) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) issue1(pc, j, decltype(pc)::value * SLAB, a0, b0, kr);
        });
        static_for<0, 4>([&](auto pc) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) issue1(pc, j, (4 + decltype(pc)::value) * SLAB, a0 + soA1, b0 + soB1, kr - BK);
        });
    }

    STAMP(1);
#ifndef RING16_ZERO_FIRST
    // the accumulators are cleared while the first K-steps are in flight (128 / 256 v_accvgpr_write: in front of the DMA issue they delayed the
    // first byte by as many issue slots; -DRING16_ZERO_FIRST: the round-4 order, for A/B builds)
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int i = 0; i < NBN; ++i)
#pragma unroll
        for (int j = 0; j < NBM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment addresses.  Row-major image: lane (r, kg) reads 16 B of row blk*16 + r at chunk kh*4 + kg, swizzled with (r >> 1) & 7
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned r16 = (unsigned)lane & 15u, kg = (unsigned)lane >> 4;
    unsigned xo[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) xo[kh] = r16 * 128u + ((((unsigned)(kh * 4) + kg) ^ ((r16 >> 1) & 7u)) << 4);
    // K-major image [64 k][128 rows]: lane s = lane & 15 of k-group kg addresses k-row kh*32 + kg*8 + h*4 + (s >> 2), 8 B at row
    // blk*16 + (s & 3)*4: + (((blk >> 1) ^ kj) << 6) + (((blk & 1) ^ (kg & 1)) << 5); immediates kh*8192 + h*1024
    const unsigned kj = r16 >> 2;
    const unsigned klane = kg * 2048u + kj * 256u + (r16 & 3u) * 8u;
    unsigned kxa[AKM ? NBM : 1], kxb[BKM ? NBN : 1];
    if constexpr (AKM) {
#pragma unroll
        for (int i = 0; i < NBM; ++i) kxa[i] = klane + ((((unsigned)(i >> 1)) ^ kj) << 6) + ((((unsigned)(i & 1)) ^ (kg & 1u)) << 5);
    }
    if constexpr (BKM) {
#pragma unroll
        for (int i = 0; i < NBN; ++i) {
            const unsigned blk = (unsigned)(wno + i);
            kxb[i] = klane + (((blk >> 1) ^ kj) << 6) + (((blk & 1u) ^ (kg & 1u)) << 5);
        }
    }
    const unsigned b_row_off = BKM ? 0u : (unsigned)wno * 2048u;      // row-major image: the wave's first B block inside the slab

    bf16x8 fa[2][NBM], fb[2][NBN];
    // memory op OP of the fragment set (step slabs a_base / b_base, k-half KH) into buffer BUF: B fragments first.  Every index is a
    // template-level constant (static_for / integral_constant): the accumulator and fragment arrays must never see a dynamic index, or
    // they leave the register file -- a pragma-unrolled loop gave that guarantee only up to the unroller's size heuristics
    constexpr int OPS_B = BKM ? 2 * NBN : NBN, OPS_A = AKM ? 2 * NBM : NBM, NOPS = OPS_A + OPS_B;
    auto frag_op = [&](auto opc, auto khc, auto bufc, unsigned a_base, unsigned b_base) {
        constexpr int op = decltype(opc)::value, kh = decltype(khc)::value, buf = decltype(bufc)::value;
        if constexpr (op < OPS_B) {
            if constexpr (BKM) lds_read_tr64_h<kh * 8192>(fb[buf][op >> 1], b_base + kxb[op >> 1], op & 1);
            else lds_read_b128_v<op * 2048>(fb[buf][op], b_base + xo[kh]);
        } else {
            constexpr int o = op - OPS_B;
            if constexpr (AKM) lds_read_tr64_h<kh * 8192>(fa[buf][o >> 1], a_base + kxa[o >> 1], o & 1);
            else lds_read_b128_v<o * 2048>(fa[buf][o], a_base + xo[kh]);
        }
    };
    // one half of a K-step: NMF MFMAs on buffer BUF; the next fragment set (slabs na / nb_, k-half NKH) is read into the other buffer in
    // the shadows of the first three quarters of them (op o rides in shadow o * SL / NOPS), and 2 * PPW DMA pieces -- parts P0, P0 + 1 of
    // the step two ahead, slab offsets d0 / d1 -- in the shadows 5, 12, 19, ...
    constexpr int SL = NMF * 3 / 4;
    auto half_step = [&](auto bufc, auto nkhc, auto p0c, unsigned na, unsigned nb_, unsigned d0, unsigned d1, unsigned soa, unsigned sob,
                         int krem) {
        constexpr int buf = decltype(bufc)::value, p0 = decltype(p0c)::value;
        static_for<0, NMF>([&](auto ic) {
            constexpr int i = decltype(ic)::value, tn = i / NBM, tm = i % NBM;
            mfma16(acc[tn][tm], fb[buf][tn], fa[buf][tm]);
#ifndef RING16_PROBE_NO_FRAGS
            if constexpr (i < SL) {
                constexpr int o0 = (i * NOPS + SL - 1) / SL, o1 = ((i + 1) * NOPS + SL - 1) / SL;      // ops with o * SL / NOPS == i
                static_for<o0, (o1 < NOPS ? o1 : NOPS)>([&](auto oc) { frag_op(oc, nkhc, ic_<buf ^ 1>{}, na, nb_); });
            }
#endif
#ifndef RING16_PROBE_NO_DMA              // timing probes only (tools/r03 experiments): results are wrong with any of these defined
            if constexpr (i % 7 == 5 && i / 7 < 2 * PPW) {
                constexpr int q = i / 7;
                if constexpr (q < PPW) issue1(ic_<p0>{}, q, d0, soa, sob, krem);
                else issue1(ic_<p0 + 1>{}, q - PPW, d1, soa, sob, krem);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // 8 waves: the second-dispatched half (waves 4-7, each sharing a SIMD with one of 0-3) loses the instruction arbitration on every
    // segment at equal priority; one static s_setprio 1 for it (no per-segment flips).  Round 3, same box, two alternating runs: headline
    // step 284.3 / 284.4 -> 282.8 / 282.1 ms; per shape within +-1.5 % either way (profiles/r03_experiments.md, 16)
    if constexpr (NW == 8) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }
    if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // step t0 landed (this wave's pieces); step t0 + 1 may
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                          // still be in flight
    __builtin_amdgcn_s_barrier();
    STAMP(2);
    const unsigned a_part = (unsigned)(2 * wm) * SLAB, b_part = (unsigned)(1 + 2 * wnh) * SLAB;
    static_for<0, NOPS>([&](auto oc) { frag_op(oc, ic_<0>{}, ic_<0>{}, lds0 + a_part, lds0 + b_part + b_row_off); });
    unsigned base = 0;                                             // slab offset of step t, part 0
    unsigned soa = (unsigned)(t0 + 2) * soA1, sob = (unsigned)(t0 + 2) * soB1;      // scalar offsets of step t + 2
    int krem = Kseg - (t0 + 2) * BK;
    for (int t = t0; t < t1; ++t) {
        const unsigned a_base = lds0 + wrap(base + a_part), b_base = lds0 + wrap(base + b_part) + b_row_off;
        const unsigned a_next = lds0 + wrap(base + 4 * SLAB + a_part), b_next = lds0 + wrap(base + 4 * SLAB + b_part) + b_row_off;
        const unsigned d0 = wrap(base + 8 * SLAB), d1 = wrap(base + 9 * SLAB), d2 = base, d3 = wrap(base + SLAB);   // (t+2: 0..3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // fragments (t, k-half 0)
        __builtin_amdgcn_sched_barrier(0);
        half_step(ic_<0>{}, ic_<1>{}, ic_<0>{}, a_base, b_base, d0, d1, soa, sob, krem);
        // every fragment read of step t has returned and, with this wave's pieces of step t + 1 landed (all but the 2 PPW newest,
        // (t+2: 0,1)), the barrier makes step t + 1 visible and step t's slabs free
        if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
#ifndef RING16_PROBE_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        half_step(ic_<1>{}, ic_<0>{}, ic_<2>{}, a_next, b_next, d2, d3, soa, sob, krem);
        base = wrap(base + 4 * SLAB);
        soa += soA1;
        sob += soB1;
        krem -= BK;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // the last MFMAs' results are read as soon as the epilogue starts
    __syncthreads();                                             // every wave's fragment reads done: the LDS becomes epilogue scratch
    STAMP(3);
#ifdef RING16_STAMPS
    if (threadIdx.x == 0) stamp_[4] = ((unsigned long long)(unsigned)tile_id << 32) | ((unsigned)part << 16) | (unsigned)(bid >= full ? S : 1);
#endif

    if (bid >= full) {
        const int rt = tile_id - full;
        float* slabs = sk_slabs + (size_t)rt * S * SK_SLAB_FLOATS;
        if (plan.units != 0) sk_store16<NBN, NBM, NW * 64>(acc, sk_slabs + (size_t)sk_slab * SK_SLAB_FLOATS, tid);
        else sk_store16<NBN, NBM, NW * 64>(acc, slabs + (size_t)part * SK_SLAB_FLOATS, tid);
        if (sk_cnt == nullptr) {
            // finishing-kernel mode (launch_gemm_ring): no ticket, no reduction here -- gemm_ring16_finish_kernel, launched behind this
            // kernel on the same stream, sums the S slabs of every remainder tile on ALL compute units and runs the epilogue (the kernel
            // boundary publishes the slabs).  A sumsq launch zeroes the tile's slot for the finishing halves' two atomic adds.
            if constexpr (!SWIGLU && PAIR == PAIR_NONE) {
                if ((flags & EPI_SUMSQ) && t0 == 0 && tid == 0) reinterpret_cast<float*>(aux0)[tile_id] = 0.f;      // the segment that begins the tile
            }
            return;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int ticket = __hip_atomic_fetch_add(sk_cnt + rt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == (unsigned)(S - 1)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(sk_cnt + rt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *reinterpret_cast<volatile unsigned int*>(smem) = ticket;
        }
        __syncthreads();
        const unsigned int ticket = *reinterpret_cast<volatile unsigned int*>(smem);
        if (__builtin_amdgcn_readfirstlane(ticket) != (unsigned)(S - 1)) return;
        __syncthreads();
        if (S == 2) {
            sk_add16<NBN, NBM, NW * 64>(acc, slabs + (size_t)(part ^ 1) * SK_SLAB_FLOATS, tid);
        } else {
#pragma unroll
            for (int i = 0; i < NBN; ++i)
#pragma unroll
                for (int j = 0; j < NBM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < S; ++p) sk_add16<NBN, NBM, NW * 64>(acc, slabs + (size_t)p * SK_SLAB_FLOATS, tid);
        }
    }
    ring16_epilogue<NW, NBM, (AKM || BKM), SWIGLU, PAIR, false>(acc, smem, wave, tid, lane, wn, C, M, N, ldc, bias, res, ldr, flags, m0 + wm * 128, n0,
                                                                 tile_id, aux0, aux1, aux_ld, aux_n);
}

// K-split finishing pass of the ring16 kernels (round 5).  Until round 4 the LAST ARRIVER of a remainder tile summed the tile's S slabs and ran
// the epilogue: S x 256 KiB through ONE compute unit's memory pipeline while most of the chip idled -- 38 - 49 us behind the K loop of every
// launch with an incomplete last round (S = 2), 47 - 95 us at S = 8 (profiles/r04_gemm_anatomy.md).  Now the split units only store their slabs and
// this kernel, launched behind the GEMM on the same stream, spreads the reduction over the whole device: workgroup = (remainder tile, 64-row
// half pm of every wave tile, group of `blockDim.x / 64` of the tile's NW waves); a thread takes exactly the accumulator elements thread
// (wave, lane) of the GEMM kernel held for that half -- slab element ((tn * 8 + pm * 4 + tm) * NW * 64 + wave * 64 + lane) -- summed over the
// slabs in the order the last arriver used (S = 2: slab 0 + slab 1; S > 2: 0 + slab 0 + slab 1 + ...: bit-identical results), and runs
// ring16_epilogue on them.  One-wave workgroups for every epilogue but the sum of squares (whose fixed-order reduction wants the tile's NW
// waves in one block).  No tickets, no fences, no spinning: nothing here can deadlock under CU masks or a shared GPU.
template <int NW, bool KM, bool SWIGLU, int PAIR, int WPB>
__global__ __launch_bounds__(WPB * 64) void gemm_ring16_finish_kernel(
    bf16_t* __restrict__ C, int M, int N, long ldc, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res, long ldr, int flags,
    int tiles_m, int tiles_n, int full, int S, const float* __restrict__ sk_slabs, bf16_t* __restrict__ aux0, const bf16_t* __restrict__ aux1,
    long aux_ld, int aux_n, int sk_units, int nk) {
    constexpr int NBN = NW == 4 ? 8 : 4, NT = NW * 64;
    static_assert(WPB == 1 || WPB == NW, "one wave per workgroup, or all waves of the tile (sum of squares)");
    __shared__ __attribute__((aligned(16))) char fin_smem[(PAIR != PAIR_NONE && NW == 4 ? 2 * WPB : WPB) * EPI_STRIP + 64];
    const int tid_l = threadIdx.x, lane = tid_l & 63;
    const int wave_l = __builtin_amdgcn_readfirstlane(tid_l >> 6);
    constexpr int wpb = WPB, groups = NW / WPB;
    const int bid = blockIdx.x;
    const int grp = bid % groups, pm = (bid / groups) & 1, rt = bid / (2 * groups);
    const int wave = grp * wpb + wave_l;
    const int wm = NW == 4 ? wave >> 1 : wave >> 2, wn = NW == 4 ? wave & 1 : wave & 3;
    const int tile_id = full + rt;
    const int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int g = tile_id / per_group;
    const int first_m = g * GROUP;
    const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int in_g = tile_id - g * per_group;
    const int m0 = (first_m + in_g % gsz) * 256, n0 = (in_g / gsz) * 256;
    constexpr int SLAB4 = SK_SLAB_FLOATS / 4;
    // equal split: the tile's S slabs are rt * S + p; balanced plan (sk_units != 0): absolute slab indices, see below
    const f32x4* slab = reinterpret_cast<const f32x4*>(sk_slabs + (sk_units ? (size_t)0 : (size_t)rt * S * SK_SLAB_FLOATS)) + wave * 64 + lane;
    f32x4 acc[NBN][4];
    auto load_part = [&](int p, f32x4 (&v)[NBN][4]) {
#pragma unroll
        for (int i = 0; i < NBN; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = __builtin_nontemporal_load(slab + (size_t)p * SLAB4 + (i * 8 + pm * 4 + j) * NT);
    };
    if (sk_units != 0) {
        // the parts of remainder tile rt in K order (SkPlan): the head left by a range that started in tile rt - 1 (slab units + rt - 1), then
        // the first segments of the ranges that start inside the tile (slab = range index)
        const int rem = S, T = rem * nk, lo = rt * nk, hi = lo + nk;
        int c = (int)((unsigned)lo * (unsigned)sk_units / (unsigned)T);      // lo < T < 2^23, units <= 256
        while (c + 1 <= sk_units && sk_bound(c + 1, sk_units, T, nk) <= lo) ++c;
        while (c > 0 && sk_bound(c, sk_units, T, nk) > lo) --c;
#pragma unroll
        for (int i = 0; i < NBN; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto add_part = [&](int p) {
            f32x4 o[NBN][4];
            load_part(p, o);
#pragma unroll
            for (int i = 0; i < NBN; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += o[i][j];
        };
        if (sk_bound(c, sk_units, T, nk) < lo) {
            add_part(sk_units + rt - 1);
            ++c;
        }
        for (; c < sk_units && sk_bound(c, sk_units, T, nk) < hi; ++c) add_part(c);
    } else if (S == 2) {
        f32x4 o[NBN][4];
        load_part(0, acc);
        load_part(1, o);
#pragma unroll
        for (int i = 0; i < NBN; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += o[i][j];
    } else {
#pragma unroll
        for (int i = 0; i < NBN; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < S; ++p) {
            f32x4 o[NBN][4];
            load_part(p, o);
#pragma unroll
            for (int i = 0; i < NBN; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += o[i][j];
        }
    }
    ring16_epilogue<NW, 4, KM, SWIGLU, PAIR, true>(acc, fin_smem, wave_l, tid_l, lane, wn, C, M, N, ldc, bias, res, ldr, flags,
                                                   m0 + wm * 128 + pm * 64, n0, tile_id, aux0, aux1, aux_ld, aux_n);
}

// split-K workspace (caller-owned, see mantis_gemm_workspace_bytes): [ticket counters: (#CU + 1) u32, padded to 256 B][#CU fp32
// slabs of 256 x 256].  Zero-initialised by the caller ONCE; every launch leaves the counters at zero (the last arriver of a
// tile resets its ticket), so one workspace serves any number of stream-ordered launches.  No allocation, no global state here.
static int g_num_cu[64];
static inline size_t sk_cnt_bytes(int cus) { return (((size_t)(cus + 1) * sizeof(unsigned int)) + 255) / 256 * 256; }
// slabs: #CU for the equal split; the balanced remainder round (SkPlan) needs one per range (<= #CU) + one per crossed tile boundary (< #CU)
static inline size_t sk_ws_bytes(int cus) { return sk_cnt_bytes(cus) + (size_t)(SK_MAX_ROUNDS > 2 ? SK_MAX_ROUNDS : 2) * cus * SK_SLAB_FLOATS * sizeof(float); }

static int num_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (g_num_cu[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_num_cu[dev] = n;
    }
    return g_num_cu[dev];
}

// CU budget of the tile scheduler.  Rounds, the K split of an incomplete last round and the kernel choice are planned for plan_cus(req)
// compute units: all of the device by default; fewer when something else holds CUs for the length of a GEMM -- every RCCL channel is a
// workgroup that cannot share a CU with a ring workgroup (160 KiB of LDS each), so with C channels busy a launch planned for 256 CUs
// runs its "one round" as two.  The budget is a PER-CALL argument (bits 16-27 of `flags`; round 5 -- until round 4 a process-wide
// variable, which let two models / reducers of one process interfere); 0 = the default, MANTIS_GEMM_CUS (an environment constant,
// read once) or the whole device.  The split-K workspace is always sized for the whole device, so any budget fits it.  Results are
// deterministic for a given budget; across budgets the K-split of the remainder tiles differs, i.e. the fp32 summation order of those
// tiles (bf16 results agree to rounding, not bit for bit).
static int plan_cus(int req) {
    const int dev = num_cus();
    if (req <= 0) {
        static int env = -1;              // the process' configuration constant, never written after the first read
        if (env < 0) {
            const char* e = getenv("MANTIS_GEMM_CUS");
            const int v = e ? atoi(e) : 0;
            env = v > 0 ? v : 0;
        }
        req = env;
    }
    if (req == 0) return dev;
    return req < 8 ? 8 : (req > dev ? dev : req);
}
static inline int flags_cus(int flags) { return (flags & EPI_CUS_MASK) >> EPI_CUS_SHIFT; }

// K parts for the tiles of the ring kernel's incomplete last round: minimise K-steps per part + the measured reduction cost
// (slab write, publish, S slab reads by the last arriver ~ 8 + 1.7 S K-step equivalents; sc1 write-through slabs instead of the
// release/acquire pair were measured equal at S = 2 and 12 % slower at S = 8).  S * rem <= #CU: ONE sub-round of split units.  Round 3
// tried letting S * rem exceed the CU count (SK_MAX_ROUNDS = 3: for dX of gate|up, 96 remainder tiles x 448 K-steps, S = 8 fills all 256
// CUs for three short sub-rounds instead of 192 CUs for one long one; the model predicted -15 % on the remainder): measured SLOWER,
// 984 -> 1216 us on the 8-wave kernel, 1027 -> 1481 us on the 4-wave one -- 768 slabs of 256 KiB written and read back cost far more
// than the 8 + 1.7 S model extrapolates.  The code path stays (SK_MAX_ROUNDS), the limit is 1.
static int ring_split(long ntiles, int nk, int cus) {
    const int rem = (int)(ntiles % cus);
    if (!rem) return 1;
    int best = 1;
    double cost = nk + 5.0, cost1 = -1.0;    // cost1: best single-sub-round split
    int best1 = 1;
    static int s_max = -1;                   // MANTIS_GEMM_SPLIT_MAX: cap on S (measurements)
    if (s_max < 0) { const char* e = getenv("MANTIS_GEMM_SPLIT_MAX"); s_max = e && atoi(e) > 0 ? atoi(e) : 8; }
    for (int S = 2; S <= s_max; ++S) {
        if (nk / S < 8) break;
        const int rounds = (S * rem + cus - 1) / cus;
        if (rounds > SK_MAX_ROUNDS) break;
        const double c = rounds * ((double)nk / S + 5.0) + 8.0 + 1.7 * S;
        if (rounds == 1 && (cost1 < 0 || c < cost1)) { cost1 = c; best1 = S; }
        if (c < cost) { cost = c; best = S; }
    }
    if (best > 1 && (best * rem + cus - 1) / cus > 1) {          // a multi-sub-round split must beat the alternatives clearly
        const double alt = cost1 > 0 && cost1 < nk + 5.0 ? cost1 : nk + 5.0;
        if (cost > 0.9 * alt) return cost1 > 0 && cost1 < nk + 5.0 ? best1 : 1;
    }
    return best;
}
// tile variant by a cost model fitted to measurements (profiles/r01_gemm_experiments.md), in microseconds:
//   ring 256x256, 1 workgroup/CU: 1.45 per K-step + 5 K-step equivalents per round (prologue, epilogue, launch)
//   generic 128x128, 2 workgroups/CU: 1.05 per K-step per round of 2 x #CU tiles + 5.2 equivalents
static int gemm_pick_variant(int M, int N, int K, int cus_req = 0) {
    // below two tile rows / columns the cost model decides too (M = 504 label rows of the lm_head used to fall to the 128x128 kernel
    // -- and, on the dX side, to a transposed weight copy: 2.3 ms instead of 0.58 ms); only genuinely small operands skip it
    if (M < 384 || N < 384) return 1;
    const int cus = plan_cus(cus_req), nk = cdiv(K, BK);
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256), t128 = (long)cdiv(M, 128) * cdiv(N, 128);
    const int S = ring_split(t256, nk, cus);
    const long rem = t256 % cus;
    const int sub = rem ? (int)((S * rem + cus - 1) / cus) : 0;
    const double ring = 1.45 * ((double)(t256 / cus) * (nk + 5.0) + (rem ? sub * ((double)nk / S + 5.0) + (S > 1 ? 8.0 + 1.7 * S : 0.0) : 0.0));
    const double gen = 1.05 * (double)((t128 + 2 * cus - 1) / (2 * cus)) * (nk + 5.2);
    return ring <= gen ? 12 : 1;
}

// ring kernel per shape (measured, profiles/r03_gemm_ring_variants.md)
// (1x MI355X, the fifteen GEMMs of a Mantis-8B decoder layer + head, every layout): the 8-wave 16x16x32 kernel (14) is 2-7 % faster than
// the 32x32x16 one (12) on twelve of them and within 3 % on the rest; the 4-wave kernel (13) is faster still -- 4-5 % over 14 -- where the
// main loop is all there is: row-major operands (NT) and a long per-CU K walk (rounds x K-steps), and slower everywhere else (its
// prologue, epilogue and K-split reduction run on half the waves)
static int ring_variant_for(int M, int N, int K, bool akm, bool bkm, int cus_req = 0) {
    if (akm || bkm) return 14;
    const long tiles = (long)cdiv(M, 256) * cdiv(N, 256);
    const long rounds = (tiles + plan_cus(cus_req) - 1) / plan_cus(cus_req);
    return rounds * cdiv(K, BK) >= 400 ? 13 : 14;
}

static bool sk_balanced_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MANTIS_GEMM_SK_BALANCED"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
// the balanced remainder plan of a launch (SkPlan), or units = 0 when the equal split already fills the planned CUs / the ranges would be short
static void sk_make_plan(SkPlan& plan, int rem, int S, int nk, int cus) {
    plan.units = 0;
    plan.nspan = 0;
    const long T = (long)rem * nk;
    if (!sk_balanced_enabled() || S * rem == cus || rem >= cus || cus > 256 || T / cus < SK_BALANCED_MIN_STEPS || T >= (1L << 23)) return;
    // what it buys: nk / S - T / units K-steps per CU; what it costs: a second prologue + slab store on the CUs that take a tail, more
    // slabs for the finishing pass -- measured ~25 - 30 us, i.e. ~20 K-steps (profiles/r05_experiments.md: dX(gate|up), 448 K-steps, -43 us;
    // down_proj forward, 224, -8 us; the 64- and 96-step shapes +8 ... +16 us): only where the saving is clearly larger
    if (nk / S - (int)(T / cus) < SK_BALANCED_MIN_GAIN) return;
    const int U = cus;
    if (U % 8) return;                                          // the tail placement assumes workgroup j of the remainder section runs on XCD j % 8
    int f[8][32], idx[8][32], n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < U; ++c) {
        const int a0 = sk_bound(c, U, (int)T, nk), a1 = sk_bound(c + 1, U, (int)T, nk);
        const int e = (a0 / nk + 1) * nk;
        if (a1 <= a0) return;                                   // a degenerate range: keep the equal split
        if (a1 > e) {
            if (a1 - e > nk) return;                            // a range spanning three tiles cannot happen for rem < units; be safe
            const int x = c & 7;
            int j = n[x]++;                                     // insertion by first-segment length (then range index): <= 32 per XCD
            while (j > 0 && f[x][j - 1] > e - a0) { f[x][j] = f[x][j - 1]; idx[x][j] = idx[x][j - 1]; --j; }
            f[x][j] = e - a0;
            idx[x][j] = c;
        }
    }
    int R = 0;
    for (int x = 0; x < 8; ++x) R = n[x] > R ? n[x] : R;
    plan.units = U;
    plan.nspan = 8 * R;
    for (int r = 0; r < R; ++r)
        for (int x = 0; x < 8; ++x) plan.span[8 * r + x] = r < n[x] ? (unsigned short)idx[x][r] : (unsigned short)SK_EMPTY;
}
static bool sk_finish_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MANTIS_GEMM_SK_FINISH"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
template <bool AKM, bool BKM, bool SWIGLU = false, int R16 = 0, int PAIR = PAIR_NONE>       // R16: 0 = the 32x32x16 kernel, 4 / 8 = ring16 waves
static int launch_gemm_ring(hipStream_t s, const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, long lda, long ldb,
                            long ldc, const bf16_t* bias, const bf16_t* res, long ldr, int flags, void* ws, long ws_bytes,
                            bf16_t* aux0 = nullptr, const bf16_t* aux1 = nullptr, long aux_ld = 0, int aux_n = 0) {
    const int tiles_m = cdiv(M, 256), tiles_n = cdiv(N, 256), nk = cdiv(K, BK);
    const long ntiles = (long)tiles_m * tiles_n;
    const int cus = plan_cus(flags_cus(flags));
    const int rem = (int)(ntiles % cus);
    const int S = ring_split(ntiles, nk, cus);
    const int full = S > 1 ? (int)(ntiles - rem) : (int)ntiles;
    const int grid = S > 1 ? full + S * rem : (int)ntiles;
    float* slabs = nullptr;
    unsigned int* cnt = nullptr;
    if (S > 1) {
        const int dev_cus = num_cus();            // the workspace layout is that of the whole device, whatever the planning budget
        if (!ws || ((uintptr_t)ws & 255) || ws_bytes < (long)sk_ws_bytes(dev_cus)) return MANTIS_EINVAL;   // see mantis_gemm_workspace_bytes
        cnt = (unsigned int*)ws;
        slabs = (float*)((char*)ws + sk_cnt_bytes(dev_cus));
    }
    if constexpr (R16 != 0) {
        // remainder tiles: the split units leave their slabs and gemm_ring16_finish_kernel reduces them on all CUs (cnt = nullptr tells the GEMM
        // kernel); MANTIS_GEMM_SK_FINISH=0 keeps the round-4 in-kernel reduction by the last arriver (A/B measurements)
        const bool finish = S > 1 && !(flags & EPI_SK_INKERNEL) && sk_finish_enabled();
        SkPlan plan;
        plan.units = 0;
        plan.nspan = 0;
        if (finish) sk_make_plan(plan, rem, S, nk, cus);
        const int grid16 = plan.units ? full + plan.units + plan.nspan : grid;
        const int S16 = plan.units ? rem : S;                  // balanced plan: the kernels take the number of remainder tiles here
        MANTIS_LAUNCH((gemm_nt_ring16_kernel<R16, AKM, BKM, SWIGLU, PAIR>), dim3(grid16), dim3(R16 * 64), 0, s, A, B, C, M, N, K, lda, ldb, ldc,
                           bias, res, ldr, flags, tiles_m, tiles_n, full, S16, slabs, finish ? nullptr : cnt, aux0, aux1, aux_ld, aux_n, plan, NoGroup{});
        if (finish) {
            // one wave per workgroup, except for the sum of squares (all R16 waves of a tile half in one block: fixed-order reduction)
#define FIN_ARGS 0, s, C, M, N, ldc, bias, res, ldr, flags, tiles_m, tiles_n, full, S16, slabs, aux0, aux1, aux_ld, aux_n, plan.units, nk
            bool done = false;
            if constexpr (!SWIGLU && PAIR == PAIR_NONE) {
                if (flags & EPI_SUMSQ) {
                    MANTIS_LAUNCH((gemm_ring16_finish_kernel<R16, (AKM || BKM), false, PAIR_NONE, R16>), dim3(rem * 2), dim3(R16 * 64), FIN_ARGS);
                    done = true;
                }
            }
            if (!done)
                MANTIS_LAUNCH((gemm_ring16_finish_kernel<R16, (AKM || BKM), SWIGLU, PAIR, 1>), dim3(rem * 2 * R16), dim3(64), FIN_ARGS);
#undef FIN_ARGS
        }
    } else {
        MANTIS_LAUNCH((gemm_nt_ring_kernel<AKM, BKM, SWIGLU>), dim3(grid), dim3(512), 0, s, A, B, C, M, N, K, lda, ldb, ldc, bias, res, ldr, flags,
                           tiles_m, tiles_n, full, S, slabs, cnt);
    }
    return mantis_check_launch();
}

// ---- the 176-row kernel: launcher and planner
// (gemm176.hip)
int mantis_launch_ring176(hipStream_t s, const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, long lda, long ldb, long ldc,
                          const bf16_t* bias, const bf16_t* res, long ldr, int flags, int bkm, int kind, int cus, bf16_t* aux0, const bf16_t* aux1,
                          long aux_ld, int aux_n);
// MANTIS_GEMM_176 (read once): 0 = never, 1 = where the planner predicts a gain (default), 2 = wherever the kernel applies (A/B measurements)
static int ring176_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MANTIS_GEMM_176"); v = (e && e[0] >= '0' && e[0] <= '2' && e[1] == 0) ? e[0] - '0' : 1; }
    return v;
}
// Predicted launch time (us) of the two tilings, per CU: rounds x (K-steps x loop time + fixed cost per tile).  256 x 256: the cost model of
// gemm_pick_variant (1.45 us per K-step, 5 K-step equivalents of prologue + epilogue per round, the K-split remainder round and its reduction);
// 176 x 256: R176_STEP_US per K-step (measured, profiles/r06_gemm_176.md) and the same fixed cost, whole tiles only.
#ifndef R176_STEP_US
#define R176_STEP_US 1.02
#endif
#ifndef R176_FIXED_US
#define R176_FIXED_US 7.5
#endif
static bool ring176_wins(int M, int N, int K, int cus_req) {
    const int mode = ring176_mode();
    if (mode == 0 || M < 176 * 2 || N < 256) return false;
    if (mode == 2) return true;
    const int cus = plan_cus(cus_req), nk = cdiv(K, BK);
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256), t176 = (long)cdiv(M, 176) * cdiv(N, 256);
    const int S = ring_split(t256, nk, cus);
    const long rem = t256 % cus;
    const int sub = rem ? (int)((S * rem + cus - 1) / cus) : 0;
    const double us256 = 1.45 * ((double)(t256 / cus) * (nk + 5.0) + (rem ? sub * ((double)nk / S + 5.0) + (S > 1 ? 8.0 + 1.7 * S : 0.0) : 0.0));
    const double us176 = (double)((t176 + cus - 1) / cus) * (nk * R176_STEP_US + R176_FIXED_US);
    return us176 < 0.97 * us256;
}

// which ring kernel the automatic choice takes (read once): MANTIS_GEMM_RING = 12 (8 waves, 32x32x16), 13 (4 waves x 128x128, 16x16x32) or
// 14 (8 waves x 128x64, 16x16x32) -- A/B measurements; the default is the measured per-shape choice of ring_variant_for()
static int default_ring_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MANTIS_GEMM_RING");
        v = (e && e[0] == '1' && e[1] >= '2' && e[1] <= '4' && e[2] == 0) ? 10 + (e[1] - '0') : 0;
    }
    return v;
}

extern "C" {

#ifdef RING16_STAMPS
// probe builds only: copy the stamps of the last ring16 launch (n workgroups x 8 u64) to host memory
int mantis_probe_ring16_stamps(void* host_dst, int n_wg) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_ring16_stamps), (size_t)n_wg * 64, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif

// tile variant the auto heuristic picks for C[M,N] over K (12 = 256x256 ring kernel, 1 = 128x128 generic kernel)
int mantis_gemm_pick_variant(int M, int N, int K) { return gemm_pick_variant(M, N, K); }
// the same for a launch planned for `cus` compute units (bits 16-27 of its flags); cus <= 0: the default budget
int mantis_gemm_pick_variant_cus(int M, int N, int K, int cus) { return gemm_pick_variant(M, N, K, cus > 0 ? cus : 0); }

// The remainder-round plan of a ring16 launch of C[M,N] over K planned for `cus` compute units (<= 0: the default budget), as the launcher and
// the kernels compute it -- for tests and tools, no device work.  out[0..7] = {tiles, full, remainder tiles, S (equal split; 1 = no split),
// balanced units (0 = equal split), tail slots, remainder workgroups in the grid, K-steps}; then, while they fit `cap` ints, one record of 4
// ints per remainder workgroup in grid order: {remainder tile (-1: an empty tail slot), first K-step, end K-step, slab}.  Returns the number of
// ints the full description has.  (The records restate the kernel's own arithmetic, gemm_nt_ring16_kernel: the same sk_bound.)
int mantis_gemm_remainder_plan(int M, int N, int K, int cus_req, int* out, int cap) {
    if (M <= 0 || N <= 0 || K <= 0 || !out || cap < 8) return MANTIS_EINVAL;
    const int tiles_m = cdiv(M, 256), tiles_n = cdiv(N, 256), nk = cdiv(K, BK);
    const long ntiles = (long)tiles_m * tiles_n;
    const int cus = plan_cus(cus_req > 0 ? cus_req : 0);
    const int rem = (int)(ntiles % cus);
    const int S = ring_split(ntiles, nk, cus);
    const int full = S > 1 ? (int)(ntiles - rem) : (int)ntiles;
    SkPlan plan;
    plan.units = 0;
    plan.nspan = 0;
    if (S > 1 && sk_finish_enabled()) sk_make_plan(plan, rem, S, nk, cus);
    const int nwg = S > 1 ? (plan.units ? plan.units + plan.nspan : S * rem) : 0;
    const int head[8] = {(int)ntiles, full, S > 1 ? rem : 0, S, plan.units, plan.nspan, nwg, nk};
    for (int i = 0; i < 8; ++i) out[i] = head[i];
    int n = 8;
    for (int j = 0; j < nwg; ++j, n += 4) {
        int rec[4];
        if (plan.units == 0) {
            const int nu = nwg, q = nu >> 3, r = nu & 7, xcd = j & 7;
            const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (j >> 3);
            const int part = lin / rem, tl = lin - part * rem;
            rec[0] = tl;
            rec[1] = (int)((unsigned)nk * (unsigned)part / (unsigned)S);
            rec[2] = (int)((unsigned)nk * (unsigned)(part + 1) / (unsigned)S);
            rec[3] = tl * S + part;
        } else {
            const int U = plan.units, T = rem * nk;
            const bool first = j < U;
            const int c = first ? j : (int)plan.span[j - U];
            if (c == (int)SK_EMPTY) {
                rec[0] = -1; rec[1] = rec[2] = rec[3] = 0;
            } else {
                const int a0 = sk_bound(c, U, T, nk), a1 = sk_bound(c + 1, U, T, nk);
                int tl = a0 / nk;
                if (first) {
                    const int e = (tl + 1) * nk;
                    rec[0] = tl; rec[1] = a0 - tl * nk; rec[2] = (a1 < e ? a1 : e) - tl * nk; rec[3] = c;
                } else {
                    tl += 1;
                    rec[0] = tl; rec[1] = 0; rec[2] = a1 - tl * nk; rec[3] = U + tl - 1;
                }
            }
        }
        if (n + 4 <= cap) for (int i = 0; i < 4; ++i) out[n + i] = rec[i];
    }
    return n;
}

// C[M,N] (bf16, row stride ldc) = epilogue(A[M,K] . B[N,K]^T); A,B,C 16-B aligned, lda/ldb % 8 == 0, K % 8 == 0.
// flags: bit0 bias[n] add | bits1-3 activation (1 gelu-erf, 2 gelu-tanh, 3 quick-gelu) | bit4 + residual[m,n] (stride ldr)
//        | bit5 accumulate into C (C += result, used for gradient accumulation) | bits8-11 tile variant (0 = auto)
// Bytes of caller-owned, 256-B aligned, ZERO-INITIALISED device workspace the launch of C[M,N] over K needs (0 = none): only the
// 256x256 ring kernel's split-K remainder round uses it.  M = N = K = 0 returns the largest requirement of any shape on this
// device, so a caller can allocate one buffer per stream up front (launches sharing a workspace must be stream-ordered).
int mantis_gemm_workspace_bytes(int M, int N, int K) {
    const int cus = num_cus();
    if (M <= 0 && N <= 0 && K <= 0) return (int)sk_ws_bytes(cus);
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
    return ring_split(t256, cdiv(K, BK), plan_cus(0)) > 1 ? (int)sk_ws_bytes(cus) : 0;
}

// CU budget a launch plans for when it asks for `cus` (bits 16-27 of its flags): cus > 0 -> clamped to [8, #CU]; cus <= 0 -> the default
// (MANTIS_GEMM_CUS, else the whole device).  A pure query: there is no process-wide budget to set any more (round 5).
int mantis_gemm_cu_budget(int cus) { return plan_cus(cus > 0 ? cus : 0); }

int mantis_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                        const void* bias, const void* residual, int64_t ldr, int flags, void* workspace, int64_t workspace_bytes,
                        void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (flags & EPI_SUMSQ)) return MANTIS_EINVAL;      // bit 7 belongs to mantis_gemm_bf16_nt_sumsq
    const bool akm = flags & EPI_A_KMAJOR, bkm = flags & EPI_B_KMAJOR;
    if (lda % 8 || ldb % 8 || ldc < N) return MANTIS_EUNSUPPORTED;
    const int K8 = (K + 7) / 8 * 8;
    if ((!akm && lda < K8) || (!bkm && ldb < K8) || (akm && lda < M) || (bkm && ldb < N)) return MANTIS_EUNSUPPORTED;
    // row-major operands are fetched in whole 16-B chunks along K: with K % 8 != 0 the tail chunk must meet zeros on the other
    // side, which only a K-major operand (rows k >= K come from the zero page) guarantees; the row-major pad must be finite
    if (!akm && !bkm && (K % 8)) return MANTIS_EUNSUPPORTED;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return MANTIS_EUNSUPPORTED;
    if ((flags & EPI_BIAS) && !bias) return MANTIS_EINVAL;
    if ((flags & EPI_RESIDUAL) && !residual) return MANTIS_EINVAL;
    if (flags & EPI_SWIGLU_BWD) {      // residual = [gate | up], C = [dgate | dup], both [M, 2N]; only the ring kernel's epilogue has it
        if (!residual || (flags & (EPI_BIAS | EPI_ACT_MASK | EPI_RESIDUAL | EPI_ACCUM)) || ldc < 2 * (long)N || ldr < 2 * (long)N)
            return MANTIS_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    int variant = (flags & EPI_VARIANT_MASK) >> EPI_VARIANT_SHIFT;
    // variant 12 = the 8-wave ring kernel (32x32x16 MFMA), 13 / 14 = the ring16 kernel (16x16x32 MFMA) with 4 / 8 waves: same tile, same
    // work split, same results up to the accumulation order inside a K-step
    if (variant == 0) {
        variant = (flags & EPI_SWIGLU_BWD) ? 12 : gemm_pick_variant(M, N, K, flags_cus(flags));
        if (variant == 12) {
            variant = default_ring_variant() ? default_ring_variant() : ring_variant_for(M, N, K, akm, bkm, flags_cus(flags));
            // the 176-row tile where it turns the grid into whole rounds (A row-major: forward and dX)
            if (!default_ring_variant() && !akm && ring176_wins(M, N, K, flags_cus(flags))) variant = 15;
        }
    }
    const bool r176 = variant == 15;
    const bool ring = variant >= 12 && variant <= 15, big = variant == 2;
    if (r176 && akm) return MANTIS_EUNSUPPORTED;               // the M side is row-major in the 176-row kernel
    if ((flags & EPI_SWIGLU_BWD) && (!ring || akm || !bkm)) return MANTIS_EUNSUPPORTED;
    // the ring kernels address their operands through buffer descriptors: unsigned 32-bit num_records, unsigned 32-bit lane offset and
    // unsigned 32-bit scalar K-step offset (the hardware adds them to the 48-bit base without wrapping).  An operand of 4 GiB or more
    // would wrap silently -> such shapes go to the generic kernel (64-bit addresses) when the choice is ours, and are refused when a
    // ring kernel was asked for explicitly.  (The first guard stood at 2 GiB; the lm_head logits of 8192 rows x 152064 columns, 2.5 GB,
    // then fell to the generic kernel at 0.34 PFLOP/s.  `gemm_operand_over_2gib` checks the 2-4 GiB range against the generic kernel.)
    if (ring) {
        const long a_bytes = (long)(akm ? K : M) * (long)lda * 2, b_bytes = (long)(bkm ? K : N) * (long)ldb * 2;
        const long lim = (1L << 32) - (1L << 16);
        if (a_bytes >= lim || b_bytes >= lim) {
            if ((flags & EPI_VARIANT_MASK) || (flags & EPI_SWIGLU_BWD)) return MANTIS_EUNSUPPORTED;
            variant = 1;
            return mantis_gemm_bf16_nt(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, flags | (1 << EPI_VARIANT_SHIFT), workspace,
                                       workspace_bytes, stream);
        }
    }
#define GEMM_ARGS s, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, (long)lda, (long)ldb, (long)ldc, \
                  (const bf16_t*)bias, (const bf16_t*)residual, (long)ldr, flags
#define RING_ARGS GEMM_ARGS, workspace, (long)workspace_bytes
    // variant 1 = 128x128 generic kernel, 2 = 256x256 generic kernel, 12 / 13 = 256x256 ring kernels (default for well-quantised shapes)
    if (!ring && !big && variant != 1) return MANTIS_EINVAL;
    if (r176) {
        return mantis_launch_ring176(GEMM_ARGS, bkm ? 1 : 0, 0, plan_cus(flags_cus(flags)), nullptr, nullptr, 0L, 0);
    }
#define RING_DISPATCH(AK, BK_, SW) (variant == 13 ? launch_gemm_ring<AK, BK_, SW, 4>(RING_ARGS) \
                                    : variant == 14 ? launch_gemm_ring<AK, BK_, SW, 8>(RING_ARGS) : launch_gemm_ring<AK, BK_, SW, 0>(RING_ARGS))
    if (akm && bkm)
        return ring ? RING_DISPATCH(true, true, false)
                    : big ? launch_gemm<256, 256, 128, 64, true, true>(GEMM_ARGS) : launch_gemm<128, 128, 64, 64, true, true>(GEMM_ARGS);
    if (flags & EPI_SWIGLU_BWD) return RING_DISPATCH(false, true, true);
    if (bkm)
        return ring ? RING_DISPATCH(false, true, false)
                    : big ? launch_gemm<256, 256, 128, 64, false, true>(GEMM_ARGS) : launch_gemm<128, 128, 64, 64, false, true>(GEMM_ARGS);
    if (akm)
        return ring ? RING_DISPATCH(true, false, false)
                    : big ? launch_gemm<256, 256, 128, 64, true, false>(GEMM_ARGS) : launch_gemm<128, 128, 64, 64, true, false>(GEMM_ARGS);
    return ring ? RING_DISPATCH(false, false, false)
                : big ? launch_gemm<256, 256, 128, 64, false, false>(GEMM_ARGS) : launch_gemm<128, 128, 64, 64, false, false>(GEMM_ARGS);
#undef RING_DISPATCH
#undef RING_ARGS
#undef GEMM_ARGS
}

// Weight-gradient GEMM that also leaves the squared norm of its result: C (+)= A . B^T exactly as mantis_gemm_bf16_nt computes it (flags:
// 32 accumulate | 4096 / 8192 K-major operands | bits 8-11 variant 13 / 14 or 0), and tile_sumsq[t] = sum over the 256 x 256 output tile t
// (tile ids in the kernel's XCD-grouped order; every one of the cdiv(M,256) * cdiv(N,256) entries is written exactly once) of the squares
// of the bf16 values stored -- after the accumulation when flag 32 is set.  The optimizer's clip_grad_norm_ (HF trainer.py:2535-2545) then
// sums tile_sumsq instead of re-reading the gradients.  Deterministic.  Ring16 kernels only: N % 256 == 0, ldc % 8 == 0, C 16-B aligned,
// no bias / activation / residual; anything else returns MANTIS_EUNSUPPORTED and the caller runs mantis_gemm_bf16_nt + mantis_sumsq.
int mantis_gemm_bf16_nt_sumsq(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K, int flags,
                              float* tile_sumsq, void* workspace, int64_t workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !tile_sumsq) return MANTIS_EINVAL;
    if (flags & ~(EPI_ACCUM | EPI_A_KMAJOR | EPI_B_KMAJOR | EPI_VARIANT_MASK | EPI_SK_INKERNEL | EPI_CUS_MASK)) return MANTIS_EUNSUPPORTED;
    const bool akm = flags & EPI_A_KMAJOR, bkm = flags & EPI_B_KMAJOR;
    if (N % 256 || ldc % 8 || ldc < N || lda % 8 || ldb % 8 || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15)) return MANTIS_EUNSUPPORTED;
    const int K8 = (K + 7) / 8 * 8;
    if ((!akm && lda < K8) || (!bkm && ldb < K8) || (akm && lda < M) || (bkm && ldb < N) || (!akm && !bkm && (K % 8))) return MANTIS_EUNSUPPORTED;
    const long a_bytes = (long)(akm ? K : M) * (long)lda * 2, b_bytes = (long)(bkm ? K : N) * (long)ldb * 2, lim = (1L << 32) - (1L << 16);
    if (a_bytes >= lim || b_bytes >= lim) return MANTIS_EUNSUPPORTED;
    int variant = (flags & EPI_VARIANT_MASK) >> EPI_VARIANT_SHIFT;
    if (variant == 0) {
        if (gemm_pick_variant(M, N, K, flags_cus(flags)) != 12) return MANTIS_EUNSUPPORTED;
        variant = default_ring_variant() >= 13 ? default_ring_variant() : ring_variant_for(M, N, K, akm, bkm, flags_cus(flags));
    }
    if (variant != 13 && variant != 14) return MANTIS_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int f = (flags & ~EPI_VARIANT_MASK) | EPI_SUMSQ;
#define SS_ARGS s, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, (long)lda, (long)ldb, (long)ldc, (const bf16_t*)nullptr, \
                (const bf16_t*)nullptr, 0L, f, workspace, (long)workspace_bytes, (bf16_t*)tile_sumsq, (const bf16_t*)nullptr, 0L, 0
#define SS_DISPATCH(AK, BK_) (variant == 13 ? launch_gemm_ring<AK, BK_, false, 4>(SS_ARGS) : launch_gemm_ring<AK, BK_, false, 8>(SS_ARGS))
    if (akm && bkm) return SS_DISPATCH(true, true);
    if (bkm) return SS_DISPATCH(false, true);
    if (akm) return SS_DISPATCH(true, false);
    return SS_DISPATCH(false, false);
#undef SS_DISPATCH
#undef SS_ARGS
}

// ---- two weight-gradient GEMMs in one grid (see RingGroup2 at the kernel)
// predicted launch time (us) of C[M,N] over K on the 256 x 256 ring kernels for `cus` CUs, the cost model of gemm_pick_variant
static double ring256_us(int M, int N, int K, int cus) {
    const int nk = cdiv(K, BK);
    const long t = (long)cdiv(M, 256) * cdiv(N, 256);
    const int S = ring_split(t, nk, cus);
    const long rem = t % cus;
    const int sub = rem ? (int)((S * rem + cus - 1) / cus) : 0;
    return 1.45 * ((double)(t / cus) * (nk + 5.0) + (rem ? sub * ((double)nk / S + 5.0) + (S > 1 ? 8.0 + 1.7 * S : 0.0) : 0.0));
}
static bool tn_pair_wins(int M1, int N1, int M2, int N2, int K, int cus_req) {
    static int mode = -1;              // MANTIS_GEMM_PAIR (read once): 0 never, 1 where the model predicts a gain (default), 2 always
    if (mode < 0) { const char* e = getenv("MANTIS_GEMM_PAIR"); mode = (e && e[0] >= '0' && e[0] <= '2' && e[1] == 0) ? e[0] - '0' : 1; }
    if (mode == 0 || default_ring_variant()) return false;
    if (mode == 2) return true;
    const int cus = plan_cus(cus_req), nk = cdiv(K, BK);
    const long t = (long)cdiv(M1, 256) * cdiv(N1, 256) + (long)cdiv(M2, 256) * cdiv(N2, 256);
    const double both = 1.45 * (double)((t + cus - 1) / cus) * (nk + 5.0);
    return both < 0.97 * (ring256_us(M1, N1, K, cus) + ring256_us(M2, N2, K, cus));
}
static int tn_operand_ok(const void* A, int64_t lda, const void* B, int64_t ldb, const void* C, int64_t ldc, int M, int N, int K) {
    if (M <= 0 || N <= 0) return MANTIS_EINVAL;
    if (N % 256 || ldc % 8 || ldc < N || lda % 8 || ldb % 8 || lda < M || ldb < N || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15)) return MANTIS_EUNSUPPORTED;
    const long lim = (1L << 32) - (1L << 16);
    if ((long)K * lda * 2 >= lim || (long)K * ldb * 2 >= lim) return MANTIS_EUNSUPPORTED;
    return 0;
}

// 1 when mantis_gemm_bf16_tn_pair would run C1[M1,N1] and C2[M2,N2] (both over K, both operands K-major) as ONE grid and the cost model
// predicts that to beat two launches by >= 3 % for `cus` CUs (<= 0: the default budget) -- e.g. dW(down_proj) + dW(q|k|v) of a Llama-3-8B /
// Mistral-7B layer: 896 + 384 tiles = 5.0 rounds on 256 CUs; 0 otherwise.  A pure query.
int mantis_gemm_tn_pair_wins(int M1, int N1, int M2, int N2, int K, int cus) {
    if (M1 <= 0 || N1 <= 0 || M2 <= 0 || N2 <= 0 || K <= 0 || N1 % 256 || N2 % 256) return 0;
    return tn_pair_wins(M1, N1, M2, N2, K, cus > 0 ? cus : 0) ? 1 : 0;
}
// Two weight-gradient GEMMs in ONE launch: C1 (+)= A1^T-view . B1 and C2 (+)= A2^T-view . B2, i.e. mantis_gemm_bf16_nt_sumsq's TN form
// (A given [K, M], B given [K, N]; flags: 32 accumulate | bits 16-27 CU budget) for two problems that share K.  ts1 / ts2: per 256 x 256 tile
// sums of squares of what was stored (cdiv(M,256) * cdiv(N,256) floats each), or both NULL for none.  Whole tiles of the 8-wave ring16 kernel
// in one XCD-contiguous order over the union of the two grids: no K split, no workspace.  Each result is bit-identical to the one
// mantis_gemm_bf16_nt(_sumsq) writes with variant 14 for a shape WITHOUT a K-split remainder (same kernel code, whole tiles); against a
// K-split launch of the same shape the fp32 summation order of the remainder tiles differs (bf16 rounding).  MANTIS_EUNSUPPORTED: N % 256, unaligned
// operands, >= 4 GiB operands.
int mantis_gemm_bf16_tn_pair(const void* A1, int64_t lda1, const void* B1, int64_t ldb1, void* C1, int64_t ldc1, int M1, int N1, float* ts1,
                             const void* A2, int64_t lda2, const void* B2, int64_t ldb2, void* C2, int64_t ldc2, int M2, int N2, float* ts2,
                             int K, int flags, void* stream) {
    if (K <= 0 || (ts1 == nullptr) != (ts2 == nullptr)) return MANTIS_EINVAL;
    if (flags & ~(EPI_ACCUM | EPI_CUS_MASK)) return MANTIS_EUNSUPPORTED;
    int rc = tn_operand_ok(A1, lda1, B1, ldb1, C1, ldc1, M1, N1, K);
    if (rc == 0) rc = tn_operand_ok(A2, lda2, B2, ldb2, C2, ldc2, M2, N2, K);
    if (rc != 0) return rc;
    RingGroup2 g;
    g.A = (const bf16_t*)A2; g.B = (const bf16_t*)B2; g.C = (bf16_t*)C2; g.aux0 = (bf16_t*)ts2;
    g.lda = (long)lda2; g.ldb = (long)ldb2; g.ldc = (long)ldc2;
    g.M = M2; g.N = N2; g.tiles_m = cdiv(M2, 256); g.tiles_n = cdiv(N2, 256);
    const int tm1 = cdiv(M1, 256), tn1 = cdiv(N1, 256);
    g.tiles1 = tm1 * tn1;
    const int total = g.tiles1 + g.tiles_m * g.tiles_n;
    SkPlan plan;
    plan.units = 0;
    plan.nspan = 0;
    const int f = (flags & EPI_ACCUM) | EPI_A_KMAJOR | EPI_B_KMAJOR | (ts1 ? EPI_SUMSQ : 0);
    MANTIS_LAUNCH((gemm_nt_ring16_kernel<8, true, true, false, PAIR_NONE, RingGroup2>), dim3(total), dim3(512), 0, (hipStream_t)stream,
                       (const bf16_t*)A1, (const bf16_t*)B1, (bf16_t*)C1, M1, N1, K, (long)lda1, (long)ldb1, (long)ldc1, (const bf16_t*)nullptr,
                       (const bf16_t*)nullptr, 0L, f, tm1, tn1, total, 1, (float*)nullptr, (unsigned int*)nullptr, (bf16_t*)ts1,
                       (const bf16_t*)nullptr, 0L, 0, plan, g);
    return mantis_check_launch();
}

// Forward projections with a two-column epilogue fused in (ring16 kernels, NT layout, see PAIR_* at the kernel):
//   mode 1 (SwiGLU): A = x [M, K], B = [gate | up] weight [2 I, K] -> C = x . B^T [M, 2 I] as mantis_gemm_bf16_nt writes it, and
//                    aux0 = silu(gate) * up [M, I] (bf16, row stride aux_ld) as mantis_swiglu_fwd computes it; N = 2 I, I % 128 == 0
//   mode 2 (RoPE):   q|k|v projection with heads of 128 columns, optional bias: columns [0, aux_n) leave with the rotary embedding applied
//                    (aux0 = cos, aux1 = sin: bf16 [M, 64], row stride aux_ld) exactly as mantis_rope_apply(forward) would rotate them
//                    afterwards; N % 256 == 0, aux_n % 128 == 0
// Returns MANTIS_EUNSUPPORTED for shapes outside these conditions: the caller then runs the two launches.
int mantis_gemm_bf16_nt_fused(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                              const void* bias, int mode, void* aux0, const void* aux1, int64_t aux_ld, int aux_n, int variant,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (mode != PAIR_SWIGLU && mode != PAIR_ROPE) || !aux0) return MANTIS_EINVAL;
    if (lda % 8 || ldb % 8 || ldc % 8 || ldc < N || lda < K || ldb < K || K % 8 || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15))
        return MANTIS_EUNSUPPORTED;
    if (aux_ld % 8 || ((uintptr_t)aux0 & 15)) return MANTIS_EUNSUPPORTED;
    if (mode == PAIR_SWIGLU && (N % 256 || aux_ld < N / 2 || bias)) return MANTIS_EUNSUPPORTED;
    if (mode == PAIR_ROPE && (N % 256 || aux_n % 128 || aux_n > N || aux_n < 0 || !aux1 || aux_ld < 64 || ((uintptr_t)aux1 & 15)))
        return MANTIS_EUNSUPPORTED;
    const long lim = (1L << 32) - (1L << 16);
    if ((long)M * lda * 2 >= lim || (long)N * ldb * 2 >= lim) return MANTIS_EUNSUPPORTED;
    const bool inkernel = variant & 64;          // bit 6 of `variant`: see EPI_SK_INKERNEL
    const bool shared = variant & 128;           // bit 7 of `variant`: see EPI_SHARED_GPU
    const int cus_bits = variant & EPI_CUS_MASK; // bits 16-27 of `variant`: the CU budget, as in mantis_gemm_bf16_nt's flags
    variant &= 15;
    if (variant == 0) {
        variant = default_ring_variant() >= 13 ? default_ring_variant() : ring_variant_for(M, N, K, false, false, cus_bits >> EPI_CUS_SHIFT);
        if (default_ring_variant() < 13 && ring176_wins(M, N, K, cus_bits >> EPI_CUS_SHIFT)) variant = 15;
    }
    if (variant < 13 || variant > 15) return MANTIS_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int flags = (bias ? EPI_BIAS : 0) | (inkernel ? EPI_SK_INKERNEL : 0) | (shared ? EPI_SHARED_GPU : 0) | cus_bits;
#define PAIR_ARGS s, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, (long)lda, (long)ldb, (long)ldc, (const bf16_t*)bias, \
                  (const bf16_t*)nullptr, 0L, flags, workspace, (long)workspace_bytes, (bf16_t*)aux0, (const bf16_t*)aux1, (long)aux_ld, aux_n
    if (variant == 15) {
        return mantis_launch_ring176(s, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, (long)lda, (long)ldb, (long)ldc, (const bf16_t*)bias,
                                     (const bf16_t*)nullptr, 0L, flags, 0, mode, plan_cus(cus_bits >> EPI_CUS_SHIFT), (bf16_t*)aux0,
                                     (const bf16_t*)aux1, (long)aux_ld, aux_n);
    }
    if (mode == PAIR_SWIGLU)
        return variant == 13 ? launch_gemm_ring<false, false, false, 4, PAIR_SWIGLU>(PAIR_ARGS)
                             : launch_gemm_ring<false, false, false, 8, PAIR_SWIGLU>(PAIR_ARGS);
    return variant == 13 ? launch_gemm_ring<false, false, false, 4, PAIR_ROPE>(PAIR_ARGS)
                         : launch_gemm_ring<false, false, false, 8, PAIR_ROPE>(PAIR_ARGS);
#undef PAIR_ARGS
}

}  // extern "C"
