// 176 x 256 bf16 ring GEMM for gfx950 (round 6) -- one of the tile geometries behind mantis_gemm_bf16_nt / mantis_gemm_bf16_nt_fused (gemm.hip owns
// the entry points, the planner and the reference citations; this translation unit holds the kernel so that it builds on its own).
#include "gemm_ring.h"

// ---------------------------------------------------------------------------------------------------------------------
// 176 x 256 "ring176" kernel (round 6): whole-round tilings for the M = 5624 shapes of the headline step.
// 5624 merged rows = 21.97 tile rows of 256 -- the forward / dX grids of the step come to 352 / 528 tiles on 256 CUs, 1.375 / 2.06 rounds, and the
// K-split remainder round + its finishing pass cost more than they save on the short-K shapes (o, q|k|v forward, dX(o), dX(q|k|v): 0.37 - 0.46
// of peak against 0.49 - 0.54 on the whole-round shapes, profiles/r05_gemm_in_step_by_shape.md).  5624 = 32 x 175.75: with a 176-row tile every N of
// the step (a multiple of 8 tile columns of 256) gives a WHOLE number of rounds: 32 x 16 = 512 = 2.0 (N = 4096), 32 x 24 = 768 = 3.0 (N = 6144).
// No K split, no slabs, no finishing kernel, no second prologue; 0.14 % of the rows are padding.
// Wave layout: 4 waves (one per SIMD, 512 registers each), wave w = all 176 rows x columns 64 w .. 64 w + 63: 11 x 4 blocks of
// v_mfma_f32_16x16x32_bf16 = 176 accumulators in AGPRs; per 32-k half 11 A + 4 B fragment reads (ds_read_b128) feed 44 MFMAs.
// LDS: a K-step is A 22 KiB (176 rows x 128 B) + B 32 KiB, which does not divide the ring16 kernels' ten 16-KiB slabs, and 3 steps (162 KiB) do not
// fit 160 KiB -- so TWO rings with the same 2.5-step schedule: B in three 32-KiB slots (issued in the first half of a step, 1.5 steps of lead),
// A in two 22-KiB slots (issued behind the mid-step barrier that frees the slot of the step being computed, one step of lead): 140 KiB.
//   half 0 of step t: MFMAs (t, k 0-31);  shadows: fragments (t, k 32-63), DMA B(t+2) -> the B slot step t-1 used (three slots in turn)
//   lgkmcnt(0), vmcnt(8) [all but B(t+2)], s_barrier: step t read by every wave, step t+1 landed
//   half 1 of step t: MFMAs (t, k 32-63); shadows: fragments (t+1, k 0-31), DMA A(t+2) -> A slot t % 2
// DMA pieces (1 KiB = 8 rows x 128 B): B 32 = 8 per wave (wave w loads the 64 B rows it reads); A 22: piece w + 4 j, waves 0,1 six, waves 2,3
// five (a wave-uniform branch; the counted wait names only the uniform B pieces).  A is row-major (forward and dX: the M side is the
// activation); B row-major (forward) or K-major (dX reads the weight as stored).  Epilogue: ring_epilogue<4, 4, 11> (gemm_ring.h) -- the ring16 code
// with a short last pass -- plain / bias / activations / residual / accumulate / fused SwiGLU backward (fast read-back: gate | up requested a group
// ahead) / the two-column forward fusions.
// PERSIST (launches with more tiles than CUs): #CU workgroups walk their tiles in the hardware's dispatch order; behind the K loop's last barrier
// the NEXT tile's first two K-steps are issued and fly under the epilogue, whose strips are 48-row passes in [B slot 2 | spare] (see the LDS map
// below).  Same arithmetic in the same order as one tile per workgroup: bit-identical (check gemm_ring176_persistent).  Measured:
// profiles/r06_experiments.md 1 - 3.
#ifndef R176_FASTSW
#define R176_FASTSW 8       // read-back group of the fused SwiGLU backward (0 = the general read-back, 4, 8)
#endif
template <bool BKM, bool SWIGLU = false, int PAIR = PAIR_NONE, bool PERSIST = true>
__global__ __launch_bounds__(256) void gemm_nt_ring176_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M, int N, int K, long lda, long ldb,
    long ldc, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res, long ldr, int flags, int tiles_m, int tiles_n,
    bf16_t* __restrict__ aux0, const bf16_t* __restrict__ aux1, long aux_ld, int aux_n) {
    static_assert(PAIR == PAIR_NONE || (!BKM && !SWIGLU), "the pair epilogues are forward (NT) fusions");
    constexpr int NBM = 11, NBN = 4, NMF = NBN * NBM, BMT = 176;
    // LDS map: [B0 32K][B1 32K][A0 22K][A1 22K][B2 32K][spare 20K].  PERSIST: while the epilogue of a tile runs, steps 0 and 1 of the
    // workgroup's NEXT tile land in B0 / A0 / B1 / A1, and the epilogue's four wave strips take [B2 | spare] = 52 KiB: 48-row passes (3 blocks,
    // 4 x 12.75 KiB) instead of the ring16 kernels' 64-row ones.  Without PERSIST (one tile per workgroup) the strips lie at 0 as 64-row passes.
    constexpr int B_SLOT = 32768, A_SLOT = 22528, A_OFF = 2 * B_SLOT, B2_OFF = A_OFF + 2 * A_SLOT, LDS_BYTES = 163840;
    constexpr int PB = PERSIST ? 3 : 4, STRIP_OFF = PERSIST ? B2_OFF : 0;
    static_assert(B2_OFF + B_SLOT <= LDS_BYTES && STRIP_OFF + 4 * PB * 16 * EPI_PITCH <= LDS_BYTES, "LDS map");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wnh = wave >> 1, wno = (wave & 1) * 4;         // K-major B image: 128-column half, first 16-column block inside it

    // (virtual) workgroup v -> tile: XCD-contiguous ranges of tile ids, groups of 8 tile rows walked column-major (as the ring16 kernels).
    // PERSIST: workgroup b runs v = b, b + gridDim.x, ...: with a grid of #CU workgroups (a multiple of 8) every v of a workgroup lies on
    // the same XCD's range, in the order the hardware would have dispatched them
    const int nk = (K + BK - 1) / BK;
    const int ntile = tiles_m * tiles_n;
    int m0, n0, tile_id;
    auto tile_of = [&](int v, int& tm0, int& tn0, int& tid_) {
        const int q = ntile >> 3, r = ntile & 7, xcd = v & 7;
        tid_ = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        const int GROUP = 8;
        const int per_group = GROUP * tiles_n;
        const int g = tid_ / per_group;
        const int first_m = g * GROUP;
        const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
        const int in_g = tid_ - g * per_group;
        tm0 = (first_m + in_g % gsz) * BMT;
        tn0 = (in_g / gsz) * 256;
    };
    int v = blockIdx.x;
    tile_of(v, m0, n0, tile_id);
    const int pair_dist = PAIR == PAIR_SWIGLU ? (N >> 1) : 64;

    f32x4 acc[NBN][NBM];

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(unsigned)(((long)M * lda) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)B, 0, (int)(unsigned)(((long)(BKM ? K : N) * ldb) * 2), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned voA[6], voB[8];
    int kcA, kcB[8];
    // per-lane DMA offsets of a tile (operands < 4 GiB: 32-bit arithmetic on the byte offsets)
    auto lane_offsets = [&](int tm0, int tn0) {
        {
            // A, row-major: piece wave + 4 j = rows 8 piece .. + 7; 16-B chunk swizzled with the row pair (slot = chunk ^ ((row >> 1) & 7));
            // (4 piece + (rl >> 1)) & 7 does not depend on j
            const int rl = lane >> 3, fz = (wave * 4 + (rl >> 1)) & 7, chunk = (lane & 7) ^ fz;
            kcA = chunk * 8;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                int grow = tm0 + (wave + 4 * j) * 8 + rl;
                grow = grow < M ? grow : M - 1;
                voA[j] = (unsigned)grow * (unsigned)(lda * 2) + (unsigned)(chunk * 16);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (!BKM) {
                // B, row-major: piece 8 wave + j of the 256-row slot
                const int piece = 8 * wave + j, rl = lane >> 3, fz = (piece * 4 + (rl >> 1)) & 7, chunk = (lane & 7) ^ fz;
                const int row = piece * 8 + rl;
                kcB[j] = chunk * 8;
                int grow = tn0 + row;
                if constexpr (PAIR != PAIR_NONE) {
                    // a wave's 64 rows = [32 first columns | 32 second columns] of its 32 features
                    constexpr int G = NBN * 8;
                    const int half = row >> 7, r = row & 127;
                    const int phi = half * 64 + (r / (2 * G)) * G + (r % G), second = (r / G) & 1;
                    const int first = PAIR == PAIR_SWIGLU ? (tn0 >> 1) + phi : tn0 + (phi >> 6) * 128 + (phi & 63);
                    grow = first + second * pair_dist;
                }
                grow = grow < N ? grow : N - 1;
                voB[j] = (unsigned)grow * (unsigned)(ldb * 2) + (unsigned)(chunk * 16);
            } else {
                // B, K-major: two [64 k][128 columns] images per slot; piece 4 wave + (j & 3) of half j >> 2: 4 k-rows x 256 B; LDS slot s of
                // k-row kk holds the 16-B chunk s ^ ((kk & 3) << 2) ^ (((kk >> 3) & 1) << 1) (as the ring16 kernels)
                const int half = j >> 2, piece = 4 * wave + (j & 3);
                const int kk = piece * 4 + (lane >> 4), slot = lane & 15, cc = slot ^ ((kk & 3) << 2) ^ (((kk >> 3) & 1) << 1);
                const long col = (long)tn0 + half * 128 + cc * 8;
                kcB[j] = kk;
                voB[j] = (col + 8 <= ldb) ? (unsigned)kk * (unsigned)(ldb * 2) + (unsigned)(col * 2) : OOB;
            }
        }
    };
    lane_offsets(m0, n0);
    const unsigned soA1 = (unsigned)(BK * 2);
    const unsigned soB1 = BKM ? (unsigned)(BK * 2) * (unsigned)ldb : (unsigned)(BK * 2);
    // LDS destination of piece j inside a slot
    auto dstA = [&](int j) { return (unsigned)(wave + 4 * j) * 1024u; };
    auto dstB = [&](int j) { return BKM ? (unsigned)(j >> 2) * 16384u + (unsigned)(4 * wave + (j & 3)) * 1024u : (unsigned)(8 * wave + j) * 1024u; };
    auto issueB = [&](auto jc, unsigned slot_off, unsigned sob, int krem) {
        constexpr int j = decltype(jc)::value;
        lds_void* d = (lds_void*)(smem + slot_off + dstB(j));
        const unsigned vo = (kcB[j] < krem) ? voB[j] : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, d, 16, vo, sob, 0, DMA_AUX_B);
    };
    auto issueA = [&](auto jc, unsigned slot_off, unsigned soa, int krem) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j == 5) { if (wave >= 2) return; }           // pieces 22, 23 do not exist (wave-uniform)
        lds_void* d = (lds_void*)(smem + A_OFF + slot_off + dstA(j));
        const unsigned vo = (kcA < krem) ? voA[j] : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, d, 16, vo, soa, 0, DMA_AUX_A);
    };
    // a tile's first two K-steps: B(0) -> B0, A(0) -> A0, B(1) -> B1, A(1) -> A1
    auto prologue = [&]() {
        static_for<0, 8>([&](auto jc) { issueB(jc, 0u, 0u, K); });
        static_for<0, 6>([&](auto jc) { issueA(jc, 0u, 0u, K); });
        static_for<0, 8>([&](auto jc) { issueB(jc, (unsigned)B_SLOT, soB1, K - BK); });
        static_for<0, 6>([&](auto jc) { issueA(jc, (unsigned)A_SLOT, soA1, K - BK); });
    };
    prologue();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NBN; ++i)
#pragma unroll
        for (int j = 0; j < NBM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned r16 = (unsigned)lane & 15u, kg = (unsigned)lane >> 4;
    unsigned xo[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) xo[kh] = r16 * 128u + ((((unsigned)(kh * 4) + kg) ^ ((r16 >> 1) & 7u)) << 4);
    const unsigned kj = r16 >> 2;
    const unsigned klane = kg * 2048u + kj * 256u + (r16 & 3u) * 8u;
    unsigned kxb[BKM ? NBN : 1];
    if constexpr (BKM) {
#pragma unroll
        for (int i = 0; i < NBN; ++i) {
            const unsigned blk = (unsigned)(wno + i);
            kxb[i] = klane + (((blk >> 1) ^ kj) << 6) + (((blk & 1u) ^ (kg & 1u)) << 5);
        }
    }
    const unsigned b_wave_off = BKM ? (unsigned)wnh * 16384u : (unsigned)wave * 8192u;

    bf16x8 fa[2][NBM], fb[2][NBN];
    constexpr int OPS_B = BKM ? 2 * NBN : NBN, NOPS = OPS_B + NBM;
    auto frag_op = [&](auto opc, auto khc, auto bufc, unsigned a_base, unsigned b_base) {
        constexpr int op = decltype(opc)::value, kh = decltype(khc)::value, buf = decltype(bufc)::value;
        if constexpr (op < OPS_B) {
            if constexpr (BKM) lds_read_tr64_h<kh * 8192>(fb[buf][op >> 1], b_base + kxb[op >> 1], op & 1);
            else lds_read_b128_v<op * 2048>(fb[buf][op], b_base + xo[kh]);
        } else {
            constexpr int o = op - OPS_B;
            lds_read_b128_v<o * 2048>(fa[buf][o], a_base + xo[kh]);
        }
    };
    // one half of a K-step: 44 MFMAs on buffer BUF; the next fragment set (k-half NKH of the slots na / nb_) is read into the other buffer in the
    // shadows of the first three quarters of them, the DMA pieces of the step two ahead (8 of B in half 0, 6 / 5 of A in half 1) in shadows
    // 3, 8, 13, ...
    constexpr int SL = NMF * 3 / 4;
    auto half_step = [&](auto bufc, auto nkhc, auto isAc, unsigned na, unsigned nb_, unsigned dslot, unsigned so, int krem) {
        constexpr int buf = decltype(bufc)::value;
        constexpr bool isA = decltype(isAc)::value != 0;
        static_for<0, NMF>([&](auto ic) {
            constexpr int i = decltype(ic)::value, tn = i / NBM, tm = i % NBM;
            mfma16(acc[tn][tm], fb[buf][tn], fa[buf][tm]);
            if constexpr (i < SL) {
                constexpr int o0 = (i * NOPS + SL - 1) / SL, o1 = ((i + 1) * NOPS + SL - 1) / SL;
                static_for<o0, (o1 < NOPS ? o1 : NOPS)>([&](auto oc) { frag_op(oc, nkhc, ic_<buf ^ 1>{}, na, nb_); });
            }
            if constexpr (i % 5 == 3 && i / 5 < (isA ? 6 : 8)) {
                if constexpr (isA) issueA(ic_<i / 5>{}, dslot, so, krem);
                else issueB(ic_<i / 5>{}, dslot, so, krem);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto b_next_slot = [](unsigned bb) { return bb == 0u ? (unsigned)B_SLOT : (bb == (unsigned)B_SLOT ? (unsigned)B2_OFF : 0u); };

    asm volatile("s_waitcnt vmcnt(13)" ::: "memory");          // step 0 landed (this wave's pieces: 8 + 6 or 5); step 1 may still be in flight
    while (true) {
        __builtin_amdgcn_s_barrier();
        static_for<0, NOPS>([&](auto oc) { frag_op(oc, ic_<0>{}, ic_<0>{}, lds0 + A_OFF, lds0 + b_wave_off); });
        unsigned bb = 0, ab = 0;                                     // slot offsets of step t in the B / A ring
        unsigned soa = 2u * soA1, sob = 2u * soB1;                   // scalar K-step offsets of step t + 2
        int krem = K - 2 * BK;
        for (int t = 0; t < nk; ++t) {
            const unsigned bb1 = b_next_slot(bb), bb2 = b_next_slot(bb1);
            const unsigned ab1 = (unsigned)A_SLOT - ab;
            const unsigned a_base = lds0 + A_OFF + ab, b_base = lds0 + bb + b_wave_off;
            const unsigned a_next = lds0 + A_OFF + ab1, b_next = lds0 + bb1 + b_wave_off;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // fragments (t, k-half 0)
            __builtin_amdgcn_sched_barrier(0);
            half_step(ic_<0>{}, ic_<1>{}, ic_<0>{}, a_base, b_base, bb2, sob, krem);
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // all but B(t+2) landed: step t+1 complete; step t's fragments all read
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            half_step(ic_<1>{}, ic_<0>{}, ic_<1>{}, a_next, b_next, ab, soa, krem);
            bb = bb1;
            ab = ab1;
            soa += soA1;
            sob += soB1;
            krem -= BK;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __syncthreads();                                             // every wave's fragment reads done: the whole ring is free
        const int cm0 = m0, cn0 = n0, ctile = tile_id;
        bool more = false;
        if constexpr (PERSIST) {
            v += gridDim.x;
            more = v < ntile;
            if (more) {
                // the next tile's first two K-steps fly while this tile's epilogue runs
                tile_of(v, m0, n0, tile_id);
                lane_offsets(m0, n0);
                prologue();
            }
        }
        const int Mlim = (cm0 + BMT) < M ? (cm0 + BMT) : M;          // rows of the short last pass beyond the tile belong to the next tile
        ring_epilogue<4, NBN, NBM, BKM, SWIGLU, PAIR, false, PB, R176_FASTSW>(acc, smem + STRIP_OFF, wave, tid, lane, wave, C, Mlim, N, ldc, bias, res, ldr, flags,
                                                                 cm0, cn0, ctile, aux0, aux1, aux_ld, aux_n);
        if (!more) break;
#pragma unroll
        for (int i = 0; i < NBN; ++i)
#pragma unroll
            for (int j = 0; j < NBM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the DMA pieces were issued in front of the epilogue's own loads and stores (vmcnt retires in order): draining the queue costs the
        // acknowledgement of the last stores, the pieces themselves landed microseconds ago
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// MANTIS_GEMM_176P (read once): 1 = persistent workgroups with the next tile's first K-steps prefetched under the epilogue (default), 0 = one tile
// per workgroup (A/B measurements; same results bit for bit)
static int ring176_persist() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MANTIS_GEMM_176P"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}
template <bool BKM, bool SWIGLU = false, int PAIR = PAIR_NONE>
static int launch_gemm_ring176(hipStream_t s, const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, long lda, long ldb, long ldc,
                               const bf16_t* bias, const bf16_t* res, long ldr, int flags, int cus, bf16_t* aux0 = nullptr,
                               const bf16_t* aux1 = nullptr, long aux_ld = 0, int aux_n = 0) {
    const int tiles_m = cdiv(M, 176), tiles_n = cdiv(N, 256), ntile = tiles_m * tiles_n;
    if (ring176_persist() && ntile > cus && !(flags & EPI_SHARED_GPU)) {
        MANTIS_LAUNCH((gemm_nt_ring176_kernel<BKM, SWIGLU, PAIR, true>), dim3(cus), dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, bias, res, ldr,
                           flags, tiles_m, tiles_n, aux0, aux1, aux_ld, aux_n);
    } else {
        MANTIS_LAUNCH((gemm_nt_ring176_kernel<BKM, SWIGLU, PAIR, false>), dim3(ntile), dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, bias, res,
                           ldr, flags, tiles_m, tiles_n, aux0, aux1, aux_ld, aux_n);
    }
    return mantis_check_launch();
}
// The one symbol gemm.hip links against (not part of the C-ABI: hidden visibility): kind = 0 plain epilogues (flags as mantis_gemm_bf16_nt), 1 / 2 the
// two-column forward fusions (PAIR_SWIGLU / PAIR_ROPE); bkm = B given K-major; flags & EPI_SWIGLU_BWD selects the fused SwiGLU backward (B K-major).
__attribute__((visibility("hidden"))) int mantis_launch_ring176(hipStream_t s, const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K,
                                                                 long lda, long ldb, long ldc, const bf16_t* bias, const bf16_t* res, long ldr,
                                                                 int flags, int bkm, int kind, int cus, bf16_t* aux0, const bf16_t* aux1,
                                                                 long aux_ld, int aux_n) {
#define R176_ARGS s, A, B, C, M, N, K, lda, ldb, ldc, bias, res, ldr, flags, cus, aux0, aux1, aux_ld, aux_n
    if (kind == PAIR_SWIGLU) return bkm ? MANTIS_EUNSUPPORTED : launch_gemm_ring176<false, false, PAIR_SWIGLU>(R176_ARGS);
    if (kind == PAIR_ROPE) return bkm ? MANTIS_EUNSUPPORTED : launch_gemm_ring176<false, false, PAIR_ROPE>(R176_ARGS);
    if (flags & EPI_SWIGLU_BWD) return bkm ? launch_gemm_ring176<true, true>(R176_ARGS) : MANTIS_EUNSUPPORTED;
    return bkm ? launch_gemm_ring176<true>(R176_ARGS) : launch_gemm_ring176<false>(R176_ARGS);
#undef R176_ARGS
}
