// Fused softmax attention for gfx950 (flash-style, MFMA, online softmax in fp32) -- forward (ViT windows + Llama causal
// GQA with key-padding mask) and backward (Llama).
//
// Replaces (reference path): the attention call inside SiglipAttention / CLIPAttention / LlamaAttention
//   transformers/models/siglip/modeling_siglip.py:227-247,267-305 (non-causal, one window per image)
//   transformers/models/llama/modeling_llama.py:191-214,262-276 (causal + padding mask, GQA repeat_kv)
// reached from /root/reference/mantis/models/mllava/modeling_llava.py:456 and :510, where the reference calls the
// flash-attn CUDA extension (train_mllava.py:79-82) or the eager softmax.  Never materialises the LxL score matrix.
//
// Layout choices (CDNA4):
//  * scores are computed TRANSPOSED, S^T = K.Q^T via v_mfma_f32_32x32x16_bf16 with K rows as the A operand, so a lane owns ONE
//    query column (lane & 31) and 16 keys per 32-key block: the row max / row sum are in-lane reductions plus a single
//    lane <-> lane+32 exchange, and the rescale factor is lane-uniform.
//  * the P (or dS) accumulator registers feed the next MFMA's B operand DIRECTLY: the contraction order over keys is
//    permuted to match the accumulator layout (keys {4h+e+8c}), and the A operand follows the same permutation.
//  * operands whose contraction index is not their contiguous axis (V for P.V, K for dS.K, Q and dO for dK/dV) are read from
//    the SAME row-major LDS tile with ds_read_b64_tr_b16 (hardware transposing read: a 16-lane group fetches a 4x16 block and
//    lane t receives column t) -- no transposed copies in HBM or LDS.
//  * K/V (or Q/dO) tiles are double buffered in LDS: the next tile's global loads are issued into registers before the MFMA
//    work of the current tile and written to the other buffer afterwards (one barrier per tile).
//  * causal work is dispatched heaviest-first; the dK/dV kernel runs one workgroup per (key block, QUERY head) and a small
//    reduction sums the GQA group, so the grid is H/Hkv times larger than a per-kv-head walk.
// Algorithmic FLOPs: forward 4*L*Lk*hd per head (half for causal); backward 2.5x forward (+1x recompute of S and dP here).

#include "attn_common.h"

// ------------------------------------------------------------------------------------------------ forward
// grid (ceil(L/128), H, B); 4 waves x 32 query rows; KV tiles of 64 keys, double buffered.
// (Three workgroups per CU for the small head dims -- their 53.8 KB of LDS would allow it -- was measured: the 170-VGPR budget spills
// ~19 registers and the kernel gets SLOWER: hd 80 864 -> 1367 us, hd 72 162 -> 243 us, hd 96 206 -> 324 us; hd 64, no spills: equal.)
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                       const bf16_t* __restrict__ V, const int* __restrict__ kmask,
                                                       bf16_t* __restrict__ O, float* __restrict__ LSE, int L, int Lk, int H, int Hkv,
                                                       long ldq, long ldk, long ldv, long ldo, float scale,
                                                       const int* __restrict__ kstart) {
    using C = AttnCfg<HD>;
    using Y = Lay<HD>;
    constexpr int TILE = 64 * Y::PITCH;
    constexpr int BUF = 2 * TILE + 64 * 4 + 16;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5, lq = lane & 31;
    const int gx = (L + 127) >> 7;
    int bx, h, b;
    xcd_tile_map(gx, H, bx, h, b, CAUSAL);
    const int hk = h / (H / Hkv);
    const int qb = CAUSAL ? (gx - 1 - bx) : bx;   // causal: longest rows first
    const int qblk0 = qb * 128, q0 = qblk0 + wave * 32, q = q0 + lq;
    const int qc = q < L ? q : L - 1;
    const float c = scale * LOG2E;
    // packed samples: query q attends keys >= kstart[q] only (its own sample; non-decreasing in q).  ks_q = this lane's bound,
    // ks_hi = the largest bound in the wave's 32 rows (tiles below it need per-element masking), t_first = first tile any row of
    // the workgroup's 128-query block can see
    const int ks_q = kstart ? kstart[(long)b * L + qc] : 0;
    const int ks_hi = kstart ? kstart[(long)b * L + (q0 + 31 < L ? q0 + 31 : L - 1)] : 0;
    const int t_first = kstart ? (kstart[(long)b * L + (qblk0 < L ? qblk0 : L - 1)] >> 6) : 0;

    bf16x8 qf[C::NKS];
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
        const int ch = ks * 2 + hh;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ch * 8 < HD) v = *reinterpret_cast<const u32x4*>(Q + ((long)b * L + qc) * ldq + (long)h * HD + ch * 8);
        qf[ks] = as_bf16x8(v);
    }
    f32x16 oacc[C::NDB];
#pragma unroll
    for (int d = 0; d < C::NDB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int kend = CAUSAL ? (qblk0 + 128 < L ? qblk0 + 128 : L) : Lk;      // Lk: number of keys (== L except for cross attention)
    const int ntiles = (kend + 63) / 64;
    const bf16_t* Kb = K + (long)b * Lk * ldk + (long)hk * HD;
    const bf16_t* Vb = V + (long)b * Lk * ldv + (long)hk * HD;

    // tile t -> LDS buffer (t & 1): global -> registers -> LDS in one go (registers are only live across the copy, so the kernel
    // fits 2 waves per SIMD; the copy of tile t+1 overlaps the MFMA work of the co-resident workgroup / partner waves)
    // the key-mask word of a tile is fetched one stage() call ahead of its use (stage calls walk consecutive tiles): a load issued
    // here, behind the tile's DMA, would be waited for with vmcnt(0) -- wave 0 would sit out the whole DMA round trip on every tile
    // (measured on the Llama-3 geometry, round 3: forward 169 us without a key mask, 207 us with one)
    auto key_live = [&](int t) -> bool {
        const int key = t * 64 + (int)threadIdx.x;
        return threadIdx.x < 64 && key < Lk && (kmask == nullptr || kmask[(long)b * Lk + key] != 0);
    };
    bool live_next = key_live(t_first);
    auto stage = [&](int t) {
        const int key0 = t * 64;
        char* base = smem + (t & 1) * BUF;
        const bool ok = live_next;
        live_next = key_live(t + 1);
        if constexpr (Y::DMA) {
            stage_tile_dma<64>(Kb + (long)key0 * ldk, ldk, Lk - key0, base);
            stage_tile_dma<64>(Vb + (long)key0 * ldv, ldv, Lk - key0, base + TILE);
        } else {
            {
                TileRegs<64, C::NCH> rk;
                rk.load(Kb + (long)key0 * ldk, ldk, Lk - key0, HD);
                rk.store(base, C::PITCH);
            }
            {
                TileRegs<64, C::NCH> rv;
                rv.load(Vb + (long)key0 * ldv, ldv, Lk - key0, HD);
                rv.store(base + TILE, C::PITCH);
            }
        }
        if (threadIdx.x < 64) {     // wave 0: additive key bias (0 / -inf) + one flag "this tile has a masked key"
            float* bp = reinterpret_cast<float*>(base + 2 * TILE);
            bp[threadIdx.x] = ok ? 0.f : -INFINITY;
            const unsigned long long okm = __ballot(ok);
            if (threadIdx.x == 0) bp[64] = (okm == ~0ull) ? 0.f : 1.f;
        }
    };
    // register-staged layouts (hd != 128): the next tile's global loads are issued BEFORE this tile's MFMA / softmax work and written to
    // LDS AFTER it, so the HBM / L2 latency rides behind the compute instead of stalling the wave in front of it (24 VGPRs at hd 72 /
    // 80 / 96); the key bias words of the next tile go to its (free) LDS buffer right away
    auto stage_bias = [&](int t) {
        const bool ok = live_next;
        live_next = key_live(t + 1);
        if (threadIdx.x < 64) {
            float* bp = reinterpret_cast<float*>(smem + (t & 1) * BUF + 2 * TILE);
            bp[threadIdx.x] = ok ? 0.f : -INFINITY;
            const unsigned long long okm = __ballot(ok);
            if (threadIdx.x == 0) bp[64] = (okm == ~0ull) ? 0.f : 1.f;
        }
    };
    stage(t_first);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = t_first; t < ntiles; ++t) {
        const int key0 = t * 64;
        const char* sK = smem + (t & 1) * BUF;
        const char* sV = sK + TILE;
        const float* sBias = reinterpret_cast<const float*>(sK + 2 * TILE);
        const bool more = t + 1 < ntiles;
        TileRegs<64, C::NCH> nk_regs, nv_regs;          // (unused on the one-go paths)
        // hd <= 64 keeps the one-go copy: it runs 3 waves per SIMD at 168 VGPRs, which the 24 staging registers would cost
        // (measured: CLIP hd 64 29.3 -> 36.8 us with the split; hd 72 / 80 / 96, at 2 waves either way: -9 ... -16 %)
        constexpr bool SPLIT_STAGE = !Y::DMA && HD > 64;
        if constexpr (!SPLIT_STAGE) {
            if (more) stage(t + 1);
        } else {
            if (more) {
                const int kn = (t + 1) * 64;
                nk_regs.load(Kb + (long)kn * ldk, ldk, Lk - kn, HD);
                nv_regs.load(Vb + (long)kn * ldv, ldv, Lk - kn, HD);
                stage_bias(t + 1);
            }
        }
        if (!(CAUSAL && key0 > q0 + 31)) {   // wave-uniform: skip tiles entirely in this wave's future (tiles wholly before the
                                             // wave's samples are harmless: fully masked, p = 0, running max stays -inf)
            f32x16 s[2];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[sb][e] = 0.f;
            constexpr int FD = ATTN_FRAG_DEPTH;
            FragU fb4[8];                       // fragment ring (FD in flight) shared by the S phase (K rows) and the P.V phase (V^T)
            unsigned vaddr[4], vaddr8[4];
            if constexpr (Y::DMA) {
                const unsigned ldsK = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)sK;
                unsigned kaddr[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) kaddr[ks] = ldsK + Y::chunk_off(lq, ks * 2 + hh);
                {
                    const int sl = lane & 15, g16 = (lane >> 4) & 1;
                    const int row = 4 * hh + (sl >> 2);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int col = d * 32 + 16 * g16 + (sl & 3) * 4;
                        const int off = Y::chunk_off(row, col >> 3) + (col & 7) * 2;
                        vaddr[d] = ldsK + TILE + off;
                        vaddr8[d] = ldsK + TILE + ((off + 8 * Y::PITCH) ^ 32);
                    }
                }
                // S = K.Q^T: 16 MFMAs, K fragment i + 3 requested in the shadow of MFMA i
                // slot i computes key block i & 1, k-chunk i >> 1: consecutive MFMAs alternate between the two score accumulators (round 3:
                // eight back-to-back MFMAs into ONE accumulator wait for each other's result -- PMC: 32 % of the wave cycles were
                // MFMA-dependency stalls -- ATTN_S_INTERLEAVE=0 restores the old order for A/B)
#ifndef ATTN_S_INTERLEAVE
#define ATTN_S_INTERLEAVE 1
#endif
                static_for<0, FD>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, fr = ATTN_S_INTERLEAVE ? (i & 1) * 8 + (i >> 1) : i;
                    issue_kfrag<fr>(fb4[i & 7].f, kaddr);
                });
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int sb = ATTN_S_INTERLEAVE ? (i & 1) : (i >> 3), ks = ATTN_S_INTERLEAVE ? (i >> 1) : (i & 7);
                    wait_lgkm<(15 - i) < FD - 1 ? (15 - i) : FD - 1>();      // requested so far: fragments <= i + FD - 1
                    __builtin_amdgcn_sched_barrier(0);
                    s[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb4[i & 7].f, qf[ks], s[sb], 0, 0, 0);
                    if constexpr (i + FD < 16) {
                        constexpr int n = i + FD, fr = ATTN_S_INTERLEAVE ? (n & 1) * 8 + (n >> 1) : n;
                        issue_kfrag<fr>(fb4[n & 7].f, kaddr);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // the first FD V^T fragments travel while the softmax runs
                static_for<0, FD>([&](auto jc) { issue_vfrag<decltype(jc)::value>(fb4[decltype(jc)::value & 7], vaddr, vaddr8); });
            } else {
#pragma unroll
                for (int ks = 0; ks < C::NKS; ++ks)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {        // alternate the two accumulators (see the hd-128 branch)
                        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + Y::chunk_off(sb * 32 + lq, ks * 2 + hh));
                        s[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[sb], 0, 0, 0);
                    }
            }
            float mx = -INFINITY;
            // wave-uniform: only tiles that touch the diagonal or contain masked keys pay for per-element masking
            const bool need_mask = (CAUSAL && key0 + 63 > q0) || sBias[64] != 0.f || key0 < ks_hi;
            if (need_mask) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        float v = s[sb][r] * c + sBias[kl];
                        if (CAUSAL && key0 + kl > q) v = -INFINITY;
                        if (key0 + kl < ks_q) v = -INFINITY;
                        s[sb][r] = v;
                        mx = fmaxf(mx, v);
                    }
            } else {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sb][r]);      // raw scores: the scale is folded into the exp below
                mx *= c;                                                        // c > 0: max commutes with the scaling
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
            if (need_mask) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(s[sb][r] - msafe);
                        s[sb][r] = p;
                        psum += p;
                    }
            } else {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sb][r], c, -msafe));
                        s[sb][r] = p;
                        psum += p;
                    }
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
            if (__any(alpha != 1.f)) {     // wave-uniform: once the running maxima have settled nothing needs rescaling
#pragma unroll
                for (int d = 0; d < C::NDB; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
            }
            if constexpr (Y::DMA) {
                // O^T += V^T.P: 16 MFMAs, V^T fragment j + 3 (two transposing reads) requested in the shadow of MFMA j
                bf16x8 pfr[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) pfr[g] = pack_frag(s[g >> 1], g & 1);
                static_for<0, 16>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    wait_lgkm<((15 - j) < FD - 1 ? (15 - j) : FD - 1) * 2>();      // requested so far: fragments <= j + FD - 1 (two reads each)
                    __builtin_amdgcn_sched_barrier(0);
                    oacc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb4[j & 7].f, pfr[j >> 2], oacc[j & 3], 0, 0, 0);
                    if constexpr (j + FD < 16) issue_vfrag<j + FD>(fb4[(j + FD) & 7], vaddr, vaddr8);
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int cp = 0; cp < 2; ++cp) {
                        const bf16x8 pf = pack_frag(s[sb], cp);
#pragma unroll
                        for (int d = 0; d < C::NDB; ++d) {
                            const bf16x8 vf = read_tr_frag<HD>(sV, sb * 32 + 16 * cp, d * 32, lane);
                            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
                        }
                    }
            }
        }
        if constexpr (SPLIT_STAGE) {
            if (more) {
                char* nb = smem + ((t + 1) & 1) * BUF;      // released by the barrier that ended iteration t - 1
                nk_regs.store(nb, C::PITCH);
                nv_regs.store(nb + TILE, C::PITCH);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile's DMA landed (this wave's pieces)
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (q < L) {
        bf16_t* op = O + ((long)b * L + q) * ldo + (long)h * HD;
#pragma unroll
        for (int d = 0; d < C::NDB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = d * 32 + 8 * g4 + 4 * hh;
                if (dd < HD) {
                    u32x2 o;
                    o[0] = pack_bf2(oacc[d][4 * g4] * inv, oacc[d][4 * g4 + 1] * inv);
                    o[1] = pack_bf2(oacc[d][4 * g4 + 2] * inv, oacc[d][4 * g4 + 3] * inv);
                    *reinterpret_cast<u32x2*>(op + dd) = o;
                }
            }
        if (hh == 0 && LSE) LSE[((long)b * H + h) * L + q] = l_tot > 0.f ? m_run * LN2 + logf(l_tot) : INFINITY;
    }
}

// ------------------------------------------------------------------------------------------------ backward: D = rowsum(dO * O)
__global__ void attn_dsum_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O, float* __restrict__ Dsum, long rows,
                                 int H, int HD, int L, long ldo) {
    // one wave per (token row, head); Dsum layout [B, H, L]
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= rows * H) return;
    const long row = w / H;
    const int h = (int)(w - row * H);
    float s = 0.f;
    for (int c = lane; c < HD / 8; c += 64) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(dO + row * ldo + (long)h * HD + c * 8);
        const u32x4 o = *reinterpret_cast<const u32x4*>(O + row * ldo + (long)h * HD + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += bf2f_lo(a[e]) * bf2f_lo(o[e]) + bf2f_hi(a[e]) * bf2f_hi(o[e]);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const long b = row / L, l = row - b * L;
        Dsum[(b * H + h) * L + l] = s;
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// Same walk as the forward.  dQ^T[d][q] += K^T[d][key] . dS^T[key][q]  (K^T fragments: transposing reads of the K tile).
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                          const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                          const int* __restrict__ kmask, const float* __restrict__ LSE,
                                                          float* __restrict__ Dsum, bf16_t* __restrict__ dQ, int L, int Lk, int H,
                                                          int Hkv, long ldq, long ldk, long ldv, long ldo, long lddq, float scale,
                                                          const bf16_t* __restrict__ Ofwd, long ldout,
                                                          const int* __restrict__ kstart) {
    using C = AttnCfg<HD>;
    using Y = Lay<HD>;
    constexpr int TILE = 64 * Y::PITCH;
    constexpr int BUF = 2 * TILE + 64 * 4 + 16;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5, lq = lane & 31;
    const int gx = (L + 127) >> 7;
    int bx, h, b;
    xcd_tile_map(gx, H, bx, h, b, CAUSAL);
    const int hk = h / (H / Hkv);
    const int qb = CAUSAL ? (gx - 1 - bx) : bx;
    const int qblk0 = qb * 128, q0 = qblk0 + wave * 32, q = q0 + lq;
    const int qc = q < L ? q : L - 1;
    const float c = scale * LOG2E;
    // segment bounds of packed samples: see attn_fwd_kernel
    const int ks_q = kstart ? kstart[(long)b * L + qc] : 0;
    const int ks_hi = kstart ? kstart[(long)b * L + (q0 + 31 < L ? q0 + 31 : L - 1)] : 0;
    const int t_first = kstart ? (kstart[(long)b * L + (qblk0 < L ? qblk0 : L - 1)] >> 6) : 0;

    bf16x8 qf[C::NKS], dof[C::NKS];
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
        const int ch = ks * 2 + hh;
        u32x4 v = {0u, 0u, 0u, 0u}, w = {0u, 0u, 0u, 0u};
        if (ch * 8 < HD) {
            v = *reinterpret_cast<const u32x4*>(Q + ((long)b * L + qc) * ldq + (long)h * HD + ch * 8);
            w = *reinterpret_cast<const u32x4*>(dO + ((long)b * L + qc) * ldo + (long)h * HD + ch * 8);
        }
        qf[ks] = as_bf16x8(v);
        dof[ks] = as_bf16x8(w);
    }
    const float lse2 = LSE[((long)b * H + h) * L + qc] * LOG2E;
    // D = rowsum(dO * O): given, or (Ofwd != nullptr) computed here from the dO fragments already in registers -- the lane pair
    // (l, l + 32) covers the row's head dimension -- and published for the dK/dV kernel that runs after this one
    float dsum;
    if (Ofwd != nullptr) {
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) {
            const int ch = ks * 2 + hh;
            if (ch * 8 < HD) {
                const u32x4 o = *reinterpret_cast<const u32x4*>(Ofwd + ((long)b * L + qc) * ldout + (long)h * HD + ch * 8);
                union { bf16x8 f; u32x4 u; } dv;
                dv.f = dof[ks];
#pragma unroll
                for (int e = 0; e < 4; ++e) part += bf2f_lo(o[e]) * bf2f_lo(dv.u[e]) + bf2f_hi(o[e]) * bf2f_hi(dv.u[e]);
            }
        }
        dsum = part + __shfl_xor(part, 32, 64);
        if (hh == 0 && q < L) Dsum[((long)b * H + h) * L + q] = dsum;
    } else {
        dsum = Dsum[((long)b * H + h) * L + qc];
    }
    f32x16 acc[C::NDB];
#pragma unroll
    for (int d = 0; d < C::NDB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;

    const int kend = CAUSAL ? (qblk0 + 128 < L ? qblk0 + 128 : L) : Lk;
    const int ntiles = (kend + 63) / 64;
    const bf16_t* Kb = K + (long)b * Lk * ldk + (long)hk * HD;
    const bf16_t* Vb = V + (long)b * Lk * ldv + (long)hk * HD;

    // tile t -> LDS buffer (t & 1): global -> registers -> LDS in one go (registers are only live across the copy, so the kernel
    // fits 2 waves per SIMD; the copy of tile t+1 overlaps the MFMA work of the co-resident workgroup / partner waves)
    // the key-mask word of a tile is fetched one stage() call ahead of its use (stage calls walk consecutive tiles): a load issued
    // here, behind the tile's DMA, would be waited for with vmcnt(0) -- wave 0 would sit out the whole DMA round trip on every tile
    // (measured on the Llama-3 geometry, round 3: forward 169 us without a key mask, 207 us with one)
    auto key_live = [&](int t) -> bool {
        const int key = t * 64 + (int)threadIdx.x;
        return threadIdx.x < 64 && key < Lk && (kmask == nullptr || kmask[(long)b * Lk + key] != 0);
    };
    bool live_next = key_live(t_first);
    auto stage = [&](int t) {
        const int key0 = t * 64;
        char* base = smem + (t & 1) * BUF;
        const bool ok = live_next;
        live_next = key_live(t + 1);
        if constexpr (Y::DMA) {
            stage_tile_dma<64>(Kb + (long)key0 * ldk, ldk, Lk - key0, base);
            stage_tile_dma<64>(Vb + (long)key0 * ldv, ldv, Lk - key0, base + TILE);
        } else {
            {
                TileRegs<64, C::NCH> rk;
                rk.load(Kb + (long)key0 * ldk, ldk, Lk - key0, HD);
                rk.store(base, C::PITCH);
            }
            {
                TileRegs<64, C::NCH> rv;
                rv.load(Vb + (long)key0 * ldv, ldv, Lk - key0, HD);
                rv.store(base + TILE, C::PITCH);
            }
        }
        if (threadIdx.x < 64) {     // wave 0: additive key bias (0 / -inf) + one flag "this tile has a masked key"
            float* bp = reinterpret_cast<float*>(base + 2 * TILE);
            bp[threadIdx.x] = ok ? 0.f : -INFINITY;
            const unsigned long long okm = __ballot(ok);
            if (threadIdx.x == 0) bp[64] = (okm == ~0ull) ? 0.f : 1.f;
        }
    };
    stage(t_first);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = t_first; t < ntiles; ++t) {
        const int key0 = t * 64;
        const char* sK = smem + (t & 1) * BUF;
        const char* sV = sK + TILE;
        const float* sBias = reinterpret_cast<const float*>(sK + 2 * TILE);
        const bool more = t + 1 < ntiles;
        if (more) stage(t + 1);
        if (!(CAUSAL && key0 > q0 + 31)) {
            const bool need_mask = (CAUSAL && key0 + 63 > q0) || sBias[64] != 0.f || key0 < ks_hi;   // wave-uniform
            // one 32-key half at a time: S, dP -> dS -> dQ contribution, so only 32 score registers are live (2 waves per SIMD)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                f32x16 s, dp;
#pragma unroll
                for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < C::NKS; ++ks) {
                    const int off = Y::chunk_off(sb * 32 + lq, ks * 2 + hh);
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + off);
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + off);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp, 0, 0, 0);
                }
                if (need_mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kl = sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        float v = s[r] * c + sBias[kl];
                        if (CAUSAL && key0 + kl > q) v = -INFINITY;
                        if (key0 + kl < ks_q) v = -INFINITY;
                        const float p = __builtin_amdgcn_exp2f(v - lse2);  // lse = +inf for fully masked rows -> p = 0
                        s[r] = p * (dp[r] - dsum);             // the softmax scale is applied once, to the dQ accumulators
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(s[r] * c - lse2);
                        s[r] = p * (dp[r] - dsum);             // the softmax scale is applied once, to the dQ accumulators
                    }
                }
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    const bf16x8 dsf = pack_frag(s, cp);
                    if constexpr (kDqTrAsm && Y::DMA) {
                        // hd 128: hand-issued K^T reads (the builtin's s_waitcnt vmcnt(0) would wait for the NEXT tile's DMA right here)
                        bf16x8 kt[4];
                        read_tr_frag4_sync(sK, sb * 32 + 16 * cp, lane, kt);
#pragma unroll
                        for (int d = 0; d < 4; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[d], dsf, acc[d], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int d = 0; d < C::NDB; ++d) {
                            const bf16x8 ktf = read_tr_frag<HD>(sK, sb * 32 + 16 * cp, d * 32, lane);
                            acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, acc[d], 0, 0, 0);
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (q < L) {
        bf16_t* op = dQ + ((long)b * L + q) * lddq + (long)h * HD;
#pragma unroll
        for (int d = 0; d < C::NDB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = d * 32 + 8 * g4 + 4 * hh;
                if (dd < HD) {
                    u32x2 o;
                    o[0] = pack_bf2(acc[d][4 * g4] * scale, acc[d][4 * g4 + 1] * scale);
                    o[1] = pack_bf2(acc[d][4 * g4 + 2] * scale, acc[d][4 * g4 + 3] * scale);
                    *reinterpret_cast<u32x2*>(op + dd) = o;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// One workgroup per (key block, QUERY head, batch); GQA groups are summed afterwards.  A wave owns 32 keys and -- to keep the
// accumulators at 64 registers so that two waves fit per SIMD -- HALF of the head dimension when hd >= 64: waves (kg, dh)
// with the same key group kg recompute S and dP (cheap next to the exposed latency of 1 wave/SIMD) and each accumulates its
// own d-half of dK and dV.  Walks 32-row query tiles, double buffered.
//   S[q][key]  = Q . K^T      (lane owns ONE key column, 16 query rows per block)
//   dV^T[d][key] += dO^T[d][q] . P[q][key]        dK^T[d][key] += Q^T[d][q] . dS[q][key]
// dKp/dVp: per-QUERY-head outputs [B*L, H*hd] (row stride ldp), or the final dK/dV when H == Hkv.
template <int HD>
struct DkvCfg {
    static constexpr int DS = (AttnCfg<HD>::NDB >= 2 && AttnCfg<HD>::NDB % 2 == 0) ? 2 : 1;   // waves sharing a key group (d-split)
    static constexpr int NDW = AttnCfg<HD>::NDB / DS;          // 32-wide d-blocks per wave
    static constexpr int KEYS = (4 / DS) * 32;                 // keys per workgroup
};

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                              const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                              const int* __restrict__ kmask, const float* __restrict__ LSE,
                                                              const float* __restrict__ Dsum, bf16_t* __restrict__ dKp,
                                                              bf16_t* __restrict__ dVp, int L, int Lk, int H, int Hkv, long ldq, long ldk,
                                                              long ldv, long ldo, long ldpk, long ldpv, float scale,
                                                              const int* __restrict__ kstart, const int* __restrict__ qend) {
    using C = AttnCfg<HD>;
    using D = DkvCfg<HD>;
    constexpr int QT = 64;                          // query rows staged per barrier (processed as two 32-row passes)
    using Y = Lay<HD>;
    constexpr int TILE = QT * Y::PITCH;
    constexpr int BUF = 2 * TILE + 3 * QT * 4;      // Q tile | dO tile | lse2[QT] dsum[QT] kstart[QT]
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5, lk = lane & 31;
    const int kg = wave / D::DS, dh = wave % D::DS;
    int bx, h, b;
    xcd_tile_map((Lk + D::KEYS - 1) / D::KEYS, H, bx, h, b, CAUSAL);  // L queries, Lk keys (== L except for cross attention)
    const int hk = h / (H / Hkv);
    const int kblk0 = bx * D::KEYS, k0 = kblk0 + kg * 32, key = k0 + lk;   // causal: key block 0 is the heaviest, first
    const int keyc = key < Lk ? key : Lk - 1;
    const float c = scale * LOG2E;
    const bool key_ok = key < Lk && (kmask == nullptr || kmask[(long)b * Lk + keyc] != 0);

    bf16x8 kf[C::NKS], vf[C::NKS];
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
        const int ch = ks * 2 + hh;
        u32x4 a = {0u, 0u, 0u, 0u}, w = {0u, 0u, 0u, 0u};
        if (ch * 8 < HD) {
            a = *reinterpret_cast<const u32x4*>(K + ((long)b * Lk + keyc) * ldk + (long)hk * HD + ch * 8);
            w = *reinterpret_cast<const u32x4*>(V + ((long)b * Lk + keyc) * ldv + (long)hk * HD + ch * 8);
        }
        kf[ks] = as_bf16x8(a);
        vf[ks] = as_bf16x8(w);
    }
    f32x16 dkacc[D::NDW], dvacc[D::NDW];
#pragma unroll
    for (int d = 0; d < D::NDW; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dkacc[d][e] = 0.f; dvacc[d][e] = 0.f; }

    const int qstart = CAUSAL ? (kblk0 / QT) : 0;  // first query tile that can see this key block
    int qlim = L;                                   // packed samples: no query at or beyond the end of these keys' sample sees them
    if (qend != nullptr) {
        __shared__ int s_qlim;
        if (threadIdx.x == 0) s_qlim = 0;
        __syncthreads();
        if (threadIdx.x < D::KEYS && kblk0 + (int)threadIdx.x < Lk) atomicMax(&s_qlim, qend[(long)b * Lk + kblk0 + threadIdx.x]);
        __syncthreads();
        qlim = s_qlim < L ? s_qlim : L;
    }
    const int nqt = (qlim + QT - 1) / QT;
    const bf16_t* Qb = Q + (long)b * L * ldq + (long)h * HD;
    const bf16_t* dOb = dO + (long)b * L * ldo + (long)h * HD;

    auto stage = [&](int qt) {
        const int q0 = qt * QT;
        char* base = smem + ((qt - qstart) & 1) * BUF;
        if constexpr (Y::DMA) {
            stage_tile_dma<QT>(Qb + (long)q0 * ldq, ldq, L - q0, base);
            stage_tile_dma<QT>(dOb + (long)q0 * ldo, ldo, L - q0, base + TILE);
        } else {
            {
                TileRegs<QT, C::NCH> rq;
                rq.load(Qb + (long)q0 * ldq, ldq, L - q0, HD);
                rq.store(base, C::PITCH);
            }
            {
                TileRegs<QT, C::NCH> rdo;
                rdo.load(dOb + (long)q0 * ldo, ldo, L - q0, HD);
                rdo.store(base + TILE, C::PITCH);
            }
        }
        if (threadIdx.x < QT) {
            const int qq = q0 + threadIdx.x;
            reinterpret_cast<float*>(base + 2 * TILE)[threadIdx.x] = qq < L ? LSE[((long)b * H + h) * L + qq] * LOG2E : INFINITY;
            reinterpret_cast<float*>(base + 2 * TILE)[QT + threadIdx.x] = qq < L ? Dsum[((long)b * H + h) * L + qq] : 0.f;
            // rows beyond L: "sees no key" (keeps the wave-uniform need_mask test below, which looks at the pass's LAST row, sound)
            reinterpret_cast<int*>(base + 2 * TILE)[2 * QT + threadIdx.x] =
                kstart == nullptr ? 0 : (qq < L ? kstart[(long)b * L + qq] : 0x7fffffff);
        }
    };
    if (qstart < nqt) stage(qstart);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int qt = qstart; qt < nqt; ++qt) {
        const char* tQ = smem + ((qt - qstart) & 1) * BUF;
        if (qt + 1 < nqt) stage(qt + 1);
#pragma unroll 1
        for (int pass = 0; pass < QT / 32; ++pass) {
        const int q0 = qt * QT + pass * 32;
        const char* sQ = tQ + pass * 32 * Y::PITCH;
        const char* sdO = sQ + TILE;
        const float* sLse = reinterpret_cast<const float*>(tQ + 2 * TILE) + pass * 32;
        const float* sDs = sLse + QT;
        const int* sKs = reinterpret_cast<const int*>(tQ + 2 * TILE) + 2 * QT + pass * 32;
        if (!(CAUSAL && q0 + 31 < k0)) {  // wave-uniform: skip passes whose every query precedes this wave's keys
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
                const int off = Y::chunk_off(lk, ks * 2 + hh);
                const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sQ + off);
                const bf16x8 da = *reinterpret_cast<const bf16x8*>(sdO + off);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
            }
            f32x16 ds;
            const bool need_mask = (CAUSAL && k0 + 31 > q0) || !__all(key_ok) || sKs[31] > k0;   // wave-uniform
            if (need_mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    float v = key_ok ? s[r] * c : -INFINITY;
                    if (CAUSAL && key > q0 + ql) v = -INFINITY;
                    if (key < sKs[ql]) v = -INFINITY;
                    const float p = __builtin_amdgcn_exp2f(v - sLse[ql]);
                    s[r] = p;
                    ds[r] = p * (dp[r] - sDs[ql]);       // the softmax scale is applied once, to the dK accumulators
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const float p = __builtin_amdgcn_exp2f(s[r] * c - sLse[ql]);
                    s[r] = p;
                    ds[r] = p * (dp[r] - sDs[ql]);       // the softmax scale is applied once, to the dK accumulators
                }
            }
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                const bf16x8 pf = pack_frag(s, cp);
                const bf16x8 dsf = pack_frag(ds, cp);
#pragma unroll
                for (int d = 0; d < D::NDW; ++d) {
                    const int dcol = (dh * D::NDW + d) * 32;
                    const bf16x8 dot = read_tr_frag<HD>(sdO, 16 * cp, dcol, lane);
                    const bf16x8 qtf = read_tr_frag<HD>(sQ, 16 * cp, dcol, lane);
                    dvacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pf, dvacc[d], 0, 0, 0);
                    dkacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dkacc[d], 0, 0, 0);
                }
            }
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (key < Lk) {
        bf16_t* kp = dKp + ((long)b * Lk + key) * ldpk + (long)(H == Hkv ? hk : h) * HD;
        bf16_t* vp = dVp + ((long)b * Lk + key) * ldpv + (long)(H == Hkv ? hk : h) * HD;
#pragma unroll
        for (int d = 0; d < D::NDW; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = (dh * D::NDW + d) * 32 + 8 * g4 + 4 * hh;
                if (dd < HD) {
                    u32x2 o, w;
                    o[0] = pack_bf2(dkacc[d][4 * g4] * scale, dkacc[d][4 * g4 + 1] * scale);
                    o[1] = pack_bf2(dkacc[d][4 * g4 + 2] * scale, dkacc[d][4 * g4 + 3] * scale);
                    w[0] = pack_bf2(dvacc[d][4 * g4], dvacc[d][4 * g4 + 1]);
                    w[1] = pack_bf2(dvacc[d][4 * g4 + 2], dvacc[d][4 * g4 + 3]);
                    *reinterpret_cast<u32x2*>(kp + dd) = o;
                    *reinterpret_cast<u32x2*>(vp + dd) = w;
                }
            }
    }
}

// dK[m, hk*hd + d] = sum_g dKp[m, (hk*G + g)*hd + d]  (and dV): fp32 sum of the G per-query-head partials, 16 B per lane
__global__ void attn_group_reduce_kernel(const bf16_t* __restrict__ pk, const bf16_t* __restrict__ pv, bf16_t* __restrict__ dK,
                                         bf16_t* __restrict__ dV, long rows, int Hkv, int G, int HD, long ldp, long lddk, long lddv) {
    const int cph = HD >> 3;
    const long total = rows * Hkv * cph;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cph);
        const long t = i / cph;
        const int hk = (int)(t % Hkv);
        const long r = t / Hkv;
        float ak[8], av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ak[e] = 0.f; av[e] = 0.f; }
        for (int g = 0; g < G; ++g) {
            const long off = r * ldp + (long)(hk * G + g) * HD + cc * 8;
            const u32x4 a = *reinterpret_cast<const u32x4*>(pk + off);
            const u32x4 w = *reinterpret_cast<const u32x4*>(pv + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ak[2 * e] += bf2f_lo(a[e]); ak[2 * e + 1] += bf2f_hi(a[e]);
                av[2 * e] += bf2f_lo(w[e]); av[2 * e + 1] += bf2f_hi(w[e]);
            }
        }
        u32x4 ok, ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ok[e] = pack_bf2(ak[2 * e], ak[2 * e + 1]); ov[e] = pack_bf2(av[2 * e], av[2 * e + 1]); }
        *reinterpret_cast<u32x4*>(dK + r * lddk + (long)hk * HD + cc * 8) = ok;
        *reinterpret_cast<u32x4*>(dV + r * lddv + (long)hk * HD + cc * 8) = ov;
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV for GQA 4:1, hd 128
// One workgroup per (64-key block, KV head, batch); NO per-query-head partials in HBM, no group-reduce pass, S and dP computed once.
// Wave (kb, hp): kb = which 32 keys of the block, hp = which PAIR of the group's 4 query heads.  A wave owns its 32 keys over the
// FULL head dimension (dK and dV accumulators: 128 registers, K and V fragments: 64 registers -> one wave per SIMD, 512-register
// budget) and walks the 32-row query tiles of its two heads back to back as one stream; the two waves of a stream share the
// Q / dO tiles (LDS-DMA into a 3-stage ring, requested two tiles ahead, one barrier per tile).  The loop is software pipelined by
// hand: while the VALU turns tile i's S, dP into P and dS, the matrix pipe already computes S, dP of tile i+1.
//   MFMAs per (32 keys x 32 queries x head): S 8 + dP 8 + dV 8 + dK 8 = 32   (the per-query-head kernel above: 48, + the reduce)
// At the end the two streams' partial sums meet in LDS: wave hp = 0 finishes dK, wave hp = 1 finishes dV (fixed order: deterministic).
// Work per workgroup is at most (L/32 tiles x 4 heads x 2 key groups) / 4 waves, far below the per-CU average, so the causal
// imbalance is absorbed by the dispatch order (heaviest key block first) -- the reason the key block is 64 and not 128.
// Optional segment bounds for packed samples: kstart[b, q] = first key position query q may attend (its sample's start),
// qend[b, key] = one past the last query position that may attend key (its sample's end).
#ifdef DKV_STAMPS
// timing probe (tools/build_probe_lib.sh attn dkvstamps -DDKV_STAMPS; tools/attn_dkv_anatomy.py; never in the product build): wave 0 of every
// workgroup sums, over the steps of its software-pipelined loop, the s_memtime intervals step top -> DMA of tile j + 3 issued, first fragments
// requested -> MFMA 15 (S, dP of tile j + 2) -> MFMA 31 (dV, dK of tile j) -> behind the step's barrier; plus entry -> loop, entry -> exit.
// (s_memtime returns through lgkmcnt: every stamp also drains the LDS reads in flight -- read the PROPORTIONS.)
__device__ unsigned long long g_dkv_stamps[4096 * 8];
#define DKV_STAMP(var) do { var = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DKV_STAMP(var) do { } while (0)
#endif
template <bool CAUSAL, bool SEG>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_g4_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
    const int* __restrict__ kmask, const int* __restrict__ kstart, const int* __restrict__ qend, const float* __restrict__ LSE,
    const float* __restrict__ Dsum, bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, int L, int H, int Hkv, long ldq, long ldk,
    long ldv, long ldo, long lddk, long lddv, float scale) {
    constexpr int HD = 128, QT = 32, NKS = 8, NDB = 4;
    using Y = Lay<HD>;
    constexpr int TILE = QT * 256;                 // 8 KiB: 32 rows x 128 bf16
    constexpr int STREAM = 2 * TILE + 512;         // Q tile | dO tile | lse2[32] dsum[32] kstart[32] (+pad: keeps tile bases 256-B aligned)
    constexpr int STAGE = 2 * STREAM;              // the two head-pair streams
    __shared__ __attribute__((aligned(256))) char smem[4 * STAGE];

    const int lane = threadIdx.x & 63, hh = lane >> 5, lk = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kb = wave & 1, hp = wave >> 1;
#ifdef DKV_STAMPS
    unsigned long long st_entry, st_loop = 0, st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, sum_a = 0, sum_b = 0, sum_c = 0, sum_d = 0, n_steps = 0;
    DKV_STAMP(st_entry);
#endif
    // GQA group of G query heads per KV head, walked as two head streams (hp = 0, 1) of NH = ceil(G / 2) heads each.  Odd G: the last
    // head slot of stream 1 is a dummy (it re-reads the group's last head with lse = +inf, i.e. p = 0 and dS = 0: it adds exact zeros,
    // and keeps the two streams' barrier counts equal).  G = 4: Llama-3 / Mistral; G = 7: Qwen2-7B.
    const int G = H / Hkv, NH = (G + 1) >> 1;
    int bx, hk, b;
    xcd_tile_map((L + 63) >> 6, Hkv, bx, hk, b, CAUSAL);
    const int kblk0 = bx * 64, k0 = kblk0 + kb * 32, key = k0 + lk;
    const int keyc = key < L ? key : L - 1;
    const float c = scale * LOG2E;
    const bool key_ok = key < L && (kmask == nullptr || kmask[(long)b * L + keyc] != 0);

    bf16x8 kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int ch = ks * 2 + hh;
        kf[ks] = as_bf16x8(*reinterpret_cast<const u32x4*>(K + ((long)b * L + keyc) * ldk + (long)hk * HD + ch * 8));
        vf[ks] = as_bf16x8(*reinterpret_cast<const u32x4*>(V + ((long)b * L + keyc) * ldv + (long)hk * HD + ch * 8));
    }
    f32x16 dkacc[NDB], dvacc[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dkacc[d][e] = 0.f; dvacc[d][e] = 0.f; }

    // query tiles that can see this key block: from the block's first key (causal) to the end of the sequence / of the keys' sample
    const int qstart = CAUSAL ? (kblk0 >> 5) : 0;
    int qlim = L;
    if constexpr (SEG) {
        const int kk = kblk0 + lane;
        int e = kk < L ? qend[(long)b * L + kk] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(e, o, 64); e = e > y ? e : y; }
        e = __builtin_amdgcn_readfirstlane(e);
        qlim = e < L ? e : L;
    }
    const int ntl = ((qlim + QT - 1) >> 5) - qstart;       // tiles per head (may be <= 0)
    // Tiles that need per-element masking are walked FIRST by a plain loop; the rest -- the bulk: interior tiles of fully valid key
    // blocks -- by the software-pipelined loop, whose body is one branch-free basic block.  Causal: the first two tiles of a head
    // touch the diagonal of this 64-key block.  Irregular blocks (padded keys; with segments: keys of more than one sample) mask
    // every tile.  The order only permutes a sum that is the same for every launch: deterministic.
    // (decided on all 64 keys of the block by every wave alike: the four waves must agree on the tile split)
    bool irregular;
    {
        const int kk = kblk0 + lane, kkc = kk < L ? kk : L - 1;
        irregular = !__all(kk < L && (kmask == nullptr || kmask[(long)b * L + kkc] != 0));
    }
    if constexpr (SEG) {
        const int kk = kblk0 + lane, kkc = kk < L ? kk : L - 1;
        const int e = qend[(long)b * L + kkc], e0 = __builtin_amdgcn_readfirstlane(e);
        irregular = irregular || !__all(e == e0 || kk >= L);
    }
    irregular = __builtin_amdgcn_readfirstlane((int)irregular) != 0;
    const int ntc = ntl > 0 ? ntl : 0;
    const int nm = irregular ? ntc : (CAUSAL ? (ntc < 2 ? ntc : 2) : 0);     // masked tiles per head
    const int nu1 = ntc - nm;                                                // unmasked tiles per head

    // stage = (a) request the tile's lse / rowsum(dO*O) / kstart words into registers FIRST (oldest entries of the vmcnt queue:
    // their wait does not drain the DMA behind them), (b) issue the LDS-DMA of the Q and dO rows; (c) `stage_finish`, called at the
    // end of the iteration, puts the words into LDS.  Everything goes through buffer descriptors: the per-lane offsets are fixed for
    // the whole kernel, a tile adds one wave-uniform offset to them (no 64-bit address arithmetic in the loop), and rows beyond L are
    // out of range of the descriptor -> the DMA writes zeros (no clamping, no branches: the loop body stays one basic block).
    // A tile is (gi = which head of the pair, qt = query tile index).
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(Q + (long)b * L * ldq), 0, (int)(unsigned)((long)L * ldq * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)(dO + (long)b * L * ldo), 0, (int)(unsigned)((long)L * ldo * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void*)(LSE + (long)b * H * L), 0, (int)(unsigned)((long)H * L * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(Dsum + (long)b * H * L), 0, (int)(unsigned)((long)H * L * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(SEG ? kstart + (long)b * L : nullptr), 0, SEG ? (int)(unsigned)((long)L * 4) : 0, 0x00020000);
    unsigned voQ[4], voD[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (kb * 4 + j) * 4 + (lane >> 4);
        const int cc = (lane & 15) ^ lds_swz(row);
        voQ[j] = (unsigned)(((long)row * ldq + cc * 8) * 2);
        voD[j] = (unsigned)(((long)row * ldo + cc * 8) * 2);
    }
    auto stage = [&](int gi, int qt, int slot, float& vl, float& vs, int& ksv) {
        const int q0 = qt * QT;
        const int hl = hp * NH + gi;                                  // head slot inside the group (hl >= G: the dummy slot)
        const int h = hk * G + (hl < G ? hl : G - 1);
        char* base = smem + slot * STAGE + hp * STREAM;
        // NB the hardware range check of a raw buffer covers the VECTOR offset only (the scalar offset is added after it), so the
        // tile offset is added into the vector offset: one v_add per access buys "rows beyond L read zeros" for the last tile
        const unsigned soL = (unsigned)(((long)h * L + q0) * 4);
        // a query row beyond L reads lse / dsum words of the next head (or zeros past the end): replaced in stage_finish
        vl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, (unsigned)lk * 4u + soL, 0, 0));
        vs = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsS, (unsigned)lk * 4u + soL, 0, 0));
        if constexpr (SEG) ksv = (int)__builtin_amdgcn_raw_buffer_load_b32(rsK, (unsigned)lk * 4u + (unsigned)q0 * 4u, 0, 0);
        const unsigned soQ = (unsigned)(((long)q0 * ldq + (long)h * HD) * 2), soD = (unsigned)(((long)q0 * ldo + (long)h * HD) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = kb * 4 + j;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_void_t*)(base + piece * 1024), 16, voQ[j] + soQ, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_t*)(base + TILE + piece * 1024), 16, voD[j] + soD, 0, 0, 0);
        }
    };
    // the same in two parts for the merged step (ATTN_DKV_MERGED): the words at the top of the step, the DMA pieces one per MFMA slot
    auto stage_words = [&](int gi, int qt, float& vl, float& vs, int& ksv) {
        const int q0 = qt * QT;
        const int hl = hp * NH + gi;
        const int h = hk * G + (hl < G ? hl : G - 1);
        const unsigned soL = (unsigned)(((long)h * L + q0) * 4);
        vl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, (unsigned)lk * 4u + soL, 0, 0));
        vs = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsS, (unsigned)lk * 4u + soL, 0, 0));
        if constexpr (SEG) ksv = (int)__builtin_amdgcn_raw_buffer_load_b32(rsK, (unsigned)lk * 4u + (unsigned)q0 * 4u, 0, 0);
    };
    auto stage_piece = [&](auto qc, int gi, int qt, int slot) {       // piece q = 2 j + (0: Q rows, 1: dO rows) of the tile
        constexpr int q = decltype(qc)::value, j = q >> 1;
        const int q0 = qt * QT;
        const int hl = hp * NH + gi;
        const int h = hk * G + (hl < G ? hl : G - 1);
        char* base = smem + slot * STAGE + hp * STREAM;
        const int piece = kb * 4 + j;
        if constexpr ((q & 1) == 0) {
            const unsigned soQ = (unsigned)(((long)q0 * ldq + (long)h * HD) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_void_t*)(base + piece * 1024), 16, voQ[j] + soQ, 0, 0, 0);
        } else {
            const unsigned soD = (unsigned)(((long)q0 * ldo + (long)h * HD) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_t*)(base + TILE + piece * 1024), 16, voD[j] + soD, 0, 0, 0);
        }
    };
    auto stage_finish = [&](int gi, int qt, int slot, float vl, float vs, int ksv) {
        const int qq = (hp * NH + gi < G) ? qt * QT + lk : 0x7fffffff;      // dummy head slot: every row is "beyond the last query"
        float* fp = reinterpret_cast<float*>(smem + slot * STAGE + hp * STREAM + 2 * TILE);
        const float val = hh == 0 ? vl * LOG2E : vs;
        // +inf beyond the last query that can see this key block -> p = 0: beyond L, and (packed samples) beyond the end of the
        // keys' sample -- for a regular block every key shares that end, so the tile straddling it needs no per-element mask
        fp[lane] = qq < qlim ? val : (hh == 0 ? INFINITY : 0.f);
        if constexpr (SEG) reinterpret_cast<int*>(fp)[2 * QT + lk] = ksv;
    };
    // S, dP of the tile in `slot` -> s, dp   (the first MFMA of each chain takes the constant 0 as its accumulator input)
    auto scores = [&](int slot, f32x16& s, f32x16& dp) {
        const char* sQ = smem + slot * STAGE + hp * STREAM;
        const char* sdO = sQ + TILE;
        f32x16 z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int off = Y::chunk_off(lk, ks * 2 + hh);
            const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sQ + off);
            const bf16x8 da = *reinterpret_cast<const bf16x8*>(sdO + off);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], ks == 0 ? z : s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], ks == 0 ? z : dp, 0, 0, 0);
        }
    };
    // P and dS of a tile from its S, dP (in place: s <- P, dp <- dS)
    auto softmax_bwd = [&](auto masked, int q0, int slot, f32x16& s, f32x16& dp) {
        const float* sLse = reinterpret_cast<const float*>(smem + slot * STAGE + hp * STREAM + 2 * TILE);
        const float* sDs = sLse + QT;
        const int* sKs = reinterpret_cast<const int*>(sLse + 2 * QT);
        if constexpr (decltype(masked)::value) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = (r & 3) + 8 * (r >> 2) + 4 * hh;
                float v = key_ok ? s[r] * c : -INFINITY;
                if (CAUSAL && key > q0 + ql) v = -INFINITY;
                if constexpr (SEG) { if (key < sKs[ql]) v = -INFINITY; }
                const float p = __builtin_amdgcn_exp2f(v - sLse[ql]);
                s[r] = p;
                dp[r] = p * (dp[r] - sDs[ql]);       // the softmax scale is applied once, to the dK accumulators
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -sLse[ql]));
                s[r] = p;
                dp[r] = p * (dp[r] - sDs[ql]);
            }
        }
    };
    // dV^T += dO^T . P, dK^T += Q^T . dS  (transposing reads of the same row-major tiles)
    auto accumulate = [&](int slot, const f32x16& p, const f32x16& ds) {
        const char* sQ = smem + slot * STAGE + hp * STREAM;
        const char* sdO = sQ + TILE;
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const bf16x8 pf = pack_frag(p, cp);
            const bf16x8 dsf = pack_frag(ds, cp);
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
                const bf16x8 dot = read_tr_frag<HD>(sdO, 16 * cp, d * 32, lane);
                const bf16x8 qtf = read_tr_frag<HD>(sQ, 16 * cp, d * 32, lane);
                dvacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pf, dvacc[d], 0, 0, 0);
                dkacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dkacc[d], 0, 0, 0);
            }
        }
    };

    // ---- masked tiles: plain double-buffered walk (slots 0 / 1), tile j -> (gi = j / nm, qt = qstart + j % nm)
    {
        const int ntot = NH * nm;
        if (ntot > 0) {
            int gi = 0, qt = qstart;           // tile j = (head slot j / nm, query tile qstart + j % nm), carried instead of divided
            float vl, vs;
            int ksv = 0;
            stage(gi, qt, 0, vl, vs, ksv);
            stage_finish(gi, qt, 0, vl, vs, ksv);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int j = 0; j < ntot; ++j) {
                const int slot = j & 1;
                int gn = gi, qn = qt;
                if (j + 1 < ntot) {
                    const bool wrap = (qn + 1 == qstart + nm);
                    qn = wrap ? qstart : qn + 1;
                    gn += wrap ? 1 : 0;
                }
                stage(gn, qn, slot ^ 1, vl, vs, ksv);
                stage_finish(gn, qn, slot ^ 1, vl, vs, ksv);
                f32x16 s, dp;
                scores(slot, s, dp);
                softmax_bwd(std::true_type{}, qt * QT, slot, s, dp);
                accumulate(slot, s, dp);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                gi = gn;
                qt = qn;
            }
        }
    }
    // ---- unmasked tiles: 4-slot ring, three-stage software pipeline; tile j -> (gi = j / nu1, qt = qstart + nm + j % nu1)
    // Step j runs, on ONE wave per SIMD:   MS(j+2): S, dP of tile j+2 (16 MFMAs, slots 0-15)
    //                                       SM(j+1): softmax backward of tile j+1 on the VALU, one element per two slots
    //                                       MA(j):   dV, dK accumulation of tile j (16 MFMAs, slots 16-31)
    // so every one of the 32 MFMA slots of a step shadows the same small bundle: its LDS fragment reads (requested PF slots ahead),
    // half a softmax element, a pack.  Register roles rotate with period 3, ring slots with period 4, the packed-fragment buffers
    // with period 2 -> the loop body is written out for 12 steps (all indices compile-time).  The score MFMAs are issued through
    // inline asm with VGPR destinations and the K / V fragments as AGPR operands: S and dP then need no v_accvgpr_read per
    // element (the accumulators proper -- dK, dV -- stay in AGPRs under the compiler's builtin).
    {
        const int n = NH * nu1;
        // tile j = (head slot j / nu1, query tile qstart + nm + j % nu1); the staging cursor (pg, pq) walks the tiles one by one and
        // stays on the last tile once it is reached (re-staging it is harmless), so the loop carries no division
        int pg = 0, pq = qstart + nm, pj = 0;
        auto cursor_next = [&]() {
            if (pj + 1 < n) {
                const bool wrap = (pq + 1 == qstart + nm + nu1);
                pq = wrap ? qstart + nm : pq + 1;
                pg += wrap ? 1 : 0;
                ++pj;
            }
        };
        if (n > 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) { asm volatile("" : "+a"(kf[ks])); asm volatile("" : "+a"(vf[ks])); }
            {
                float v[3], w[3];
                int zz[3] = {0, 0, 0}, g[3], q[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) { g[t] = pg; q[t] = pq; stage(g[t], q[t], t, v[t], w[t], zz[t]); cursor_next(); }
#pragma unroll
                for (int t = 0; t < 3; ++t) stage_finish(g[t], q[t], t, v[t], w[t], zz[t]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            f32x16 sv[3], dv[3];          // S / dP -> P / dS of three tiles in flight (roles rotate)
            u32x4 pk[2][2][2];            // [parity][0: P, 1: dS][cp] packed bf16 fragments
            auto rowfrag = [&](const char* tq, int g) -> bf16x8 {          // g = 2 * ks + which (0: Q for S, 1: dO for dP)
                return *reinterpret_cast<const bf16x8*>(tq + (g & 1) * TILE + Y::chunk_off(lk, (g >> 1) * 2 + hh));
            };
            auto mfma_s = [&](auto first, f32x16& acc, const bf16x8& a, const bf16x8& bagpr) {
                if constexpr (decltype(first)::value)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(bagpr));
                else
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(bagpr));
            };
            // softmax backward of ONE element r (in place) with the lse / dsum words of its tile
            auto sm_elem = [&](auto rc, f32x16& sx, f32x16& dx, const f32x4 (&lse4)[4], const f32x4 (&dsm4)[4]) {
                constexpr int r = decltype(rc)::value;
                if constexpr (kDkvAsmElems) {
                    // the element's four instructions as ONE statement: placed where the slot puts them (a volatile asm neither sinks to its
                    // user's block nor moves across the slot's MFMA), and without the two empty pinning statements of the C++ form, each of
                    // which the hazard recognizer pads with an s_nop (33 - 44 s_nop per 32-MFMA step, profiles/r05_attn_anatomy.md).  Same
                    // arithmetic: p = exp2(s * c - lse), dS = p * (dP - D); the v_sub between v_exp and its use covers the trans-use hazard
                    float p, ds;
                    asm volatile("v_fma_f32 %0, %2, %3, -%4\n\tv_exp_f32 %0, %0\n\tv_sub_f32 %1, %5, %6\n\tv_mul_f32 %1, %1, %0"
                                 : "=&v"(p), "=&v"(ds)
                                 : "v"(sx[r]), "v"(c), "v"(lse4[r >> 2][r & 3]), "v"(dx[r]), "v"(dsm4[r >> 2][r & 3]));
                    sx[r] = p;
                    dx[r] = ds;
                } else {
                    float l = lse4[r >> 2][r & 3];
                    asm volatile("" : "+v"(l));            // pins the element's arithmetic behind the slot's (volatile asm) MFMA
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[r], c, -l));
                    float ds = p * (dx[r] - dsm4[r >> 2][r & 3]);
                    asm volatile("" : "+v"(p), "+v"(ds));  // ... and in front of the next one (IR sinking would move it to its user's block)
                    sx[r] = p;
                    dx[r] = ds;
                }
            };
            auto pack_pinned = [&](float a, float b2) -> unsigned {
                if constexpr (kDkvAsmElems) {
                    unsigned u;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(b2));      // == pack_bf2 (round to nearest even), pinned
                    return u;
                } else {
                    asm volatile("" : "+v"(a));
                    unsigned u = pack_bf2(a, b2);
                    asm volatile("" : "+v"(u));
                    return u;
                }
            };
            auto load_words = [&](int slot, f32x4 (&lse4)[4], f32x4 (&dsm4)[4]) {
                const float* sLse = reinterpret_cast<const float*>(smem + slot * STAGE + hp * STREAM + 2 * TILE) + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    lse4[g] = *reinterpret_cast<const f32x4*>(sLse + 8 * g);
                    dsm4[g] = *reinterpret_cast<const f32x4*>(sLse + QT + 8 * g);
                }
            };
            // ---- prologue: S, dP of tiles 0 and 1; softmax backward + packs of tile 0
            {
                const char* t0 = smem + 0 * STAGE + hp * STREAM;
                const char* t1 = smem + 1 * STAGE + hp * STREAM;
                static_for<0, 16>([&](auto gc) {
                    constexpr int g = decltype(gc)::value, ks = g >> 1;
                    if constexpr (g & 1) mfma_s(std::integral_constant<bool, ks == 0>{}, dv[0], rowfrag(t0, g), vf[ks]);
                    else mfma_s(std::integral_constant<bool, ks == 0>{}, sv[0], rowfrag(t0, g), kf[ks]);
                });
                static_for<0, 16>([&](auto gc) {
                    constexpr int g = decltype(gc)::value, ks = g >> 1;
                    if constexpr (g & 1) mfma_s(std::integral_constant<bool, ks == 0>{}, dv[1], rowfrag(t1, g), vf[ks]);
                    else mfma_s(std::integral_constant<bool, ks == 0>{}, sv[1], rowfrag(t1, g), kf[ks]);
                });
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // MFMA results -> VALU readers (the asm MFMAs are invisible to hipcc)
                f32x4 lse4[4], dsm4[4];
                load_words(0, lse4, dsm4);
                static_for<0, 16>([&](auto rc) { sm_elem(rc, sv[0], dv[0], lse4, dsm4); });
#pragma unroll
                for (int cp = 0; cp < 2; ++cp)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pk[0][0][cp][e] = pack_bf2(sv[0][8 * cp + 2 * e], sv[0][8 * cp + 2 * e + 1]);
                        pk[0][1][cp][e] = pack_bf2(dv[0][8 * cp + 2 * e], dv[0][8 * cp + 2 * e + 1]);
                    }
            }
            // this lane's transposing-read addresses (tr_issue): [ring slot pair][d-block], first read / the read eight rows further down; ring
            // slot parity, the dO tile and the second 16-row half are immediate offsets
            unsigned trA[2][4], trB[2][4];
            if constexpr (kDkvTrAsm) {
                const int s16 = lane & 15, g16 = (lane >> 4) & 1;
                const int row = 4 * hh + (s16 >> 2);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int col = d * 32 + 16 * g16 + (s16 & 3) * 4;
                    const int off = Y::chunk_off(row, col >> 3) + (col & 7) * 2;
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        trA[pr][d] = lds_addr_of(smem + pr * 2 * STAGE + hp * STREAM + off);
                        trB[pr][d] = lds_addr_of(smem + pr * 2 * STAGE + hp * STREAM + ((off + 8 * 256) ^ 32));
                    }
                }
            }
            auto step = [&](int j, auto kc) {
                constexpr int k = decltype(kc)::value;                 // position inside the 12-step trip
                constexpr int slot0 = k % 4, slot1 = (k + 1) % 4, slot2 = (k + 2) % 4, slot3 = (k + 3) % 4;
                constexpr int r0 = k % 3, r1 = (k + 1) % 3, r2 = (k + 2) % 3;     // roles of tiles j, j+1, j+2
                constexpr int par0 = k % 2, par1 = (k + 1) % 2;
                constexpr int PF = 4;
                const char* tq0 = smem + slot0 * STAGE + hp * STREAM;    // tile j: transposed fragments
                const char* tq2 = smem + slot2 * STAGE + hp * STREAM;    // tile j + 2: row fragments
                DKV_STAMP(st0);
                const int gi3 = pg, qt3 = pq;                 // tile j + 3 (or the last tile again)
                cursor_next();
                float vl, vs;
                int ksv = 0;
                // slot3 held tile j-1: released by the barrier that ended step j-1.  Merged step: only the lse / rowsum words are requested
                // here; the eight DMA pieces ride in the MFMA slots 1, 3, .. 15 (8 x ~60 cycles of issue with nothing else running cost the
                // grouped step 655 of its 3087 cycles, profiles/r05_attn_anatomy.md) -- early enough to land before the step's barrier
                if constexpr (kDkvMerged && kDkvDmaInSlots) stage_words(gi3, qt3, vl, vs, ksv);
                else stage(gi3, qt3, slot3, vl, vs, ksv);
                f32x4 lse4[4], dsm4[4];
                load_words(slot1, lse4, dsm4);
                bf16x8 fr[PF + 1];
#pragma unroll
                for (int g = 0; g < PF; ++g) fr[g] = rowfrag(tq2, g);
                DKV_STAMP(st1);
                FragU tf[4];
                // transposed fragment g = cp * 8 + d * 2 + which (0: dO^T for dV, 1: Q^T for dK) of tile j.  Hand-issued (kDkvTrAsm): the builtin
                // form put s_waitcnt vmcnt(0) in front of the step's first transposing read, i.e. the DMA of tile j + 3, issued at the top of
                // this very step, was waited for in the MIDDLE of the step instead of at its end.  The reads' completion is then counted by
                // hand: LDS operations retire in order, and behind the reads of fragment g only those of g + 1, g + 2 have been issued
                auto tr_issue = [&](auto gc2, FragU& dst) {
                    constexpr int g = decltype(gc2)::value, d = (g >> 1) & 3;
                    if constexpr (kDkvTrAsm) {
                        constexpr int imm = (slot0 & 1) * STAGE + ((g & 1) ? 0 : TILE) + (g >> 3) * 4096;
                        const unsigned a0 = trA[slot0 >> 1][d], a1 = trB[slot0 >> 1][d];      // (named: asm operands alone do not capture)
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst.h[0]) : "v"(a0), "i"(imm));
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst.h[1]) : "v"(a1), "i"(imm));
                    } else {
                        dst.f = read_tr_frag<HD>(tq0 + ((g & 1) ? 0 : TILE), 16 * (g >> 3), d * 32, lane);
                    }
                };
                auto tr_wait = [&](auto gc2, FragU& f) {
                    constexpr int g = decltype(gc2)::value;
                    if constexpr (kDkvTrAsm) {
                        if constexpr (g <= 13) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f.f));
                        else if constexpr (g == 14) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f.f));
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.f));
                    }
                };
                if constexpr (kDkvMerged) {
                    // MERGED order (round 5): the step's 32 MFMAs alternate between the two groups -- S(ks), dV, dP(ks), dK, S(ks + 1), ... -- so
                    // that two MFMAs on the SAME accumulator are four instructions apart.  In the grouped order below the score chains
                    // S(0) dP(0) S(1) dP(1) ... put every other MFMA behind its own predecessor's result (a 32x32x16 MFMA takes 16 passes;
                    // with VALU / LDS work issued in between, a dependent pair costs ~110 cycles instead of 64): the in-kernel stamps showed
                    // 1359 cycles for those 16 MFMAs against 775 for the 16 independent accumulation MFMAs (profiles/r05_attn_anatomy.md).
                    // Even slot m: item g = m / 2 of MS(j+2); odd slot: item g = (m - 1) / 2 of MA(j).  The transposed fragments of items 0-2 are
                    // requested at the top of the step (hand-issued: a builtin read would wait for the DMA issued just above), the cp = 0 packs
                    // of tile j + 1 move behind the softmax-backward elements they read (odd items 3, 7, 11, 15).
                    tr_issue(std::integral_constant<int, 0>{}, tf[0]);
                    tr_issue(std::integral_constant<int, 1>{}, tf[1]);
                    tr_issue(std::integral_constant<int, 2>{}, tf[2]);
                    static_for<0, 32>([&](auto mc) {
                        constexpr int m = decltype(mc)::value, g = m >> 1;
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr ((m & 1) == 0) {
                            constexpr int ks = g >> 1;
                            if constexpr (g & 1) mfma_s(std::integral_constant<bool, ks == 0>{}, dv[r2], fr[g % (PF + 1)], vf[ks]);
                            else mfma_s(std::integral_constant<bool, ks == 0>{}, sv[r2], fr[g % (PF + 1)], kf[ks]);
                            if constexpr (g + PF < 16) fr[(g + PF) % (PF + 1)] = rowfrag(tq2, g + PF);
                            if constexpr (g < 8) {
                                constexpr int e = g & 3;
                                if constexpr (g < 4) pk[par0][0][1][e] = pack_pinned(sv[r0][8 + 2 * e], sv[r0][8 + 2 * e + 1]);
                                else pk[par0][1][1][e] = pack_pinned(dv[r0][8 + 2 * e], dv[r0][8 + 2 * e + 1]);
                            }
                            if constexpr (g & 1) sm_elem(std::integral_constant<int, (g >> 1)>{}, sv[r1], dv[r1], lse4, dsm4);
                        } else {
                            constexpr int cp = g >> 3, d = (g >> 1) & 3;
                            asm volatile("" ::: "memory");
                            tr_wait(std::integral_constant<int, g>{}, tf[g & 3]);
                            if constexpr (g & 1) dkacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[g & 3].f, as_bf16x8(pk[par0][1][cp]), dkacc[d], 0, 0, 0);
                            else dvacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[g & 3].f, as_bf16x8(pk[par0][0][cp]), dvacc[d], 0, 0, 0);
                            if constexpr (g + 3 < 16) tr_issue(std::integral_constant<int, g + 3>{}, tf[(g + 3) & 3]);
                            if constexpr (kDkvDmaInSlots && g < 8) stage_piece(std::integral_constant<int, g>{}, gi3, qt3, slot3);
                            if constexpr (g & 1) sm_elem(std::integral_constant<int, 8 + (g >> 1)>{}, sv[r1], dv[r1], lse4, dsm4);
                            if constexpr ((g & 3) == 3) {       // elements 2 e, 2 e + 1 of tile j + 1 left the softmax backward in even item 4 e + 3
                                constexpr int e = g >> 2;
                                pk[par1][0][0][e] = pack_pinned(sv[r1][2 * e], sv[r1][2 * e + 1]);
                                pk[par1][1][0][e] = pack_pinned(dv[r1][2 * e], dv[r1][2 * e + 1]);
                            }
                        }
                    });
                    DKV_STAMP(st2);
                } else {
                    // slots 0-15: MS(j+2)  ||  SM(j+1) elements 0-7  ||  cp = 1 packs of tile j (its elements 8-15 finished last step)
                    static_for<0, 16>([&](auto gc) {
                        constexpr int g = decltype(gc)::value, ks = g >> 1;
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (g & 1) mfma_s(std::integral_constant<bool, ks == 0>{}, dv[r2], fr[g % (PF + 1)], vf[ks]);
                        else mfma_s(std::integral_constant<bool, ks == 0>{}, sv[r2], fr[g % (PF + 1)], kf[ks]);
                        if constexpr (g + PF < 16) fr[(g + PF) % (PF + 1)] = rowfrag(tq2, g + PF);
                        if constexpr (g < 8) {
                            constexpr int e = g & 3;
                            if constexpr (g < 4) pk[par0][0][1][e] = pack_pinned(sv[r0][8 + 2 * e], sv[r0][8 + 2 * e + 1]);
                            else pk[par0][1][1][e] = pack_pinned(dv[r0][8 + 2 * e], dv[r0][8 + 2 * e + 1]);
                        }
                        if constexpr (g & 1) sm_elem(std::integral_constant<int, (g >> 1)>{}, sv[r1], dv[r1], lse4, dsm4);
                        if constexpr (g >= 13) tr_issue(std::integral_constant<int, g - 13>{}, tf[g - 13]);
                    });
                    DKV_STAMP(st2);
                    // slots 16-31: MA(j)  ||  SM(j+1) elements 8-15  ||  cp = 0 packs of tile j+1
                    static_for<0, 16>([&](auto gc) {
                        constexpr int g = decltype(gc)::value, cp = g >> 3, d = (g >> 1) & 3;
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("" ::: "memory");
                        tr_wait(gc, tf[g & 3]);
                        if constexpr (g & 1) dkacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[g & 3].f, as_bf16x8(pk[par0][1][cp]), dkacc[d], 0, 0, 0);
                        else dvacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[g & 3].f, as_bf16x8(pk[par0][0][cp]), dvacc[d], 0, 0, 0);
                        if constexpr (g + 3 < 16) tr_issue(std::integral_constant<int, g + 3>{}, tf[(g + 3) & 3]);
                        if constexpr (g & 1) sm_elem(std::integral_constant<int, 8 + (g >> 1)>{}, sv[r1], dv[r1], lse4, dsm4);
                        if constexpr (g < 8) {
                            constexpr int e = g & 3;
                            if constexpr (g < 4) pk[par1][0][0][e] = pack_pinned(sv[r1][2 * e], sv[r1][2 * e + 1]);
                            else pk[par1][1][0][e] = pack_pinned(dv[r1][2 * e], dv[r1][2 * e + 1]);
                        }
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
                DKV_STAMP(st3);
                stage_finish(gi3, qt3, slot3, vl, vs, ksv);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile j+3 landed (this wave's pieces)
                __syncthreads();
#ifdef DKV_STAMPS
                DKV_STAMP(st4);
                if (st_loop == 0) st_loop = st0;
                sum_a += st1 - st0; sum_b += st2 - st1; sum_c += st3 - st2; sum_d += st4 - st3; n_steps += 1;
#endif
            };
            int j = 0;
            while (true) {
#define DKV_STEP(K_) step(j, std::integral_constant<int, K_>{}); if (++j >= n) break;
                DKV_STEP(0) DKV_STEP(1) DKV_STEP(2) DKV_STEP(3) DKV_STEP(4) DKV_STEP(5)
                DKV_STEP(6) DKV_STEP(7) DKV_STEP(8) DKV_STEP(9) DKV_STEP(10) DKV_STEP(11)
#undef DKV_STEP
            }
            // (the asm MFMAs of the final steps wrote registers nothing reads; they retired long ago: 16 accumulation MFMAs and a
            // barrier follow them inside the step, so no keep-alive is needed -- and one would force the rotating roles to a common
            // register assignment at all twelve loop exits)
        }
    }
    // cross-stream reduction: [kb][which][d][r][lane] fp32; hp = 0 hands over its dV partial and finishes dK, hp = 1 the converse
    __syncthreads();
    {
        float* red = reinterpret_cast<float*>(smem);
        float* mine = red + ((kb * 2 + hp) * 64) * 64;          // region written by this wave
        const float* theirs = red + ((kb * 2 + (hp ^ 1)) * 64) * 64;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) mine[(d * 16 + e) * 64 + lane] = hp == 0 ? dvacc[d][e] : dkacc[d][e];
        __syncthreads();
        if (key < L) {
            bf16_t* op = (hp == 0 ? dK + ((long)b * L + key) * lddk : dV + ((long)b * L + key) * lddv) + (long)hk * HD;
            const float sc = hp == 0 ? scale : 1.f;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float own = hp == 0 ? dkacc[d][4 * g4 + e] : dvacc[d][4 * g4 + e];
                        v[e] = (own + theirs[(d * 16 + 4 * g4 + e) * 64 + lane]) * sc;
                    }
                    u32x2 o;
                    o[0] = pack_bf2(v[0], v[1]);
                    o[1] = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(op + d * 32 + 8 * g4 + 4 * hh) = o;
                }
        }
    }
#ifdef DKV_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        unsigned long long st_exit;
        DKV_STAMP(st_exit);
        unsigned long long* o = g_dkv_stamps + (size_t)blockIdx.x * 8;
        o[0] = st_loop ? st_loop - st_entry : 0; o[1] = st_exit - st_entry; o[2] = n_steps; o[3] = sum_a; o[4] = sum_b; o[5] = sum_c; o[6] = sum_d;
        o[7] = (unsigned long long)bx;
    }
#endif
}

#ifdef DKV_STAMPS
// probe builds only: copy the stamps of the last attn_bwd_dkv_g4_kernel launch (n workgroups x 8 u64) to host memory
extern "C" int mantis_probe_dkv_stamps(void* host_dst, int n_wg) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_dkv_stamps), (size_t)n_wg * 64, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif

static int attn_num_cus() {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

// group sizes served by attn_bwd_dkv_g4_kernel: every even G, odd G from 5 on (one dummy head slot in G + 1: at most 1/6 wasted)
static inline bool attn_dkv_group_kernel_ok(int G) { return G >= 2 && (G % 2 == 0 || G >= 5); }

// attn_fwd64.hip: the hd-128 forward with 64 query rows per wave
int mantis_attn_fwd64_launch(bool causal, int B, hipStream_t s, const bf16_t* Q, const bf16_t* K, const bf16_t* V, const int* kmask,
                             bf16_t* O, float* LSE, int L, int Lk, int H, int Hkv, long ldq, long ldk, long ldv, long ldo, float scale,
                             const int* kstart);

// The hd-128 forward runs on attn_fwd64_kernel (64 query rows per wave) from 1024 query rows on; MANTIS_ATTN_FWD64 = 0 (read once) keeps
// attn_fwd_kernel<128> for every length, = 1 forces attn_fwd64 for every length.  Same-box A/B, round 3, Llama-3 step geometry (B 2,
// L 2812, 32/8 heads): 197 vs 210 us without a key mask, 188 vs 215 us with one; Qwen2-7B geometry (L 4096, 28/4): 139 vs 176 us.
static int attn_fwd64_mode() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MANTIS_ATTN_FWD64");
        v = (e && e[0] == '0') ? 0 : (e && e[0] == '1') ? 1 : 2;
    }
    return v;
}

template <int HD>
static int launch_fwd(bool causal, dim3 grid, hipStream_t s, const bf16_t* Q, const bf16_t* K, const bf16_t* V, const int* kmask,
                      bf16_t* O, float* LSE, int L, int Lk, int H, int Hkv, long ldq, long ldk, long ldv, long ldo, float scale,
                      const int* kstart) {
    if constexpr (HD == 128) {
        const int mode = attn_fwd64_mode();
        if ((mode == 1 || (mode == 2 && L >= 1024)) && Lk <= 65536)      // the kernel's per-tile liveness table holds 1024 tiles
            return mantis_attn_fwd64_launch(causal, (int)(grid.x / (cdiv(L, 128) * H)), s, Q, K, V, kmask, O, LSE, L, Lk, H, Hkv, ldq, ldk,
                                            ldv, ldo, scale, kstart);
    }
    if (causal)
        MANTIS_LAUNCH((attn_fwd_kernel<HD, true>), grid, dim3(256), 0, s, Q, K, V, kmask, O, LSE, L, Lk, H, Hkv, ldq, ldk, ldv,
                           ldo, scale, kstart);
    else
        MANTIS_LAUNCH((attn_fwd_kernel<HD, false>), grid, dim3(256), 0, s, Q, K, V, kmask, O, LSE, L, Lk, H, Hkv, ldq, ldk, ldv,
                           ldo, scale, kstart);
    return mantis_check_launch();
}

// (round 6: the opt-in 64-query-rows-per-wave dQ kernel of rounds 4 - 5, csrc/attn_dq64.hip, is gone: correct, but 266 vs 233 us against this
// file's attn_bwd_dq_kernel in every measurement of two rounds -- a 512-register kernel runs one workgroup per CU and nothing overlaps its
// prologue, profiles/r05_attn_anatomy.md -- and a kernel off the default path earns nothing.)
template <int HD>
static void launch_dq(bool causal, hipStream_t s, const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO, const int* kmask,
                      const float* LSE, float* Dsum, bf16_t* dQ, int B, int L, int Lk, int H, int Hkv, long ldq, long ldk, long ldv,
                      long ldo, long lddq, float scale, const bf16_t* Ofwd, long ldout, const int* kstart) {
    const dim3 gq(cdiv(L, 128) * H * B);
    if (causal)
        MANTIS_LAUNCH((attn_bwd_dq_kernel<HD, true>), gq, dim3(256), 0, s, Q, K, V, dO, kmask, LSE, Dsum, dQ, L, Lk, H, Hkv, ldq, ldk, ldv,
                           ldo, lddq, scale, Ofwd, ldout, kstart);
    else
        MANTIS_LAUNCH((attn_bwd_dq_kernel<HD, false>), gq, dim3(256), 0, s, Q, K, V, dO, kmask, LSE, Dsum, dQ, L, Lk, H, Hkv, ldq, ldk, ldv,
                           ldo, lddq, scale, Ofwd, ldout, kstart);
}

template <int HD>
static int launch_bwd(bool causal, hipStream_t s, const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO,
                      const int* kmask, const float* LSE, float* Dsum, bf16_t* dQ, bf16_t* dK, bf16_t* dV, bf16_t* ws, int B,
                      int L, int Lk, int H, int Hkv, long ldq, long ldk, long ldv, long ldo, long lddq, long lddk, long lddv, float scale,
                      const bf16_t* Ofwd, long ldout, const int* kstart, const int* qend) {
    const dim3 gk(cdiv(Lk, DkvCfg<HD>::KEYS) * H * B);       // Lk keys per batch entry
    const int G = H / Hkv;
    if constexpr (HD == 128) {
        // GQA-aware dK/dV (no HBM partials, no group reduce); dQ as before (it also publishes Dsum).  Its grid is one workgroup per
        // (64-key block, KV head) with causal work that varies 64x across key blocks: below about two workgroups per CU the heaviest
        // block sets the time (Qwen2-7B at B = 1, L = 4096: 256 workgroups, 553 us vs 490 us for the per-query-head path, measured), so
        // other group sizes than Llama's take it only when the grid is either tiny or at least two rounds deep
        const long nwg = (long)cdiv(L, 64) * Hkv * B;
        const int cus = attn_num_cus();
        if (Lk == L && attn_dkv_group_kernel_ok(G) && (G == 4 || nwg >= 2L * cus || nwg <= cus / 2 || ws == nullptr)) {
            launch_dq<HD>(causal, s, Q, K, V, dO, kmask, LSE, Dsum, dQ, B, L, Lk, H, Hkv, ldq, ldk, ldv, ldo, lddq, scale, Ofwd, ldout, kstart);
            const dim3 g4(cdiv(L, 64) * Hkv * B);
#define DKV_G4(C_, S_) MANTIS_LAUNCH((attn_bwd_dkv_g4_kernel<C_, S_>), g4, dim3(256), 0, s, Q, K, V, dO, kmask, kstart, qend, LSE, \
                                          Dsum, dK, dV, L, H, Hkv, ldq, ldk, ldv, ldo, lddk, lddv, scale)
            if (causal) { if (kstart) DKV_G4(true, true); else DKV_G4(true, false); }
            else { if (kstart) DKV_G4(false, true); else DKV_G4(false, false); }
#undef DKV_G4
            return mantis_check_launch();
        }
    }
    if (G > 1 && ws == nullptr) return MANTIS_EINVAL;
    const long rows = (long)B * Lk;
    bf16_t* pk = G == 1 ? dK : ws;
    bf16_t* pv = G == 1 ? dV : ws + rows * H * HD;
    const long ldpk = G == 1 ? lddk : (long)H * HD, ldpv = G == 1 ? lddv : (long)H * HD;
    launch_dq<HD>(causal, s, Q, K, V, dO, kmask, LSE, Dsum, dQ, B, L, Lk, H, Hkv, ldq, ldk, ldv, ldo, lddq, scale, Ofwd, ldout, kstart);
    if (causal) {
        MANTIS_LAUNCH((attn_bwd_dkv_kernel<HD, true>), gk, dim3(256), 0, s, Q, K, V, dO, kmask, LSE, Dsum, pk, pv, L, Lk, H, Hkv,
                           ldq, ldk, ldv, ldo, ldpk, ldpv, scale, kstart, qend);
    } else {
        MANTIS_LAUNCH((attn_bwd_dkv_kernel<HD, false>), gk, dim3(256), 0, s, Q, K, V, dO, kmask, LSE, Dsum, pk, pv, L, Lk, H, Hkv,
                           ldq, ldk, ldv, ldo, ldpk, ldpv, scale, kstart, qend);
    }
    if (G > 1) {
        const long total = rows * Hkv * (HD / 8);
        long g = (total + 255) / 256;
        g = g > 2048 ? 2048 : g;
        MANTIS_LAUNCH(attn_group_reduce_kernel, dim3((int)g), dim3(256), 0, s, pk, pv, dK, dV, rows, Hkv, G, HD, (long)H * HD,
                           lddk, lddv);
    }
    return mantis_check_launch();
}

extern "C" {

// Q [B,L,H,hd] (row stride ldq), K,V [B,L,Hkv,hd] (ldk, ldv), kmask int32[B,L] or NULL, O [B,L,H,hd] (ldo),
// LSE fp32 [B,H,L] or NULL.  hd in {16, 64, 72, 96, 128} (96: the Idefics2 perceiver resampler).
static int attn_fwd_impl(const void* Q, const void* K, const void* V, const int32_t* kmask, const int32_t* kstart, void* O, float* LSE,
                         int B, int L, int Lk, int H, int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
                         int causal, void* stream);

int mantis_attn_fwd(const void* Q, const void* K, const void* V, const int32_t* kmask, const int32_t* kstart, void* O, float* LSE,
                    int B, int L, int H, int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, int causal,
                    void* stream) {
    return attn_fwd_impl(Q, K, V, kmask, kstart, O, LSE, B, L, L, H, Hkv, hd, ldq, ldk, ldv, ldo, scale, causal, stream);
}

// Cross attention: Lq query rows per batch entry (Q, O: [B*Lq, ...], LSE [B, H, Lq]) over Lk keys (K, V: [B*Lk, ...], kmask int32 [B, Lk]
// or NULL), non-causal.  The Idefics2 perceiver resampler's attention of 64 latent queries over concat[context, latents]
// (/root/reference/mantis/models/idefics2/modeling_idefics2.py:812-912).
int mantis_attn_fwd_cross(const void* Q, const void* K, const void* V, const int32_t* kmask, void* O, float* LSE, int B, int Lq, int Lk,
                          int H, int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, void* stream) {
    if (Lk <= 0) return MANTIS_EINVAL;
    return attn_fwd_impl(Q, K, V, kmask, nullptr, O, LSE, B, Lq, Lk, H, Hkv, hd, ldq, ldk, ldv, ldo, scale, 0, stream);
}

static int attn_fwd_impl(const void* Q, const void* K, const void* V, const int32_t* kmask, const int32_t* kstart, void* O, float* LSE,
                         int B, int L, int Lk, int H, int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
                         int causal, void* stream) {
    if (B <= 0 || L <= 0 || H <= 0 || Hkv <= 0 || H % Hkv) return MANTIS_EINVAL;
    if (causal && Lk != L) return MANTIS_EINVAL;
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return MANTIS_EUNSUPPORTED;
    if (kstart != nullptr && !causal) return MANTIS_EUNSUPPORTED;     // a lower key bound alone is a block-diagonal mask only under causality
    const dim3 grid(cdiv(L, 128) * H * B);
    hipStream_t s = (hipStream_t)stream;
#define FWD(HD) return launch_fwd<HD>(causal != 0, grid, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, kmask, \
                                      (bf16_t*)O, LSE, L, Lk, H, Hkv, (long)ldq, (long)ldk, (long)ldv, (long)ldo, scale, kstart)
    switch (hd) {
        case 16: FWD(16);
        case 64: FWD(64);
        case 72: FWD(72);
        case 80: FWD(80);
        case 96: FWD(96);
        case 128: FWD(128);
        default: return MANTIS_EUNSUPPORTED;
    }
#undef FWD
}

// 1 if mantis_attn_bwd needs the 2 * B*L*H*hd bf16 workspace for this geometry (per-query-head dK/dV partials), 0 if not
// 0: never needed (MHA; GQA 4:1 at hd 128 always runs the group kernel).  1: pass a workspace (other group sizes choose between the
// group kernel and the per-query-head path by grid size; without a workspace they are held to the group kernel where it exists).
int mantis_attn_bwd_needs_workspace(int H, int Hkv, int hd) { return (H == Hkv || (hd == 128 && H == 4 * Hkv)) ? 0 : 1; }

// Dsum [B,H,L] = rowsum(dO * O)
int mantis_attn_dsum(const void* dO, const void* O, float* Dsum, int B, int L, int H, int hd, int64_t ldo, void* stream) {
    if (hd % 8 || ldo % 8) return MANTIS_EUNSUPPORTED;
    const long rows = (long)B * L;
    MANTIS_LAUNCH(attn_dsum_kernel, dim3(cdiv(rows * H, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dO,
                       (const bf16_t*)O, Dsum, rows, H, hd, L, (long)ldo);
    return mantis_check_launch();
}

// dQ/dK/dV written with row strides lddq/lddk/lddv (head h at column h*hd).  workspace: 2 * B*L*H*hd bf16 when H > Hkv
// (per-query-head dK/dV partials, summed over the GQA group afterwards); unused when H == Hkv and on the GQA-aware path
// (hd 128, H = 4 Hkv; see mantis_attn_bwd_workspace_bytes).  kstart / qend (both or neither): segment bounds of packed samples.
// O (optional): the forward output [B*L, H*hd] (row stride ld_out).  If given, D = rowsum(dO * O) is computed inside the dQ kernel and
// written to Dsum (then a [B,H,L] fp32 scratch/output); if NULL, Dsum must hold it already (mantis_attn_dsum).
static int attn_bwd_impl(const void* Q, const void* K, const void* V, const void* O, const void* dO, const int32_t* kmask,
                         const int32_t* kstart, const int32_t* qend, const float* LSE, float* Dsum, void* dQ, void* dK, void* dV,
                         void* workspace, int B, int L, int Lk, int H, int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv,
                         int64_t ld_out, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, float scale, int causal, void* stream);

int mantis_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const int32_t* kmask,
                    const int32_t* kstart, const int32_t* qend, const float* LSE, float* Dsum, void* dQ, void* dK, void* dV,
                    void* workspace, int B, int L, int H, int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ld_out,
                    int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, float scale, int causal, void* stream) {
    return attn_bwd_impl(Q, K, V, O, dO, kmask, kstart, qend, LSE, Dsum, dQ, dK, dV, workspace, B, L, L, H, Hkv, hd, ldq, ldk, ldv, ld_out,
                         ldo, lddq, lddk, lddv, scale, causal, stream);
}

// Backward of mantis_attn_fwd_cross: dQ [B*Lq, H*hd], dK / dV [B*Lk, Hkv*hd]; Dsum fp32 [B, H, Lq] scratch; workspace 2 * B*Lk*H*hd bf16
// when H > Hkv (per-query-head dK/dV partials, summed over the GQA group afterwards).
int mantis_attn_bwd_cross(const void* Q, const void* K, const void* V, const void* O, const void* dO, const int32_t* kmask,
                          const float* LSE, float* Dsum, void* dQ, void* dK, void* dV, void* workspace, int B, int Lq, int Lk, int H,
                          int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ld_out, int64_t ldo, int64_t lddq,
                          int64_t lddk, int64_t lddv, float scale, void* stream) {
    if (Lk <= 0 || !O) return MANTIS_EINVAL;
    return attn_bwd_impl(Q, K, V, O, dO, kmask, nullptr, nullptr, LSE, Dsum, dQ, dK, dV, workspace, B, Lq, Lk, H, Hkv, hd, ldq, ldk, ldv,
                         ld_out, ldo, lddq, lddk, lddv, scale, 0, stream);
}

static int attn_bwd_impl(const void* Q, const void* K, const void* V, const void* O, const void* dO, const int32_t* kmask,
                         const int32_t* kstart, const int32_t* qend, const float* LSE, float* Dsum, void* dQ, void* dK, void* dV,
                         void* workspace, int B, int L, int Lk, int H, int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv,
                         int64_t ld_out, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, float scale, int causal, void* stream) {
    if (B <= 0 || L <= 0 || H <= 0 || Hkv <= 0 || H % Hkv || !Dsum) return MANTIS_EINVAL;
    if (causal && Lk != L) return MANTIS_EINVAL;
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || lddq % 4 || lddk % 8 || lddv % 8 || (O && ld_out % 8)) return MANTIS_EUNSUPPORTED;
    if ((kstart == nullptr) != (qend == nullptr)) return MANTIS_EINVAL;
    if (kstart != nullptr && !causal) return MANTIS_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define BWD(HD) return launch_bwd<HD>(causal != 0, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dO, \
                                      kmask, LSE, Dsum, (bf16_t*)dQ, (bf16_t*)dK, (bf16_t*)dV, (bf16_t*)workspace, B, L, Lk, H, Hkv, \
                                      (long)ldq, (long)ldk, (long)ldv, (long)ldo, (long)lddq, (long)lddk, (long)lddv, scale, \
                                      (const bf16_t*)O, (long)ld_out, kstart, qend)
    switch (hd) {
        case 16: BWD(16);
        case 64: BWD(64);
        case 96: BWD(96);
        case 128: BWD(128);
        default: return MANTIS_EUNSUPPORTED;
    }
#undef BWD
}

}  // extern "C"
