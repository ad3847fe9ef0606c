// Shared device code of the ring GEMM kernels (gemm.hip: the 256 x 256 ring / ring16 kernels and their K-split finishing pass; gemm176.hip: the
// 176 x 256 kernel): flag bits, compile-time loops, hand-placed LDS reads, the in-place AGPR MFMA, and the LDS-transposed epilogue with its
// fused forms.  See gemm.hip for the reference lines these kernels replace.
#pragma once
#include "common.h"
#include <type_traits>

#define BK 64
#define EPI_BIAS 1
#define EPI_ACT_SHIFT 1
#define EPI_ACT_MASK (7 << EPI_ACT_SHIFT)  // 0 none, 1 gelu(erf), 2 gelu(tanh), 3 quick_gelu
#define EPI_RESIDUAL 16
#define EPI_ACCUM 32
#define EPI_SWIGLU_BWD 64                   // C = [dgate | dup][M, 2N] from dact = A.B^T and residual = [gate | up][M, 2N] (ring kernel)
#define EPI_SUMSQ 128                        // ring16 kernels, through mantis_gemm_bf16_nt_sumsq: tile_sumsq[tile] = sum of squares of the stored tile
#define EPI_VARIANT_SHIFT 8                 // bits 8-11: tile variant (0 = auto)
#define EPI_VARIANT_MASK (15 << EPI_VARIANT_SHIFT)
#define EPI_A_KMAJOR 4096                   // A given as [K, M] (element (m,k) at A[k*lda + m])
#define EPI_B_KMAJOR 8192                   // B given as [K, N]
#define EPI_CUS_SHIFT 16                    // bits 16-27: CU budget this launch is planned for (0 = the default: MANTIS_GEMM_CUS or the whole device)
#define EPI_CUS_MASK (0xFFF << EPI_CUS_SHIFT)
#define EPI_SHARED_GPU 32768                 // the launch shares the GPU with long-running kernels of other queues (RCCL collectives under data parallelism): the
                                            // 176-row kernel then runs one tile per workgroup -- a persistent workgroup that has to wait for a CU someone else
                                            // holds would walk its whole static tile list late (the hardware's own dispatch degrades gracefully instead)
#define EPI_SK_INKERNEL 16384                // ring16 kernels: remainder tiles reduced by their last arriver inside the GEMM kernel (round 4) instead of by
                                            // gemm_ring16_finish_kernel -- same results bit for bit; tests and A/B measurements

typedef __attribute__((address_space(3))) void lds_void;

template <int V> using ic_ = std::integral_constant<int, V>;
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(ic_<I>{});
        static_for<I + 1, N>(f);
    }
}

// cache policy of the operand-tile DMA (aux operand of buffer_load ... lds on gfx950: 1 = sc0, 2 = nt, 16 = sc1).  DMA_AUX: the generic
// 128x128 kernel; DMA_AUX_A / DMA_AUX_B: the ring kernels' A (the M-side operand: activations in every layout of the step) and B (the
// N-side operand: the weight in forward and dX, an activation in dW) streams -- compile-time, so that builds with other policies can be
// A/B-ed (tools/build_probe_lib.sh gemm <name> -DDMA_AUX_A=17 -DDMA_AUX_B=2; measurements: profiles/r04_experiments.md)
#define DMA_AUX 0
#ifndef DMA_AUX_A
#define DMA_AUX_A 0
#endif
#ifndef DMA_AUX_B
#define DMA_AUX_B 0
#endif

__device__ __forceinline__ float gemm_act(float x, int kind) {
    switch (kind) {
        case 1: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        case 2: {
            // 0.5 x (1 + tanh(u)) == x * sigmoid(2u): one exp and one reciprocal instead of tanhf (which was most of the SigLIP fc1 epilogue:
            // 40 us of a 136 us launch, profiles/r04_gemm_anatomy.md); |error| <= 2e-7 |x| before the bf16 rounding
            const float k2 = 2.f * 0.7978845608028654f;
            return x * __builtin_amdgcn_rcpf(1.f + __expf(-k2 * (x + 0.044715f * x * x * x)));
        }
        case 3: return x / (1.f + __expf(-1.702f * x));
        default: return x;
    }
}

typedef __attribute__((ext_vector_type(4))) short gs16x4;
typedef __attribute__((address_space(3))) gs16x4 lds_gs16x4;

// Epilogue of the 256x256 kernel: the MFMA layout gives every lane ONE output row, so storing from registers touches 32 rows x
// 16 B per instruction (measured: 17 us of a 146 us K = 4096 tile, 34 us with a residual).  Instead every wave transposes its
// 128 x 64 tile through a wave-private LDS strip, 64 rows x 64 fp32 per pass (row pitch 272 B: conflict-free ds_write_b128),
// and reads it back as 8 lanes per row x 8 columns per lane: bias / activation / residual / accumulate run on 16-B vectors and
// every global instruction covers 8 whole 128-B lines.  Arithmetic and rounding order are those of gemm_epilogue (bit-identical).
#define EPI_PITCH 272
#define EPI_STRIP (64 * EPI_PITCH)
// Streaming (`nt`) policy for epilogue traffic nobody reads soon: weight gradients (16 GB per step, next touched by the optimizer), the
// [gate | up] pre-activations (kept for the backward), the dX outputs, residual / accumulate / gate|up operands that are read exactly
// once.  They otherwise displace the operand panels of the running and the following GEMMs from L2 / Infinity Cache.  Measured per site
// (profiles/r04_experiments.md 11): family -0.6 ... -1 %; the q|k|v + RoPE output (read by the attention kernel next) and the plain
// forward store (lm_head logits, read by the loss next) measured slower with it and keep the default policy.  -DEPI_NO_STREAMING: off.
template <bool NT>
__device__ __forceinline__ void epi_st16(bf16_t* p, const u32x4& v) {
#ifndef EPI_NO_STREAMING
    if constexpr (NT) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); return; }
#endif
    *reinterpret_cast<u32x4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ u32x4 epi_ld16(const bf16_t* p) {
#ifndef EPI_NO_STREAMING
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#endif
    return *reinterpret_cast<const u32x4*>(p);
}
// Read-back half of the LDS-transposed epilogue, shared by both 256x256 kernels: a wave-private strip holds 64 rows x 64 fp32 columns
// of the result (row pitch 272 B); lane (rr = lane >> 3, cc = lane & 7) takes 8 consecutive columns of row it*8 + rr, applies bias /
// activation / residual / accumulate / the fused SwiGLU backward on 16-B vectors and stores 16 B: every global instruction covers 8
// whole 128-B lines.  Arithmetic and rounding order are those of gemm_epilogue (bit-identical).
template <bool SWIGLU, int NIT = 8>
__device__ __forceinline__ void epi_readback64(const char* __restrict__ strip, bf16_t* __restrict__ C, int M, int N, long ldc,
                                               const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res, long ldr, int flags,
                                               int m_base, int n0w, int lane) {
    const int act = (flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT;
    const bool vec_ok = !(ldc & 7) && !((uintptr_t)C & 15) && (!(flags & (EPI_RESIDUAL | EPI_SWIGLU_BWD)) || (!(ldr & 7) && !((uintptr_t)res & 15)))
                        && (!(flags & EPI_SWIGLU_BWD) || !(N & 7));
    const int rr = lane >> 3, cc = lane & 7;
    const int n = n0w + cc * 8;
#pragma unroll 2
    for (int it = 0; it < NIT; ++it) {
        const int row = it * 8 + rr;
        const int m = m_base + row;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(strip + row * EPI_PITCH + cc * 32);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(strip + row * EPI_PITCH + cc * 32 + 16);
        if (m >= M || n >= N) continue;
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bool full = vec_ok && (n + 8 <= N);
        if (flags & EPI_BIAS) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < N) v[e] += bf2f(bias[n + e]);
        }
        if (act) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gemm_act(bf2f(f2bf(v[e])), act);
        }
        bf16_t* cp = C + (long)m * ldc + n;
        if constexpr (SWIGLU) {
            // SwiGLU backward fused behind dact = dY . W_down (transformers/models/llama/modeling_llama.py:163-176, autograd):
            // dgate = dact * up * silu'(gate), dup = dact * silu(gate); dact rounded to bf16 first, as the unfused path stores it
            if (full) {
                const u32x4 g = epi_ld16<true>(res + (long)m * ldr + n);
                const u32x4 u = epi_ld16<true>(res + (long)m * ldr + N + n);
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
                }
                epi_st16<true>(cp, og);
                epi_st16<true>(cp + N, ou);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (n + e < N) {
                        const float gv = bf2f(res[(long)m * ldr + n + e]), uv = bf2f(res[(long)m * ldr + N + n + e]);
                        const float dv = bf2f(f2bf(v[e]));
                        const float sg = 1.f / (1.f + __expf(-gv));
                        const float silu = gv * sg;
                        cp[e] = f2bf(dv * uv * (sg + silu * (1.f - sg)));
                        cp[N + e] = f2bf(dv * silu);
                    }
                }
            }
            continue;
        }
        if (SWIGLU) continue;
        if (full) {
            if (flags & EPI_RESIDUAL) {
                const u32x4 rv = epi_ld16<false>(res + (long)m * ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = bf2f(f2bf(v[2 * e])) + bf2f_lo(rv[e]);
                    v[2 * e + 1] = bf2f(f2bf(v[2 * e + 1])) + bf2f_hi(rv[e]);
                }
            }
            if (flags & EPI_ACCUM) {
                const u32x4 cv = epi_ld16<false>(cp);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += bf2f_lo(cv[e]);
                    v[2 * e + 1] += bf2f_hi(cv[e]);
                }
            }
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
            epi_st16<false>(cp, o);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
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

// ---- fast read-back of the ring16 kernels (round 4).  The general read-back above takes `flags` at run time: per element it branches on
// the activation kind, fetches the bias with 2-byte loads and -- what cost most -- loads residual / accumulate / SwiGLU operands one
// iteration at a time, each a full memory round trip behind an s_waitcnt vmcnt(0): 22 - 33 us per tile with a residual against 7 us
// without (profiles/r04_gemm_anatomy.md).  Here the epilogue kind is a template parameter (the launcher's run-time switch picks among
// the handful the step uses), the bias is one 16-B load per lane, and the global operands of a whole 64-row pass are requested up front
// -- those of the NEXT pass before the current one is computed.  Same arithmetic and rounding order as epi_readback64: bit-identical.
// Preconditions (else the caller takes the general path): 16-B aligned C / residual, row strides % 8 == 0, all 64 columns inside N.
#define EPRE_NONE 0
#define EPRE_RES 1          // + residual[m, n]
#define EPRE_ACC 2          // + C[m, n] (gradient accumulation)
#define EPRE_SWIGLU 3       // fused SwiGLU backward: residual = [gate | up]
template <int G>
struct EpiPre {
    u32x4 a[G], b[G];
};
// operands of iterations it0 .. it0 + G - 1 of the pass at rows m_base ..
template <int PRE, int G, int GV = G>
__device__ __forceinline__ void epi_fast_prefetch(EpiPre<G>& p, const bf16_t* __restrict__ C, int M, int N, long ldc,
                                                  const bf16_t* __restrict__ res, long ldr, int m_base, int it0, int n, int rr) {
    if constexpr (PRE != EPRE_NONE) {
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            int m = m_base + (it0 + i) * 8 + rr;
            m = m < M ? m : M - 1;                          // rows past M: a harmless in-range address, the store is predicated
            if constexpr (PRE == EPRE_RES) p.a[i] = epi_ld16<true>(res + (long)m * ldr + n);
            if constexpr (PRE == EPRE_ACC) p.a[i] = epi_ld16<true>(C + (long)m * ldc + n);
            if constexpr (PRE == EPRE_SWIGLU) {
                p.a[i] = epi_ld16<true>(res + (long)m * ldr + n);
                p.b[i] = epi_ld16<true>(res + (long)m * ldr + N + n);
            }
        }
    }
}
template <bool BIAS, int ACT, int PRE, int G, bool SS = false, bool NTS = false, int GV = G>
__device__ __forceinline__ void epi_fast_finish(const EpiPre<G>& p, const char* __restrict__ strip, bf16_t* __restrict__ C, int M, int N,
                                                long ldc, const float (&bv)[8], int m_base, int it0, int n, int rr, int cc, float* ss = nullptr) {
#pragma unroll
    for (int i = 0; i < GV; ++i) {
        const int row = (it0 + i) * 8 + rr;
        const int m = m_base + row;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(strip + row * EPI_PITCH + cc * 32);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(strip + row * EPI_PITCH + cc * 32 + 16);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if constexpr (BIAS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[e];
        }
        if constexpr (ACT != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gemm_act(bf2f(f2bf(v[e])), ACT);
        }
        bf16_t* cp = C + (long)m * ldc + n;
        if constexpr (PRE == EPRE_SWIGLU) {
            u32x4 og, ou;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gv[2] = {bf2f_lo(p.a[i][e]), bf2f_hi(p.a[i][e])};
                const float uv[2] = {bf2f_lo(p.b[i][e]), bf2f_hi(p.b[i][e])};
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
            }
            if (m < M) {
                epi_st16<NTS>(cp, og);
                epi_st16<NTS>(cp + N, ou);
            }
        } else {
            if constexpr (PRE == EPRE_RES) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = bf2f(f2bf(v[2 * e])) + bf2f_lo(p.a[i][e]);
                    v[2 * e + 1] = bf2f(f2bf(v[2 * e + 1])) + bf2f_hi(p.a[i][e]);
                }
            }
            if constexpr (PRE == EPRE_ACC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += bf2f_lo(p.a[i][e]);
                    v[2 * e + 1] += bf2f_hi(p.a[i][e]);
                }
            }
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
            if (m < M) {
                epi_st16<NTS>(cp, o);
                if constexpr (SS) {          // sum of squares of what was stored (the bf16 pairs): one v_dot2c_f32_bf16 per pair, fixed order
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        union { unsigned int u; bf16x2_t b; } c;
                        c.u = o[e];
                        *ss = __builtin_amdgcn_fdot2_f32_bf16(c.b, c.b, *ss, false);
                    }
                }
            }
        }
    }
}
// All passes of a wave's tile through its strip: pass p = (pm, pn) covers rows mw0 + pm*64 .., columns nw0 + pn*64 ..  The pass loop is
// unrolled: rolled, all 128 / 256 accumulators stay live to the last pass and the register allocator spills hundreds of them.  The
// global operands travel in groups of G iterations, one group ahead of the arithmetic (G = 8, a whole pass, for residual / accumulate;
// G = 4 for the SwiGLU backward, which holds two operand sets and whose exp-heavy arithmetic covers the loads of the next half pass --
// requesting a whole pass up front and then computing measured 4 % slower on dX(down), HBM bursts instead of a stream).
template <int NBN, int NBM, bool BIAS, int ACT, int PRE, bool SS = false, bool NTS = false, int PB = 4, int SWG = 4>
__device__ __forceinline__ void epi_fast_run(const f32x4 (&acc)[NBN][NBM], char* __restrict__ strip, bf16_t* __restrict__ C, int M, int N,
                                             long ldc, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res, long ldr, int mw0,
                                             int nw0, int lane, float* ss = nullptr) {
    // A pass holds PB 16-row blocks (4: the 64-row strips; 3: the persistent 176-row kernel's 48-row strips) and NBM need not be a multiple of
    // PB (176 rows = 11 blocks): the last pass then holds the rest; a pass is 2 blocks read-back iterations of 8 rows, travelling in groups of
    // G of which gv <= G are live (compile time)
    constexpr int PN = NBN / 4, PM = (NBM + PB - 1) / PB, NPASS = PM * PN;
    constexpr int G = PRE == EPRE_SWIGLU ? SWG : 8, GPP = (2 * PB + G - 1) / G, NG = NPASS * GPP;
    const int rr = lane >> 3, cc = lane & 7;
    char* wr = strip + (lane & 15) * EPI_PITCH + (4 * (lane >> 4)) * 4;
    EpiPre<G> cur, nxt;
    constexpr int GV0 = (2 * (NBM < PB ? NBM : PB)) < G ? (2 * (NBM < PB ? NBM : PB)) : G;
    epi_fast_prefetch<PRE, G, GV0>(cur, C, M, N, ldc, res, ldr, mw0, 0, nw0 + cc * 8, rr);
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    static_for<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value, pass = g / GPP, sub = g % GPP;
        constexpr int pm = pass / PN, pn = pass % PN;
        constexpr int blk = (NBM - PB * pm) < PB ? (NBM - PB * pm) : PB;             // 16-row blocks of this pass
        constexpr int left = 2 * blk - sub * G, gv = left < 0 ? 0 : (left < G ? left : G);
        const int n = nw0 + pn * 64 + cc * 8;
        if constexpr (g + 1 < NG) {
            constexpr int p1 = (g + 1) / GPP, s1 = (g + 1) % GPP;
            constexpr int blk1 = (NBM - PB * (p1 / PN)) < PB ? (NBM - PB * (p1 / PN)) : PB;
            constexpr int left1 = 2 * blk1 - s1 * G, gv1 = left1 < 0 ? 0 : (left1 < G ? left1 : G);
            epi_fast_prefetch<PRE, G, gv1>(nxt, C, M, N, ldc, res, ldr, mw0 + (p1 / PN) * (PB * 16), s1 * G, nw0 + (p1 % PN) * 64 + cc * 8, rr);
        }
        if constexpr (sub == 0) {
            if constexpr (BIAS) {
                const u32x4 b4 = *reinterpret_cast<const u32x4*>(bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bv[2 * e] = bf2f_lo(b4[e]);
                    bv[2 * e + 1] = bf2f_hi(b4[e]);
                }
            }
#pragma unroll
            for (int tm4 = 0; tm4 < blk; ++tm4)
#pragma unroll
                for (int tn4 = 0; tn4 < 4; ++tn4)
                    *reinterpret_cast<f32x4*>(wr + tm4 * 16 * EPI_PITCH + tn4 * 16 * 4) = acc[pn * 4 + tn4][pm * PB + tm4];
        }
        epi_fast_finish<BIAS, ACT, PRE, G, SS, NTS, gv>(cur, strip, C, M, N, ldc, bv, mw0 + pm * (PB * 16), sub * G, n, rr, cc, ss);
        if constexpr (PRE != EPRE_NONE && g + 1 < NG) cur = nxt;
    });
}

template <int OFF>
__device__ __forceinline__ void lds_read_b128_v(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr64_h(bf16x8& dst, unsigned addr, int h) {
    // one transposing half read (4 k) into the low (h = 0) or high (h = 1) half of the 8-k fragment
    if (h == 0) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(reinterpret_cast<gs16x4*>(&dst)[0]) : "v"(addr), "i"(OFF));
    else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(reinterpret_cast<gs16x4*>(&dst)[1]) : "v"(addr), "i"(OFF + 1024));
}
// accumulators live in AGPRs, updated in place: the compiler neither moves them nor pads hazards around these (the loop has none:
// consecutive MFMAs never share an accumulator, fragment registers are rewritten only by LDS reads issued >= 16 MFMAs later)
__device__ __forceinline__ void mfma16(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

#define PAIR_NONE 0
#define PAIR_SWIGLU 1
#define PAIR_ROPE 2
// Epilogue of a ring16 wave tile: all of it (NBM_ = 8, from the GEMM kernel) or one 64-row half of it (NBM_ = 4, from the K-split finishing
// kernel).  acc[tn][tm] = 16 x 16 block (n block tn, m block tm) of the wave's 128 (64) x NBN*16 tile whose first row is mw0; wave_l / tid_l =
// the wave's / thread's index inside its block (strip ownership, the sumsq reduction: that one needs all NW waves of the tile in the block);
// KM: a K-major operand layout, i.e. a dX / dW launch whose plain result streams out.
// (round 6: NWV waves per tile, NBN x NBM_ blocks per wave as explicit parameters -- the ring16 kernels are (4, 8, 8) and (8, 4, 8), their
// finishing passes (., ., 4), the 176-row kernel (4, 4, 11): a last pass of NBM_ % 4 blocks, and the caller passes M clipped to its tile)
template <int NWV, int NBN, int NBM_, bool KM, bool SWIGLU, int PAIR, bool FIN, int PB = 4, int FASTSW = 0>
__device__ __forceinline__ void ring_epilogue(const f32x4 (&acc)[NBN][NBM_], char* __restrict__ smem, int wave_l, int tid_l, int lane,
                                                int wn, bf16_t* __restrict__ C, int M, int N, long ldc, const bf16_t* __restrict__ bias,
                                                const bf16_t* __restrict__ res, long ldr, int flags, int mw0, int n0, int tile_id,
                                                bf16_t* __restrict__ aux0, const bf16_t* __restrict__ aux1, long aux_ld, int aux_n) {
    // epilogue: the wave tile goes through the wave-private strip in 64 x 64 passes.  Block (tn, tm): lane l holds row tm*16 + (l & 15),
    // columns tn*16 + 4*(l >> 4) .. + 3 -> one 16-B strip write per block (8 consecutive lanes = 8 rows of pitch 272 B: conflict-free)
    constexpr bool TWO = NBN == 8;                        // 128-column wave tiles: two strips per wave in the pair modes
    constexpr int PM = (NBM_ + PB - 1) / PB;              // passes of PB blocks (the last one may hold fewer)
    constexpr int STRIP = PB * 16 * EPI_PITCH;            // a wave's strip: PB * 16 rows (PB = 4: EPI_STRIP)
    char* strip = smem + wave_l * STRIP;
    const int nw0 = n0 + wn * (NBN * 16);
    const int pair_dist = PAIR == PAIR_SWIGLU ? (N >> 1) : 64;
    auto pair_first = [&](int phi) { return PAIR == PAIR_SWIGLU ? (n0 >> 1) + phi : n0 + (phi >> 6) * 128 + (phi & 63); };
    if constexpr (PAIR != PAIR_NONE) {
        // 4 waves: two strips per wave (8 x 17 KiB <= 160 KiB), A = the first columns, B = the second columns of the wave's 64 features; per
        //          64-row pass, lane (rr = lane >> 3, cc = lane & 7) meets both columns of (row it*8 + rr, features cc*8 ..) at the same
        //          position of the two strips
        // 8 waves: one strip per wave (8 x 17 KiB): its 64 columns are [32 first | 32 second] -- the plain strip fill; per 64-row pass, lane
        //          (r16 = lane >> 2, fg = lane & 3) reads the first columns at fg*8 and the second ones at 32 + fg*8 of row it*16 + r16
        constexpr int G = NBN * 8;
        char* sa = smem + (TWO ? 2 * wave_l : wave_l) * STRIP;
        char* sb = TWO ? sa + STRIP : sa + 32 * 4;
        const int phi0 = wn * G;
        const int rr = TWO ? lane >> 3 : lane >> 2, cc = TWO ? lane & 7 : lane & 3;
        constexpr int RPI = TWO ? 8 : 16;                  // rows per read-back iteration
        const bool has_bias = flags & EPI_BIAS;
#pragma unroll
        for (int pass = 0; pass < PM; ++pass) {
            const int blk = (NBM_ - PB * pass) < PB ? (NBM_ - PB * pass) : PB;      // (compile-time after unrolling)
#pragma unroll
            for (int tm4 = 0; tm4 < PB; ++tm4)
#pragma unroll
                for (int tn4 = 0; tn4 < 4; ++tn4) {
                    if (tm4 >= blk) continue;
                    const int off = (tm4 * 16 + (lane & 15)) * EPI_PITCH + (tn4 * 16 + 4 * (lane >> 4)) * 4;
                    if constexpr (TWO) {
                        *reinterpret_cast<f32x4*>(sa + off) = acc[tn4][pass * PB + tm4];
                        *reinterpret_cast<f32x4*>(sb + off) = acc[4 + tn4][pass * PB + tm4];
                    } else {
                        *reinterpret_cast<f32x4*>(sa + off) = acc[tn4][pass * PB + tm4];      // blocks 0,1 = first, 2,3 = second columns
                    }
                }
            const int phi = phi0 + cc * 8;
            const int col1 = pair_first(phi), col2 = col1 + pair_dist;
            float b1[8], b2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                b1[e] = has_bias ? bf2f(bias[col1 + e]) : 0.f;
                b2[e] = has_bias ? bf2f(bias[col2 + e]) : 0.f;
            }
            // rotary tables of the whole pass requested up front (one memory round trip per pass instead of one per iteration)
            u32x4 vcs[64 / RPI], vss[64 / RPI];
            const bool rot = PAIR == PAIR_ROPE && col1 < aux_n;
            if constexpr (PAIR == PAIR_ROPE) {
                if (rot) {
#pragma unroll
                    for (int it = 0; it < 64 / RPI; ++it) {
                        if (it * RPI >= blk * 16) continue;
                        int m = mw0 + pass * (PB * 16) + it * RPI + rr;
                        m = m < M ? m : M - 1;
                        vcs[it] = *reinterpret_cast<const u32x4*>(aux0 + (long)m * aux_ld + (phi & 63));
                        vss[it] = *reinterpret_cast<const u32x4*>(aux1 + (long)m * aux_ld + (phi & 63));
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 64 / RPI; ++it) {
                if (it * RPI >= blk * 16) continue;
                const int row = it * RPI + rr;
                const int m = mw0 + pass * (PB * 16) + row;
                const f32x4 alo = *reinterpret_cast<const f32x4*>(sa + row * EPI_PITCH + cc * 32);
                const f32x4 ahi = *reinterpret_cast<const f32x4*>(sa + row * EPI_PITCH + cc * 32 + 16);
                const f32x4 blo = *reinterpret_cast<const f32x4*>(sb + row * EPI_PITCH + cc * 32);
                const f32x4 bhi = *reinterpret_cast<const f32x4*>(sb + row * EPI_PITCH + cc * 32 + 16);
                if (m >= M) continue;
                float v1[8] = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
                float v2[8] = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
                if (has_bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v1[e] += b1[e];
                        v2[e] += b2[e];
                    }
                }
                u32x4 o1, o2;
                if constexpr (PAIR == PAIR_SWIGLU) {
                    // gate, up rounded to bf16 as the unfused GEMM stores them; a = bf16(bf16(silu(gate)) * up) as swiglu_fwd_kernel
                    u32x4 oa;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o1[e] = pack_bf2(v1[2 * e], v1[2 * e + 1]);
                        o2[e] = pack_bf2(v2[2 * e], v2[2 * e + 1]);
                        const float g0 = bf2f_lo(o1[e]), g1 = bf2f_hi(o1[e]);
                        const float s0 = bf2f(f2bf(g0 * (1.f / (1.f + __expf(-g0))))), s1 = bf2f(f2bf(g1 * (1.f / (1.f + __expf(-g1)))));
                        oa[e] = pack_bf2(s0 * bf2f_lo(o2[e]), s1 * bf2f_hi(o2[e]));
                    }
                    epi_st16<false>(aux0 + (long)m * aux_ld + col1, oa);      // read by the down projection next
                } else {
                    const u32x4 vc = vcs[it], vs = vss[it];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned w1 = pack_bf2(v1[2 * e], v1[2 * e + 1]);
                        const unsigned w2 = pack_bf2(v2[2 * e], v2[2 * e + 1]);
                        if (rot) {
                            const float x1[2] = {bf2f_lo(w1), bf2f_hi(w1)}, x2[2] = {bf2f_lo(w2), bf2f_hi(w2)};
                            const float cs[2] = {bf2f_lo(vc[e]), bf2f_hi(vc[e])}, sn[2] = {bf2f_lo(vs[e]), bf2f_hi(vs[e])};
                            float r1[2], r2[2];
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                r1[h2] = bf2f(f2bf(x1[h2] * cs[h2])) - bf2f(f2bf(x2[h2] * sn[h2]));
                                r2[h2] = bf2f(f2bf(x2[h2] * cs[h2])) + bf2f(f2bf(x1[h2] * sn[h2]));
                            }
                            o1[e] = pack_bf2(r1[0], r1[1]);
                            o2[e] = pack_bf2(r2[0], r2[1]);
                        } else {
                            o1[e] = w1;
                            o2[e] = w2;
                        }
                    }
                }
                epi_st16<PAIR == PAIR_SWIGLU>(C + (long)m * ldc + col1, o1);      // [gate | up]: kept for the backward; q|k|v: read next
                epi_st16<PAIR == PAIR_SWIGLU>(C + (long)m * ldc + col2, o2);
            }
        }
        return;
    }
    {
        // fast read-back (see epi_fast_run) when every column of this wave's tile is inside N and the operands are 16-B vectors; the
        // epilogue kinds of the step are compile-time variants, anything else takes the general path below
        const bool needs_res = flags & (EPI_RESIDUAL | EPI_SWIGLU_BWD);
        const bool vec_ok = !(ldc & 7) && !((uintptr_t)C & 15) && (!needs_res || (!(ldr & 7) && !((uintptr_t)res & 15))) &&
                            (!(flags & EPI_BIAS) || !((uintptr_t)bias & 15)) && (!(flags & EPI_SWIGLU_BWD) || !(N & 7));
        if (vec_ok && nw0 + NBN * 16 <= N) {
#define EPI_FAST(B_, A_, P_, NT_) epi_fast_run<NBN, NBM_, B_, A_, P_, false, NT_, PB>(acc, strip, C, M, N, ldc, bias, res, ldr, mw0, nw0, lane)
            if constexpr (!SWIGLU && PAIR == PAIR_NONE) {
                if (flags & EPI_SUMSQ) {
                    // weight-gradient launches of mantis_gemm_bf16_nt_sumsq: the squared norm of the stored tile rides along (the optimizer's
                    // global gradient norm then needs no pass of its own over these 16 GB).  Lane partials in a fixed order, DPP wave sum,
                    // waves summed in order by thread 0: deterministic.  The entry point guarantees the fast path for every wave.
                    float ss = 0.f;
                    if (flags & EPI_ACCUM) epi_fast_run<NBN, NBM_, false, 0, EPRE_ACC, true, true, PB>(acc, strip, C, M, N, ldc, bias, res, ldr, mw0, nw0, lane, &ss);
                    else epi_fast_run<NBN, NBM_, false, 0, EPRE_NONE, true, true, PB>(acc, strip, C, M, N, ldc, bias, res, ldr, mw0, nw0, lane, &ss);
                    ss = wave_sum(ss);
                    float* red = reinterpret_cast<float*>(smem + NWV * STRIP);
                    if (lane == 0) red[wave_l] = ss;
                    __syncthreads();
                    if (tid_l == 0) {
                        float t = 0.f;
#pragma unroll
                        for (int w = 0; w < NWV; ++w) t += red[w];
                        // finishing kernel: the tile's two 64-row halves are two workgroups -- two addends on a slot the GEMM kernel zeroed:
                        // a + b == b + a bit for bit, so the atomic adds are deterministic
                        if constexpr (FIN) __hip_atomic_fetch_add(reinterpret_cast<float*>(aux0) + tile_id, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else reinterpret_cast<float*>(aux0)[tile_id] = t;
                    }
                    return;
                }
            }
            if constexpr (SWIGLU && FASTSW != 0) {
                // the 176-row kernel (one wave per SIMD, nothing else on the CU to hide a memory round trip): the SwiGLU backward on the fast
                // read-back, gate | up of FASTSW iterations (4 or 8) requested one group ahead -- measured per tile on few CUs:
                // profiles/r06_experiments.md
                epi_fast_run<NBN, NBM_, false, 0, EPRE_SWIGLU, false, true, PB, FASTSW>(acc, strip, C, M, N, ldc, bias, res, ldr, mw0, nw0, lane);
                return;
            }
            if constexpr (!SWIGLU) {      // the SwiGLU-backward read-back stays on the general path: its tile moves 512 KiB (gate, up in; dgate, dup
                                          // out) and is HBM-bound either way; requesting operands ahead measured 3 - 4 % SLOWER on dX(down) (bursts)
                switch (flags & (EPI_BIAS | EPI_ACT_MASK | EPI_RESIDUAL | EPI_ACCUM)) {
                    case 0: EPI_FAST(false, 0, EPRE_NONE, KM); return;      // dX / dW stream out; a plain forward store is read next
                    case EPI_RESIDUAL: EPI_FAST(false, 0, EPRE_RES, true); return;
                    case EPI_ACCUM: EPI_FAST(false, 0, EPRE_ACC, true); return;
                    case EPI_BIAS: EPI_FAST(true, 0, EPRE_NONE, false); return;
                    case EPI_BIAS | EPI_RESIDUAL: EPI_FAST(true, 0, EPRE_RES, false); return;
                    case EPI_BIAS | (1 << EPI_ACT_SHIFT): EPI_FAST(true, 1, EPRE_NONE, false); return;
                    case EPI_BIAS | (2 << EPI_ACT_SHIFT): EPI_FAST(true, 2, EPRE_NONE, false); return;
                    case EPI_BIAS | (3 << EPI_ACT_SHIFT): EPI_FAST(true, 3, EPRE_NONE, false); return;
                    default: break;
                }
            }
#undef EPI_FAST
        }
    }
#pragma unroll
    for (int pm = 0; pm < PM; ++pm)
#pragma unroll
        for (int pn = 0; pn < NBN / 4; ++pn) {
#pragma unroll
            for (int tm4 = 0; tm4 < PB; ++tm4)
#pragma unroll
                for (int tn4 = 0; tn4 < 4; ++tn4) {
                    if (pm * PB + tm4 >= NBM_) continue;      // the short last pass: rows past the tile are cut off by the caller's M
                    *reinterpret_cast<f32x4*>(strip + (tm4 * 16 + (lane & 15)) * EPI_PITCH + (tn4 * 16 + 4 * (lane >> 4)) * 4) =
                        acc[pn * 4 + tn4][pm * PB + tm4];
                }
            epi_readback64<SWIGLU, 2 * PB>(strip, C, M, N, ldc, bias, res, ldr, flags, mw0 + pm * (PB * 16), nw0 + pn * 64, lane);
        }
}

template <int NW, int NBM_, bool KM, bool SWIGLU, int PAIR, bool FIN>
__device__ __forceinline__ void ring16_epilogue(const f32x4 (&acc)[NW == 4 ? 8 : 4][NBM_], char* __restrict__ smem, int wave_l, int tid_l, int lane,
                                                int wn, bf16_t* __restrict__ C, int M, int N, long ldc, const bf16_t* __restrict__ bias,
                                                const bf16_t* __restrict__ res, long ldr, int flags, int mw0, int n0, int tile_id,
                                                bf16_t* __restrict__ aux0, const bf16_t* __restrict__ aux1, long aux_ld, int aux_n) {
    ring_epilogue<NW, (NW == 4 ? 8 : 4), NBM_, KM, SWIGLU, PAIR, FIN>(acc, smem, wave_l, tid_l, lane, wn, C, M, N, ldc, bias, res, ldr, flags, mw0, n0,
                                                                    tile_id, aux0, aux1, aux_ld, aux_n);
}

