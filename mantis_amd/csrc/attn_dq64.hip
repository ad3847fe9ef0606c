// Attention backward, dQ, head dimension 128, 64 query rows per wave (round 4).  gfx950 only.
//
// Reference semantics: autograd of transformers/models/llama/modeling_llama.py:191-214 (eager_attention_forward: softmax(Q K^T / sqrt(d)
// + mask) V) with respect to Q; same operands, masks and packed-sample bounds as attn_bwd_dq_kernel (attn.hip), which stays the kernel
// for every other head dimension and for short sequences.
//
// Why another kernel.  attn_bwd_dq_kernel gives a wave 32 query rows: it reads every K / V / K^T fragment from LDS for 32 rows of work --
// at two waves per SIMD the LDS read time of a 64-key tile equals its MFMA time -- each read sits right in front of its consumer, and the
// 256-register budget leaves no room to request them earlier (profiles/r04_experiments.md 14).  Here a wave owns 64 query rows as two
// 32-row blocks, so every fragment read feeds two MFMAs, and the workgroup (4 waves, 256 query rows) is alone on its CU with 512 registers
// per lane.  What the MFMAs alone touch fills the accumulator file exactly -- dQ accumulators a[0:127], Q fragments a[128:191], dO
// fragments a[192:255] -- and the compiler cannot allocate a file with no slack (it copies the fragments to VGPRs in front of every MFMA
// and spills: experiment 14b).  So the AGPRs are owned BY NUMBER by the inline-asm MFMAs (the compiler never sees them; this file is
// built with -amdgpu-spill-vgpr-to-agpr=0 and tools/attn_dq64_audit.py checks the ISA), the scores of both 32-key halves of a tile (128
// VGPRs), the fragment rings and the arithmetic stay compiler-allocated, and the tile is written slot by slot:
//   96 MFMAs per 64-key tile: i = 0..31 S, dP of keys 0-31 | 32..63 S, dP of keys 32-63 | 64..79 dQ from keys 0-31 | 80..95 dQ from keys 32-63;
//   after each MFMA a fenced slot (sched_barrier) holds the LDS reads of fragments needed 5 - 9 MFMAs later and the softmax-backward
//   arithmetic of score elements whose last MFMA is at least two MFMAs back (the hardware does not interlock an MFMA result against a
//   VALU read; two 8-pass MFMAs issued behind it cover its latency).
// Same arithmetic per element and the same summation order over the keys as attn_bwd_dq_kernel: bit-identical results.
#include "attn_common.h"

namespace {

template <int V> using ic_ = std::integral_constant<int, V>;

// accumulator-file map (by register number)
constexpr int A_ACC = 0;        // dQ^T accumulators: (row block rb, d-block d) at a[A_ACC + 16 (4 rb + d) .. + 15]
constexpr int A_Q = 128;        // Q fragments: (rb, ks) at a[A_Q + 4 (8 rb + ks) .. + 3]
constexpr int A_DO = 192;       // dO fragments: (rb, ks) at a[A_DO + 4 (8 rb + ks) .. + 3]

#define DQ64_AGPR_ALL \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", \
    "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", \
    "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", \
    "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", \
    "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", \
    "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", \
    "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", \
    "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", \
    "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", \
    "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", \
    "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", \
    "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", \
    "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", \
    "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
    "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", \
    "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", \
    "a252", "a253", "a254", "a255"

template <int R>
__device__ __forceinline__ void agpr_write(unsigned v) {
    asm volatile("v_accvgpr_write_b32 a[%0], %1" : : "n"(R), "v"(v));
}
template <int R>
__device__ __forceinline__ float agpr_read() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(R));
    return v;
}
template <int BASE>             // a[BASE .. BASE + 3] <- a 16-byte fragment
__device__ __forceinline__ void agpr_write4(const u32x4& v) {
    agpr_write<BASE>(v[0]);
    agpr_write<BASE + 1>(v[1]);
    agpr_write<BASE + 2>(v[2]);
    agpr_write<BASE + 3>(v[3]);
}
// d (VGPRs) = a . a[B : B + 3]   /   d += a . a[B : B + 3]
template <int B>
__device__ __forceinline__ void mfma_s_first(f32x16& d, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], 0" : "=&v"(d) : "v"(a), "n"(B), "n"(B + 3));
}
template <int B>
__device__ __forceinline__ void mfma_s_next(f32x16& d, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(d) : "v"(a), "n"(B), "n"(B + 3));
}
// a[C : C + 15] += a . b
template <int C>
__device__ __forceinline__ void mfma_acc(const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" : : "n"(C), "n"(C + 15), "v"(a), "v"(b));
}

// every LDS read issued so far has landed; a and b (destinations of hand-issued reads) may be used from here on
__device__ __forceinline__ void lds_wait2(bf16x8& a, bf16x8& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}

}  // namespace

#ifdef DQ64_STAMPS
// timing probe (tools/build_probe_lib.sh attn_dq64 stamps -DDQ64_STAMPS; tools/attn_dq64_anatomy.py; never in the product build): wave 0
// of every workgroup sums, over its unmasked tiles, the s_memtime intervals tile top -> MFMA 31 -> 63 -> 79 -> 95 -> behind the barrier,
// and records entry / loop start / exit.  (s_memtime returns through lgkmcnt: every stamp also drains the LDS reads in flight.)
__device__ unsigned long long g_dq64_stamps[4096 * 12];
#define DQ64_STAMP(var) do { var = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DQ64_STAMP(var) do { } while (0)
#endif

template <bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                               const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                               const float* __restrict__ LSE,
                                                               float* __restrict__ Dsum, bf16_t* __restrict__ dQ, int L, int Lk, int H,
                                                               int Hkv, long ldq, long ldk, long ldv, long ldo, long lddq, float scale,
                                                               const bf16_t* __restrict__ Ofwd, long ldout,
                                                               const int* __restrict__ kstart) {
    constexpr int HD = 128;
    using C = AttnCfg<HD>;
    using Y = Lay<HD>;
    constexpr int TILE = 64 * Y::PITCH;
    constexpr int BUF = 2 * TILE + 64 * 4 + 16;
    constexpr int NB = 4;             // LDS ring of K / V tiles: a tile's compute (~1.6 us) is shorter than its DMA round trip, three are kept in flight
    __shared__ __attribute__((aligned(16))) char smem[NB * BUF];

    asm volatile("" ::: DQ64_AGPR_ALL);      // declares the whole accumulator file as used: the kernel descriptor must allocate it
#ifdef DQ64_STAMPS
    unsigned long long st_entry, st_loop = 0, st_t0 = 0, st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_e = 0;
    unsigned long long sum_a = 0, sum_b = 0, sum_c = 0, sum_d = 0, sum_e = 0, n_plain = 0, n_masked = 0, n_skipped = 0;
    DQ64_STAMP(st_entry);
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5, lq = lane & 31;
    const int gx = (L + 255) >> 8;
    int bx, h, b;
    xcd_tile_map(gx, H, bx, h, b, CAUSAL);
    const int hk = h / (H / Hkv);
    const int qb = CAUSAL ? (gx - 1 - bx) : bx;
    const int qblk0 = qb * 256, q0 = qblk0 + wave * 64;
    const float c = scale * LOG2E;
    int q[2], qc[2], ks_q[2];
    float lse2[2], dsum[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        q[rb] = q0 + rb * 32 + lq;
        qc[rb] = q[rb] < L ? q[rb] : L - 1;
        ks_q[rb] = kstart ? kstart[(long)b * L + qc[rb]] : 0;
        lse2[rb] = LSE[((long)b * H + h) * L + qc[rb]] * LOG2E;
    }
    const int ks_hi = kstart ? kstart[(long)b * L + (q0 + 63 < L ? q0 + 63 : L - 1)] : 0;
    const int t_first = kstart ? (kstart[(long)b * L + (qblk0 < L ? qblk0 : L - 1)] >> 6) : 0;

    // Q and dO fragments -> their AGPRs; D = rowsum(dO * O) is either given or computed here from the dO rows on the way (and published
    // for the dK/dV kernel that runs after this one), exactly as attn_bwd_dq_kernel does
    // (all loads of a row block are issued before the first v_accvgpr_write: the writes are asm statements, and a load consumed by one
    // right away is waited for on the spot -- 48 serial memory round trips per workgroup in the first version of this prologue)
    static_for<0, 2>([&](auto rbc) {
        constexpr int rb = decltype(rbc)::value;
        u32x4 qv[C::NKS], dv[C::NKS], ov[C::NKS];
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) {
            const int ch = ks * 2 + hh;
            qv[ks] = *reinterpret_cast<const u32x4*>(Q + ((long)b * L + qc[rb]) * ldq + (long)h * HD + ch * 8);
            dv[ks] = *reinterpret_cast<const u32x4*>(dO + ((long)b * L + qc[rb]) * ldo + (long)h * HD + ch * 8);
            if (Ofwd != nullptr) ov[ks] = *reinterpret_cast<const u32x4*>(Ofwd + ((long)b * L + qc[rb]) * ldout + (long)h * HD + ch * 8);
        }
        if (Ofwd != nullptr) {
            float part = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks)
#pragma unroll
                for (int e = 0; e < 4; ++e) part += bf2f_lo(ov[ks][e]) * bf2f_lo(dv[ks][e]) + bf2f_hi(ov[ks][e]) * bf2f_hi(dv[ks][e]);
            dsum[rb] = part + __shfl_xor(part, 32, 64);
            if (hh == 0 && q[rb] < L) Dsum[((long)b * H + h) * L + q[rb]] = dsum[rb];
        } else {
            dsum[rb] = Dsum[((long)b * H + h) * L + qc[rb]];
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, C::NKS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            agpr_write4<A_Q + 4 * (8 * rb + ks)>(qv[ks]);
            agpr_write4<A_DO + 4 * (8 * rb + ks)>(dv[ks]);
        });
    });
    static_for<0, 128>([&](auto rc) { agpr_write<A_ACC + decltype(rc)::value>(0u); });

    const int kend = CAUSAL ? (qblk0 + 256 < L ? qblk0 + 256 : L) : Lk;
    const int ntiles = (kend + 63) / 64;
    const bf16_t* Kb = K + (long)b * Lk * ldk + (long)hk * HD;
    const bf16_t* Vb = V + (long)b * Lk * ldv + (long)hk * HD;

    // tile t -> LDS buffer t % NB by LDS-DMA, three tiles ahead of its use.  Every stage() call issues the same number of VMEM operations
    // per wave (8 DMA pieces), so "tile t + 1 has landed" is a counted wait that leaves the two newest stages in flight; past the last
    // tile the last one is staged once more (never read) to keep that count.
    // No key-padding mask here (batches with one take attn_bwd_dq_kernel): a mask word loaded in one iteration and used in the next is a
    // loop-carried VGPR written by a VMEM load, in front of whose copy at the loop latch the compiler puts s_waitcnt vmcnt(0) -- every
    // iteration, mask or not -- which drains the tile ring.  Key liveness is therefore arithmetic: key < Lk.
    auto key_word = [&](int t) -> int {
        const int key = t * 64 + (int)threadIdx.x;
        return (threadIdx.x < 64 && key < Lk) ? 1 : 0;
    };
    int word_next = key_word(t_first);
    auto stage = [&](int t) {
        const int key0 = (t < ntiles ? t : ntiles - 1) * 64;
        char* base = smem + (t & (NB - 1)) * BUF;
        const bool ok = word_next != 0;
        word_next = key_word(t + 1);
        stage_tile_dma<64>(Kb + (long)key0 * ldk, ldk, Lk - key0, base);
        stage_tile_dma<64>(Vb + (long)key0 * ldv, ldv, Lk - key0, base + TILE);
        if (threadIdx.x < 64) {     // wave 0: additive key bias (0 / -inf) + one flag "this tile has a masked key"
            float* bp = reinterpret_cast<float*>(base + 2 * TILE);
            bp[threadIdx.x] = ok ? 0.f : -INFINITY;
            const unsigned long long okm = __ballot(ok);
            if (threadIdx.x == 0) bp[64] = (okm == ~0ull) ? 0.f : 1.f;
        }
    };
    auto wait_two_stages_in_flight = [&]() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); };
    stage(t_first);
    stage(t_first + 1);
    stage(t_first + 2);
    wait_two_stages_in_flight();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 2" ::: "memory");      // the bias row's ds_write; s_nop: v_accvgpr_write -> MFMA operand
    __builtin_amdgcn_s_barrier();              // raw: __syncthreads() would add s_waitcnt vmcnt(0) and drain the ring
    __builtin_amdgcn_sched_barrier(0);

    for (int t = t_first; t < ntiles; ++t) {
        const int key0 = t * 64;
        const char* sK = smem + (t & (NB - 1)) * BUF;
        const char* sV = sK + TILE;
        const float* sBias = reinterpret_cast<const float*>(sK + 2 * TILE);
        stage(t + 3);                  // into the buffer tile t - 1 has released (every wave is past the barrier that ended it)
#ifdef DQ64_STAMPS
        if (t == t_first) DQ64_STAMP(st_loop);
        DQ64_STAMP(st_t0);
        bool st_plain = false;
#endif
        if (!(CAUSAL && key0 > q0 + 63)) {
            const bool need_mask = (CAUSAL && key0 + 63 > q0) || sBias[64] != 0.f || key0 < ks_hi;   // wave-uniform
            auto tile_body = [&](auto maskc) {
                constexpr bool MASK = decltype(maskc)::value;
                f32x16 s[2][2], dp[2][2];            // [32-key half][row block]: scores (dS in place after the softmax backward), dP
                bf16x8 kf[4], vf[4];                 // K / V fragment ring: step g = 8 half + ks lives in slot g % 4
                bf16x8 ktf[4];                       // K^T fragment ring: fragment F = 8 half + 4 cp + d lives in slot F % 4
                bf16x8 dsf[2][2];                    // packed dS of the half being multiplied: [cp][row block]
                float tb[3];                         // half 0: one element per slot, three stages in three consecutive slots
                float tc[3][2];                      // half 1: one element PAIR per slot
                auto load_kv = [&](auto gc) {
                    constexpr int g = decltype(gc)::value, sb = g >> 3, ks = g & 7;
                    const int off = Y::chunk_off(sb * 32 + lq, ks * 2 + hh);
                    kf[g % 4] = *reinterpret_cast<const bf16x8*>(sK + off);
                    vf[g % 4] = *reinterpret_cast<const bf16x8*>(sV + off);
                };
                // the transposing reads go through inline asm: in front of the builtin the compiler puts s_waitcnt vmcnt(0) (it cannot tell
                // the read apart from the LDS-DMA pieces in flight), which drains the tile ring once per tile.  Their own completion is
                // waited for by hand: lgkmcnt(0) in front of the first MFMA that uses a batch
                auto load_kt = [&](auto Fc) {
                    constexpr int F = decltype(Fc)::value, sb = F >> 3, f = F & 7;
                    const int row0 = sb * 32 + 16 * (f >> 2), col0 = (f & 3) * 32;
                    const int sl = lane & 15, g16 = (lane >> 4) & 1, h2 = lane >> 5;
                    const int row = row0 + 4 * h2 + (sl >> 2), col = col0 + 16 * g16 + (sl & 3) * 4;
                    const int off = Y::chunk_off(row, col >> 3) + (col & 7) * 2;
                    const int off8 = (off + 8 * Y::PITCH) ^ 32;       // see read_tr_frag
                    const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(sK + off);
                    const unsigned a1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(sK + off8);
                    s16x4 lo, hi;
                    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1));
                    union { s16x4 h[2]; bf16x8 f8; } u;
                    u.h[0] = lo;
                    u.h[1] = hi;
                    ktf[F % 4] = u.f8;
                };
                auto wait_kt = [&](auto Fc) {          // fragments F, F + 1 have landed
                    constexpr int F = decltype(Fc)::value;
                    lds_wait2(ktf[F % 4], ktf[(F + 1) % 4]);
                };
                // one score element -> dS in three stages (attn_bwd_dq_kernel's arithmetic): no stage depends on a result of its own slot
                auto stage1 = [&](auto sbc, auto rbc, auto rc) -> float {          // exponent: score * scale * log2(e) [masked] - lse
                    constexpr int sb = decltype(sbc)::value, rb = decltype(rbc)::value, r = decltype(rc)::value;
                    float v = s[sb][rb][r] * c;
                    if constexpr (MASK) {
                        const int kl = sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        v += sBias[kl];
                        if (CAUSAL && key0 + kl > q[rb]) v = -INFINITY;
                        if (key0 + kl < ks_q[rb]) v = -INFINITY;
                    }
                    return v - lse2[rb];                                           // lse = +inf for fully masked rows -> p = 0
                };
                auto stage3 = [&](auto sbc, auto rbc, auto rc, float p) {          // dS = p (dP - D): the softmax scale goes to the accumulators
                    constexpr int sb = decltype(sbc)::value, rb = decltype(rbc)::value, r = decltype(rc)::value;
                    s[sb][rb][r] = p * (dp[sb][rb][r] - dsum[rb]);
                };
                auto pack = [&](auto sbc, auto cpc, auto rbc) {
                    constexpr int sb = decltype(sbc)::value, cp = decltype(cpc)::value, rb = decltype(rbc)::value;
                    dsf[cp][rb] = pack_frag(s[sb][rb], cp);
                };
                load_kv(ic_<0>{});
                load_kv(ic_<1>{});
                load_kv(ic_<2>{});
                load_kv(ic_<3>{});
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 96>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    // ---- the MFMA of slot i
                    if constexpr (i < 64) {
                        constexpr int g = i >> 2, sb = g >> 3, ks = g & 7, j = i & 3, rb = j >> 1;
                        if constexpr ((j & 1) == 0) {
                            if constexpr (ks == 0) mfma_s_first<A_Q + 4 * (8 * rb + ks)>(s[sb][rb], kf[g % 4]);
                            else mfma_s_next<A_Q + 4 * (8 * rb + ks)>(s[sb][rb], kf[g % 4]);
                        } else {
                            if constexpr (ks == 0) mfma_s_first<A_DO + 4 * (8 * rb + ks)>(dp[sb][rb], vf[g % 4]);
                            else mfma_s_next<A_DO + 4 * (8 * rb + ks)>(dp[sb][rb], vf[g % 4]);
                        }
                    } else {
                        // K^T fragment F = 8 half + 4 cp + d (0 .. 15) feeds MFMAs 64 + 2 F (row block 0) and 65 + 2 F (row block 1)
                        constexpr int m = i - 64, F = m >> 1, rb = m & 1, f = F & 7;
                        if constexpr ((m & 3) == 0) wait_kt(ic_<F>{});          // first use of the batch (F, F + 1)
                        mfma_acc<A_ACC + 16 * (4 * rb + (f & 3))>(ktf[F % 4], dsf[f >> 2][rb]);
                    }
                    // ---- LDS reads, in BATCHES behind the MFMA that first uses the previous batch: the compiler waits with lgkmcnt(0) at
                    // every first use of a loaded fragment, i.e. a read issued just before such a point is waited for at full latency
                    // (first version of this kernel: reads spread one per step -> a full LDS round trip every third step).  K / V steps
                    // g + 2, g + 3 behind MFMA 4 g (g even: 8 MFMAs ahead of their use); K^T fragments F + 2, F + 3 behind MFMA 64 + 2 F
                    // (F even: 4 MFMAs ahead), the first two behind MFMA 56
                    if constexpr (i >= 8 && i <= 48 && (i & 7) == 0) { load_kv(ic_<(i >> 2) + 2>{}); load_kv(ic_<(i >> 2) + 3>{}); }
                    if constexpr (i == 56) { load_kt(ic_<0>{}); load_kt(ic_<1>{}); }
                    if constexpr (i >= 64 && i <= 88 && (i & 3) == 0) { load_kt(ic_<(i - 64) / 2 + 2>{}); load_kt(ic_<(i - 64) / 2 + 3>{}); }
                    // ---- softmax backward.  An element is touched only when its last MFMA is >= 2 MFMAs back (half 0: row block 0
                    // complete after MFMA 29, row block 1 after 31; half 1: after 61 / 63).
                    // half 0: element e = 16 rb + r: stage 1 in slot 32 + e, exp in slot 33 + e, stage 3 in slot 34 + e
                    if constexpr (i >= 34 && i <= 65) { constexpr int e = i - 34; stage3(ic_<0>{}, ic_<(e >> 4)>{}, ic_<(e & 15)>{}, tb[e % 3]); }
                    if constexpr (i >= 33 && i <= 64) { constexpr int e = i - 33; tb[e % 3] = __builtin_amdgcn_exp2f(tb[e % 3]); }
                    if constexpr (i >= 32 && i <= 63) { constexpr int e = i - 32; tb[e % 3] = stage1(ic_<0>{}, ic_<(e >> 4)>{}, ic_<(e & 15)>{}); }
                    if constexpr (i == 42) pack(ic_<0>{}, ic_<0>{}, ic_<0>{});      // row block 0, r < 8: stage 3 done in slot 41
                    if constexpr (i == 50) pack(ic_<0>{}, ic_<1>{}, ic_<0>{});
                    if constexpr (i == 58) pack(ic_<0>{}, ic_<0>{}, ic_<1>{});
                    if constexpr (i == 66) pack(ic_<0>{}, ic_<1>{}, ic_<1>{});      // first used by MFMA 73
                    // half 1: element pair P (rows r0, r0 + 1 of row block rb): P 0-3 rb 0 r < 8 | 4-7 rb 1 r < 8 | 8-11 rb 0 r >= 8 | 12-15 rb 1 r >= 8;
                    // stage 1 in slot 64 + P, exp in slot 65 + P, stage 3 in slot 66 + P
                    if constexpr (i >= 66 && i <= 81) {
                        constexpr int P = i - 66, rb = (P >> 2) & 1, r0 = ((P >> 3) << 3) + ((P & 3) << 1);
                        stage3(ic_<1>{}, ic_<rb>{}, ic_<r0>{}, tc[P % 3][0]);
                        stage3(ic_<1>{}, ic_<rb>{}, ic_<r0 + 1>{}, tc[P % 3][1]);
                    }
                    if constexpr (i >= 65 && i <= 80) {
                        constexpr int P = i - 65;
                        tc[P % 3][0] = __builtin_amdgcn_exp2f(tc[P % 3][0]);
                        tc[P % 3][1] = __builtin_amdgcn_exp2f(tc[P % 3][1]);
                    }
                    if constexpr (i >= 64 && i <= 79) {
                        constexpr int P = i - 64, rb = (P >> 2) & 1, r0 = ((P >> 3) << 3) + ((P & 3) << 1);
                        tc[P % 3][0] = stage1(ic_<1>{}, ic_<rb>{}, ic_<r0>{});
                        tc[P % 3][1] = stage1(ic_<1>{}, ic_<rb>{}, ic_<r0 + 1>{});
                    }
                    // half 0's cp-0 fragments are read until MFMA 71, its cp-1 fragments until MFMA 79: half 1's are packed behind those
                    // (r < 8 of both row blocks: stage 3 done in slot 73; r >= 8: slot 81)
                    if constexpr (i == 74) { pack(ic_<1>{}, ic_<0>{}, ic_<0>{}); pack(ic_<1>{}, ic_<0>{}, ic_<1>{}); }
                    if constexpr (i == 82) { pack(ic_<1>{}, ic_<1>{}, ic_<0>{}); pack(ic_<1>{}, ic_<1>{}, ic_<1>{}); }
#ifdef DQ64_STAMPS
                    if constexpr (i == 31) DQ64_STAMP(st_a);
                    if constexpr (i == 63) DQ64_STAMP(st_b);
                    if constexpr (i == 79) DQ64_STAMP(st_c);
                    if constexpr (i == 95) DQ64_STAMP(st_d);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            if (need_mask) tile_body(std::true_type{});
            else tile_body(std::false_type{});
#ifdef DQ64_STAMPS
            st_plain = !need_mask;
            n_masked += need_mask ? 1 : 0;
        } else {
            n_skipped += 1;
#endif
        }
        wait_two_stages_in_flight();       // tile t + 1 has landed (this wave's pieces; the barrier makes it everyone's)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this tile's LDS reads and the bias row's ds_write
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef DQ64_STAMPS
        DQ64_STAMP(st_e);
        if (st_plain) {
            sum_a += st_a - st_t0; sum_b += st_b - st_a; sum_c += st_c - st_b; sum_d += st_d - st_c; sum_e += st_e - st_d;
            n_plain += 1;
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the ring's last pieces must not outlive the workgroup's LDS
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");      // the last MFMAs' results -> v_accvgpr_read
    static_for<0, 2>([&](auto rbc) {
        constexpr int rb = decltype(rbc)::value;
        bf16_t* op = dQ + ((long)b * L + qc[rb]) * lddq + (long)h * HD;
        static_for<0, C::NDB>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            static_for<0, 4>([&](auto gc) {
                constexpr int g4 = decltype(gc)::value, R = A_ACC + 16 * (4 * rb + d) + 4 * g4;
                const int dd = d * 32 + 8 * g4 + 4 * hh;
                u32x2 o;
                o[0] = pack_bf2(agpr_read<R>() * scale, agpr_read<R + 1>() * scale);
                o[1] = pack_bf2(agpr_read<R + 2>() * scale, agpr_read<R + 3>() * scale);
                if (q[rb] < L) *reinterpret_cast<u32x2*>(op + dd) = o;
            });
        });
    });
#ifdef DQ64_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        unsigned long long* o = g_dq64_stamps + (size_t)blockIdx.x * 12;
        unsigned long long st_exit;
        DQ64_STAMP(st_exit);
        o[0] = st_loop - st_entry; o[1] = st_exit - st_entry; o[2] = n_plain; o[3] = n_masked; o[4] = n_skipped;
        o[5] = sum_a; o[6] = sum_b; o[7] = sum_c; o[8] = sum_d; o[9] = sum_e; o[10] = (unsigned long long)(ntiles - t_first); o[11] = (unsigned long long)qb;
    }
#endif
}

// (no key-padding mask: see the kernel)
int mantis_attn_dq64_launch(bool causal, int B, hipStream_t s, const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO,
                            const float* LSE, float* Dsum, bf16_t* dQ, int L, int Lk, int H, int Hkv, long ldq, long ldk,
                            long ldv, long ldo, long lddq, float scale, const bf16_t* Ofwd, long ldout, const int* kstart) {
    const dim3 grid(cdiv(L, 256) * H * B);
    if (causal)
        MANTIS_LAUNCH((attn_bwd_dq64_kernel<true>), grid, dim3(256), 0, s, Q, K, V, dO, LSE, Dsum, dQ, L, Lk, H, Hkv, ldq, ldk, ldv, ldo,
                      lddq, scale, Ofwd, ldout, kstart);
    else
        MANTIS_LAUNCH((attn_bwd_dq64_kernel<false>), grid, dim3(256), 0, s, Q, K, V, dO, LSE, Dsum, dQ, L, Lk, H, Hkv, ldq, ldk, ldv, ldo,
                      lddq, scale, Ofwd, ldout, kstart);
    return mantis_check_launch();
}

#ifdef DQ64_STAMPS
// probe builds only: copy the stamps of the last launch (n workgroups x 12 u64) to host memory
extern "C" int mantis_probe_dq64_stamps(void* host_dst, int n_wg) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_dq64_stamps), (size_t)n_wg * 96, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif
