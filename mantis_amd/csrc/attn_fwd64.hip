// Attention forward for hd = 128 with 64 query rows per wave (gfx950) -- the software-pipelined sibling of attn_fwd_kernel<128>
// (attn.hip), same entry point (mantis_attn_fwd / mantis_attn_fwd_cross), same masks, same output and LSE definition.
//
// Replaces (reference path): the LlamaAttention / Qwen2 attention call of the decoder forward,
//   transformers/models/llama/modeling_llama.py:191-214,262-276, reached from /root/reference/mantis/models/mllava/modeling_llava.py:456,510.
//
// Why a second kernel.  The 32-rows-per-wave forward issues ~7 non-MFMA instructions per MFMA where ~5 fit an MFMA's 32-cycle shadow,
// and leaves the overlap of one wave's softmax with the other wave's MFMAs to the arbiter of two waves per SIMD (round-2 PMC: MFMA pipe
// 27 % busy).  Here a workgroup is 4 waves = ONE per SIMD with the whole 512-register file, each wave owns 64 query rows as two 32-row
// blocks A and B, and ONE instruction stream interleaves the blocks by hand:
//   * every K fragment (ds_read_b128) and every V^T fragment (two transposing reads) feeds TWO MFMAs;
//   * the accumulator file holds what only MFMAs touch -- O (128 registers), the Q fragments (64), the K fragments (64, read from LDS
//     straight into AGPRs) -- and the arch VGPRs hold the scores, V^T fragments, P fragments and the softmax;
//   * per KV tile of 64 keys the stream is four phases of 16 MFMAs
//         Ph1  S_A(t)   = K(t).Q_A^T            Ph2  O_B += V(t-1)^T.P_B(t-1)
//         Ph3  S_B(t)   = K(t).Q_B^T            Ph4  O_A += V(t)^T.P_A(t)
//     so that each block's softmax has two phases of the OTHER block's MFMAs to hide in (A: Ph2-Ph3, B: Ph4-Ph1 of the next tile); the
//     fillers are dealt out per MFMA gap: one score element (fma, exp2, add, half a cvt_pk, half a max3) + the LDS fragment reads + the
//     LDS-DMA issues.  Measured (tools/mfma_filler_probe.hip): a gap hides ~28 cycles of issue -- VALU 4, v_exp 8, LDS read 8, DMA
//     piece ~54 -- so at hd 128 this loop is ISSUE-bound (~2500 cycles of issue per 2048 cycles of MFMA), not MFMA-bound;
//   * no max pass: the exps run against the row's CURRENT reference max m; the tile's largest exponent is collected on the side and
//     checked once per block and tile; only when a row exceeds m by more than 8 (exp2 domain: P <= 256, exact in fp32 / bf16 relative
//     precision) is the tile redone against a raised m and O / l rescaled (the scores stay intact for that).  Rare -- the first live
//     tile of a row, then almost never -- which it has to be: rescaling an AGPR-resident O costs 3 instructions per element;
//   * masks (key padding, causal diagonal, packed-sample start) are a pre-pass over the scores on the tiles that need one, so the
//     common stream is branch-free arithmetic; tiles with no live key for the wave (beyond its diagonal, all padding, before its
//     sample) skip the compute and only keep the workgroup's barrier / DMA protocol.
// LDS: K ring 4 x 16 KiB + V ring 4 x 16 KiB, filled by buffer_load ... lds through a per-tile descriptor (rows past Lk read as zeros),
// one barrier per tile: barrier(t) sits before Ph2(t); V(t) and K(t+1) must have landed before it (a COUNTED vmcnt: the two younger
// tiles' 16 pieces stay in flight), V(t+3) and K(t+4) are issued after it -- one workgroup per CU has nobody to hide an HBM round trip
// behind, so the ring runs 2.3 tiles (~5000 cycles) ahead.  Plus 8 KiB of per-tile key-liveness words (ballots of the padding mask),
// built once in the prologue so that the tile loop has no global load of its own.
#include "attn_common.h"

namespace {

constexpr int HD = 128;
constexpr int TILE = 64 * 256;                 // 64 keys x 256 B
constexpr int NS = 4;                          // ring slots per operand: a tile's DMA is issued NS - 2 (+ a bit) tiles before its barrier
constexpr int OFF_K = 0, OFF_V = NS * TILE;    // K ring, V ring
constexpr int OFF_BITS = 2 * NS * TILE;        // per-tile key-liveness words (8 B each)
constexpr int MAX_TILES = 1024;                // of one workgroup's key range (65536 keys); longer ranges take attn_fwd_kernel
constexpr float LAZY_TH = 8.0f;                // exp2-domain slack of the running max

typedef __attribute__((address_space(3))) void lds_void;

// ---- the accumulator file is owned by the asm statements below, by NUMBER (the compiler allocates no AGPR in this kernel: audit the
// build for v_accvgpr_* / scratch_* outside ASMSTART..ASMEND -- tools/attn_fwd64_audit.py):
//     a[  0: 63]  O^T of block A, d-block d at 16 d          a[128:159]  Q fragments of block A, hd chunk ks at 4 ks
//     a[ 64:127]  O^T of block B                             a[160:191]  Q fragments of block B
//     a[192:255]  K fragments of the current tile, fragment i = (key block) * 8 + (hd chunk) at 4 i
constexpr int AO = 0, AQ = 128, AK = 192;
// ... and so are the arch VGPRs v[192:255]: the 16 V^T fragments (fragment j = (key chunk of 16) * 4 + (d-block) at 192 + 4 j).  Left to
// the register allocator, the two 64-bit halves of a fragment (two transposing reads) become separate values that are copied into a
// 128-bit tuple -- or spilled to the accumulator file the moment they are "defined", before the LDS data has arrived.  Every statement
// that touches them lists all 64 as clobbers, so the compiler keeps nothing of its own there inside the loop.
constexpr int VF = 192;
#define VF_ALL                                                                                                                    \
    "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207",   \
    "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223",   \
    "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239",   \
    "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define AGPR_ALL                                                                                                                  \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19",  \
    "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37",       \
    "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55",       \
    "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73",       \
    "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91",       \
    "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108",    \
    "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124",    \
    "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140",    \
    "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156",    \
    "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172",    \
    "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188",    \
    "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204",    \
    "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220",    \
    "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236",    \
    "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252",    \
    "a253", "a254", "a255"

// Timing probes (wrong results; tools/build_probe_lib.sh): FWD64_PROBE_TIMING (s_memtime stamps into the LSE buffer), FWD64_PROBE_NO_VMWAIT (no wait for the ring), FWD64_PROBE_NO_EXP (no
// exp / sum / pack fillers), FWD64_PROBE_NO_DMA (no ring refill in the loop), FWD64_PROBE_NO_LDS (no fragment reads), FWD64_PROBE_NO_MAX, FWD64_PROBE_NO_BARRIER
// ---- instruction helpers (the compiler counts and schedules none of these; see the wait / hazard notes at the call sites)
template <int KF, int QF>     // scores, C = 0: S^T(key block) = K fragment KF . Q fragment QF
__device__ __forceinline__ void mfma_s0(f32x16& d) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], a[%3:%4], 0" : "=v"(d) : "n"(AK + 4 * KF), "n"(AK + 4 * KF + 3), "n"(AQ + 4 * QF), "n"(AQ + 4 * QF + 3));
}
template <int KF, int QF>
__device__ __forceinline__ void mfma_s(f32x16& d) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], a[%3:%4], %0" : "+v"(d) : "n"(AK + 4 * KF), "n"(AK + 4 * KF + 3), "n"(AQ + 4 * QF), "n"(AQ + 4 * QF + 3));
}
template <int OB, int J>      // O^T(d-block at a[OB : OB + 15]) += V^T fragment J . P fragment
__device__ __forceinline__ void mfma_o(const u32x4& p) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], v[%2:%3], %4, a[%0:%1]"
                 :
                 : "n"(OB), "n"(OB + 15), "n"(VF + 4 * J), "n"(VF + 4 * J + 3), "v"(p)
                 : VF_ALL);
}
template <int KF, int OFF>
__device__ __forceinline__ void lds_read_k(unsigned addr) {                                 // K fragment KF -> its AGPRs
#ifdef FWD64_PROBE_NO_LDS
    return;
#endif
    asm volatile("ds_read_b128 a[%0:%1], %2 offset:%3" : : "n"(AK + 4 * KF), "n"(AK + 4 * KF + 3), "v"(addr), "i"(OFF));
}
template <int J, int OFF>
__device__ __forceinline__ void lds_read_vt(unsigned addr, unsigned addr8) {                // V^T fragment J: two transposing reads
#ifdef FWD64_PROBE_NO_LDS
    return;
#endif
    asm volatile("ds_read_b64_tr_b16 v[%0:%1], %4 offset:%6\n\tds_read_b64_tr_b16 v[%2:%3], %5 offset:%6"
                 :
                 : "n"(VF + 4 * J), "n"(VF + 4 * J + 1), "n"(VF + 4 * J + 2), "n"(VF + 4 * J + 3), "v"(addr), "v"(addr8), "i"(OFF)
                 : VF_ALL);
}
template <int R>
__device__ __forceinline__ void vf_zero4() {
    asm volatile("v_mov_b32 v[%0], 0\n\tv_mov_b32 v[%1], 0\n\tv_mov_b32 v[%2], 0\n\tv_mov_b32 v[%3], 0"
                 :
                 : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3)
                 : VF_ALL);
}
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
template <int R>              // a[R .. R+3] *= f
__device__ __forceinline__ void agpr_scale4(float f) {
    float t0, t1, t2, t3;
    asm volatile(
        "v_accvgpr_read_b32 %0, a[%5]\n\tv_accvgpr_read_b32 %1, a[%6]\n\tv_accvgpr_read_b32 %2, a[%7]\n\tv_accvgpr_read_b32 %3, a[%8]\n\t"
        "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4\n\t"
        "v_accvgpr_write_b32 a[%5], %0\n\tv_accvgpr_write_b32 a[%6], %1\n\tv_accvgpr_write_b32 a[%7], %2\n\tv_accvgpr_write_b32 a[%8], %3"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(f), "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
}
__device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float max2(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// max over the lane pair (l, l ^ 32): the two lanes that share a query row
__device__ __forceinline__ float pair_max(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));       // VALU write -> permlane read -> VALU read
    return max2(a, b);
}

#define WIN_P(sm, I) (((I) % 3) == 0 ? (sm).pw0 : ((I) % 3) == 1 ? (sm).pw1 : (sm).pw2)
#define WIN_T(sm, I) (((I) % 3) == 0 ? (sm).tw0 : ((I) % 3) == 1 ? (sm).tw1 : (sm).tw2)

struct Softmax {          // per 32-row block, per lane (one query row per lane pair)
    float m;              // reference max of the exponent (exp2 domain): P = exp2(S c - m); -inf until the row's first live key
    float negm;           // -m, 0 while m = -inf
    float th;             // a tile whose largest exponent S c - m exceeds th raises m: LAZY_TH, -inf while m = -inf (any live key sets m)
    float l;              // this lane's share of the running sum
    // P and exponent of the last three elements (the sum, the bf16 pair and the max trail the exp by one element).  Scalars, and no two
    // P words adjacent: an indexed member, or a pair that can be fetched as one <2 x float>, keeps the struct in memory (which the
    // backend then "promotes" to LDS)
    float pw0;
    float l0;             // l before the current tile (for the fix-up)
    float pw1;
    float tm;             // largest exponent of the current tile so far
    float pw2;
    float tw0, tw1, tw2;
};

}  // namespace

// one 64-row block of one (batch, head)   (outside the anonymous namespace: profilers then print the kernel's plain name)
template <bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_fwd64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                            const bf16_t* __restrict__ V, const int* __restrict__ kmask,
                                                            bf16_t* __restrict__ O, float* __restrict__ LSE, int L, int Lk, int H,
                                                            int Hkv, long ldq, long ldk, long ldv, long ldo, float scale,
                                                            const int* __restrict__ kstart) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * NS * TILE + MAX_TILES * 8];

#ifdef FWD64_PROBE_TIMING
    const unsigned long long T0 = __builtin_amdgcn_s_memtime();
#endif
    const int lane = threadIdx.x & 63, hh = lane >> 5, lq = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gx = (L + 255) >> 8;
    int bx, h, b;
    xcd_tile_map(gx, H, bx, h, b, CAUSAL);
    const int hk = h / (H / Hkv);
    const int qbi = CAUSAL ? (gx - 1 - bx) : bx;   // causal: longest rows first
    const int qblk0 = qbi * 256, q0w = qblk0 + wave * 64;
    const float c = scale * LOG2E;

    // ---- prologue order: the key range, the first DMA pieces (K(f), V(f), K(f+1)), the Q rows and the key mask words all in flight
    // together, the rest of the ring behind them, register initialisation under the round trip
    asm volatile("" ::: AGPR_ALL);      // (declares the whole accumulator file as used)
    const int t_first = __builtin_amdgcn_readfirstlane(kstart ? (kstart[(long)b * L + (qblk0 < L ? qblk0 : L - 1)] >> 6) : 0);
    const int kend = CAUSAL ? (qblk0 + 256 < Lk ? qblk0 + 256 : Lk) : Lk;
    const int ntiles = (kend + 63) >> 6;

    // ---- LDS fragment addresses (tile-relative; ring slot, key block and key chunk go into the instruction's immediate offset)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
    unsigned kaddr[8], vaddr[4], vaddr8[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kaddr[ks] = lds0 + OFF_K + Lay<HD>::chunk_off(lq, ks * 2 + hh);
    {
        const int s = lane & 15, g16 = (lane >> 4) & 1;
        const int row = 4 * hh + (s >> 2);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int col = d * 32 + 16 * g16 + (s & 3) * 4;
            const int off = Lay<HD>::chunk_off(row, col >> 3) + (col & 7) * 2;
            vaddr[d] = lds0 + OFF_V + off;
            vaddr8[d] = lds0 + OFF_V + ((off + 8 * 256) ^ 32);
        }
    }
    // ---- LDS-DMA geometry: a 64-row tile is 16 pieces of 4 rows x 256 B; this wave moves pieces 4w .. 4w+3 of K and of V
    const bf16_t* Kb = K + (long)b * Lk * ldk + (long)hk * HD;
    const bf16_t* Vb = V + (long)b * Lk * ldv + (long)hk * HD;
    unsigned voK[4], voV[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ lds_swz(row);
        voK[j] = (unsigned)(((long)row * ldk + ch * 8) * 2);
        voV[j] = (unsigned)(((long)row * ldv + ch * 8) * 2);
    }
    // descriptor of tile t of K (or V): rows >= Lk (and whole tiles >= ntiles) read as zeros through the descriptor's range
    auto tile_desc = [&](const bf16_t* base, int ld, int t) __attribute__((always_inline)) {
        const int key0 = t * 64;
        int rows = Lk - key0;
        rows = rows > 64 ? 64 : rows;
        rows = t < ntiles ? rows : 0;
        rows = rows < 0 ? 0 : rows;
        int nrec = ((rows - 1) * ld + HD) * 2;
        nrec = rows > 0 ? nrec : 0;
        nrec = __builtin_amdgcn_readfirstlane(nrec);      // the clamp may be selected as v_med3: without this the descriptor lives in VGPRs and every DMA piece becomes a waterfall loop
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)key0 * ld), 0, nrec, 0x00020000);
    };
    // piece J of this wave's four -> ring slot `slot`
    auto dma_piece = [&](const __amdgpu_buffer_rsrc_t rs, const unsigned (&vo)[4], int ring_off, int slot, auto jsel)
                         __attribute__((always_inline)) {
        constexpr int J = decltype(jsel)::value;
#ifdef FWD64_PROBE_NO_DMA
        return;
#endif
        lds_void* d = (lds_void*)(smem + ring_off + slot * TILE + (wave * 4 + J) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d, 16, vo[J], 0, 0, 0);
    };
    // ---- K(f), V(f), K(f+1) first: what barrier(f) and the first K fragment reads wait for
    {
        const __amdgpu_buffer_rsrc_t rk0 = tile_desc(Kb, (int)ldk, t_first), rv0 = tile_desc(Vb, (int)ldv, t_first),
                                     rk1 = tile_desc(Kb, (int)ldk, t_first + 1);
        static_for<0, 4>([&](auto j) __attribute__((always_inline)) { dma_piece(rk0, voK, OFF_K, 0, j); });
        static_for<0, 4>([&](auto j) __attribute__((always_inline)) { dma_piece(rv0, voV, OFF_V, 0, j); });
        static_for<0, 4>([&](auto j) __attribute__((always_inline)) { dma_piece(rk1, voK, OFF_K, 1, j); });
    }
    // ---- this wave's rows: Q loads (the fragments go to a[AQ ..] further down)
    int q[2], ks_q[2];
    u32x4 qreg[2][8];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        q[blk] = q0w + blk * 32 + lq;
        const int qc = q[blk] < L ? q[blk] : L - 1;
        ks_q[blk] = kstart ? kstart[(long)b * L + qc] : 0;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            qreg[blk][ks] = *reinterpret_cast<const u32x4*>(Q + ((long)b * L + qc) * ldq + (long)h * HD + (ks * 2 + hh) * 8);
    }
    // key-liveness word of tile t (padding mask and the Lk tail): wave w ballots tiles f + w, f + w + 4, ... into LDS, eight loads in
    // flight at a time
    {
        unsigned long long* tb = reinterpret_cast<unsigned long long*>(smem + OFF_BITS);
        for (int t0 = t_first + wave; t0 < ntiles; t0 += 32) {
            int ok[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int key = (t0 + 4 * i) * 64 + lane;
                ok[i] = key < Lk;
                if (kmask != nullptr && ok[i]) ok[i] = kmask[(long)b * Lk + key] != 0;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned long long w = __ballot(ok[i]);
                if (lane == 0 && t0 + 4 * i < ntiles) tb[t0 + 4 * i - t_first] = w;
            }
        }
    }
    const unsigned bits_addr = lds0 + OFF_BITS;
    // asm: the compiler must see no LDS read in the loop.  The word of tile t+1 is requested right behind barrier(t) and collected at the
    // top of the next step, behind the lgkmcnt(0) that closes every step.
    u32x2 bits_w;
    auto request_bits = [&](int t) __attribute__((always_inline)) {
        asm volatile("ds_read_b64 %0, %1" : "=v"(bits_w) : "v"(bits_addr + (unsigned)(t - t_first) * 8u));
    };
    auto collect_bits = [&]() __attribute__((always_inline)) -> unsigned long long {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bits_w));
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)bits_w[1]) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)bits_w[0]);
    };

    // ---- accumulators and softmax state
    f32x16 sA[2], sB[2];       // S^T of the current tile: key block x (key row, query column)
    u32x4 pA[4], pB[4];        // P fragments (bf16 pairs): 16 keys each
    static_for<0, 128>([&](auto r) __attribute__((always_inline)) { agpr_write<AO + decltype(r)::value>(0u); });
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int e = 0; e < 16; ++e) { sA[sb][e] = -INFINITY; sB[sb][e] = -INFINITY; }      // the "previous tile" of the first one: P = 0
#pragma unroll
    for (int g = 0; g < 4; ++g) { pA[g] = u32x4{0u, 0u, 0u, 0u}; pB[g] = u32x4{0u, 0u, 0u, 0u}; }
    static_for<0, 16>([&](auto j) __attribute__((always_inline)) { vf_zero4<VF + 4 * decltype(j)::value>(); });      // V^T fragments
    Softmax smA, smB;
    smA.m = -INFINITY, smA.negm = 0.f, smA.th = -INFINITY, smA.l = 0.f, smA.l0 = 0.f, smA.tm = -INFINITY;
    smA.pw0 = smA.pw1 = smA.pw2 = 0.f, smA.tw0 = smA.tw1 = smA.tw2 = -INFINITY;
    smB.m = -INFINITY, smB.negm = 0.f, smB.th = -INFINITY, smB.l = 0.f, smB.l0 = 0.f, smB.tm = -INFINITY;
    smB.pw0 = smB.pw1 = smB.pw2 = 0.f, smB.tw0 = smB.tw1 = smB.tw2 = -INFINITY;

    // ---- softmax pieces
    // Score element E of a block: P = exp2(S c - m) against the row's CURRENT reference max -- no max pass in front of the exps.  The
    // tile's largest exponent is collected on the side (one v_max3 per two elements) and checked once, after element 31: only if some
    // row's exceeds th is the tile redone against a raised m (softmax_fixup; S is left intact for that -- P goes to a three-element
    // window and the bf16 fragment).  Rare: the first live tile of a row, then whenever a row's max grows by more than LAZY_TH.
    // The add into the running sum, the bf16 pair and the max trail the exp by one element: nothing right behind a v_exp depends on it.
    auto exp_elem = [&](auto ec, f32x16 (&s)[2], Softmax& sm, u32x4 (&p)[4]) __attribute__((always_inline)) {
        constexpr int E = decltype(ec)::value;
#ifdef FWD64_PROBE_NO_EXP
        return;
#endif
        if constexpr (E == 0) {
            sm.l0 = sm.l;
            sm.tm = -INFINITY;
        }
        if constexpr (E > 0) sm.l += WIN_P(sm, (E - 1));
        if constexpr (E >= 2 && (E & 1) == 0) {
            p[(E - 2) >> 3][((E - 2) & 7) >> 1] = pack_bf2(WIN_P(sm, (E - 2)), WIN_P(sm, (E - 1)));
            sm.tm = max3(sm.tm, WIN_T(sm, (E - 2)), WIN_T(sm, (E - 1)));
            asm volatile("" : "+v"(p[(E - 2) >> 3]), "+v"(sm.tm));       // pins: the code sinker would move all of this to the consumer
        }
        WIN_T(sm, E) = __builtin_fmaf(s[E >> 4][E & 15], c, sm.negm);
        WIN_P(sm, E) = __builtin_amdgcn_exp2f(WIN_T(sm, E));
        asm volatile("" : "+v"(WIN_P(sm, E)), "+v"(sm.l));
    };
    // the tile against a raised reference max (wave-uniform branch, rare): P, the fragments and the tile's sum again from S; l and O
    // (which does not hold this tile yet) rescaled
    auto softmax_fixup = [&](f32x16 (&s)[2], Softmax& sm, u32x4 (&p)[4], auto obc) __attribute__((always_inline)) {
        constexpr int OB = decltype(obc)::value;
        const float tmax = pair_max(sm.tm);
        const bool upd = tmax > sm.th;
        const float m_new = upd ? tmax - sm.negm : sm.m;
        const float alpha = upd ? __builtin_amdgcn_exp2f(sm.m - m_new) : 1.f;      // m = -inf: 0 (l0 and O are still 0)
        sm.m = m_new;
        sm.negm = m_new == -INFINITY ? 0.f : -m_new;
        sm.th = upd ? LAZY_TH : sm.th;
        float lsum = 0.f;
#pragma unroll
        for (int E = 0; E < 32; E += 2) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[E >> 4][E & 15], c, sm.negm));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[E >> 4][(E & 15) + 1], c, sm.negm));
            lsum += p0 + p1;
            p[E >> 3][(E & 7) >> 1] = pack_bf2(p0, p1);
        }
        sm.l = sm.l0 * alpha + lsum;
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");       // this O's last MFMA is at least a phase back; belt and braces
        static_for<0, 16>([&](auto r) __attribute__((always_inline)) { agpr_scale4<OB + 4 * decltype(r)::value>(alpha); });
        asm volatile("s_nop 2" ::: "memory");                  // accvgpr_write -> MFMA src C
    };
    // element 31's trailing work, then the check
    auto exp_finish = [&](f32x16 (&s)[2], Softmax& sm, u32x4 (&p)[4], auto obc) __attribute__((always_inline)) {
#ifdef FWD64_PROBE_NO_EXP
        return;
#endif
        sm.l += WIN_P(sm, 31);
        p[3][3] = pack_bf2(WIN_P(sm, 30), WIN_P(sm, 31));
        sm.tm = max3(sm.tm, WIN_T(sm, 30), WIN_T(sm, 31));
        asm volatile("" : "+v"(p[3]), "+v"(sm.l));
        if (__builtin_expect(__any(sm.tm > sm.th), 0)) softmax_fixup(s, sm, p, obc);
    };
    // mask pre-pass: dead scores -> -inf (key padding bits, causal, sample start); keys of element (sb, r): sb*32 + (r&3) + 8*(r>>2) + 4*hh
    auto mask_scores = [&](f32x16 (&s)[2], int blk, int key0, unsigned long long bits) __attribute__((always_inline)) {
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");      // the scores' last MFMA may be only one gap back
        // everything lane- or tile-dependent is folded into three per-tile values so that the per-element tests compare against
        // immediates (nothing loop-invariant for the compiler to hoist into registers)
        const unsigned long long lb = bits >> (4 * hh);
        const unsigned w[2] = {(unsigned)lb, (unsigned)(lb >> 32)};
        const int dq = CAUSAL ? q[blk] - key0 - 4 * hh : 64;     // live iff kl <= dq
        const int dk = ks_q[blk] - key0 - 4 * hh;                // live iff kl >= dk
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kb = (r & 3) + 8 * (r >> 2), kl = sb * 32 + kb;
                const bool live = ((w[sb] >> kb) & 1u) && kl <= dq && kl >= dk;
                s[sb][r] = live ? s[sb][r] : -INFINITY;
            }
    };

    // ---- the rest of the ring: V(f+1) K(f+2) | V(f+2) K(f+3) -- what the steady-state count expects to find in flight
    static_for<2, NS>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const __amdgpu_buffer_rsrc_t rv = tile_desc(Vb, (int)ldv, t_first + i - 1), rk = tile_desc(Kb, (int)ldk, t_first + i);
        static_for<0, 4>([&](auto j) __attribute__((always_inline)) { dma_piece(rv, voV, OFF_V, i - 1, j); });
        static_for<0, 4>([&](auto j) __attribute__((always_inline)) { dma_piece(rk, voK, OFF_K, i, j); });
    });
    asm volatile("" : "+v"(qreg[0][0]), "+v"(qreg[0][1]), "+v"(qreg[0][2]), "+v"(qreg[0][3]), "+v"(qreg[0][4]), "+v"(qreg[0][5]),
                 "+v"(qreg[0][6]), "+v"(qreg[0][7]), "+v"(qreg[1][0]), "+v"(qreg[1][1]), "+v"(qreg[1][2]), "+v"(qreg[1][3]),
                 "+v"(qreg[1][4]), "+v"(qreg[1][5]), "+v"(qreg[1][6]), "+v"(qreg[1][7]));
    static_for<0, 16>([&](auto ic) __attribute__((always_inline)) {
        constexpr int blk = decltype(ic)::value >> 3, ks = decltype(ic)::value & 7;
        agpr_write<AQ + 32 * blk + 4 * ks + 0>(qreg[blk][ks][0]);
        agpr_write<AQ + 32 * blk + 4 * ks + 1>(qreg[blk][ks][1]);
        agpr_write<AQ + 32 * blk + 4 * ks + 2>(qreg[blk][ks][2]);
        agpr_write<AQ + 32 * blk + 4 * ks + 3>(qreg[blk][ks][3]);
    });
    // wave-uniform sample-start bounds: no key below ks_min is live for any row, keys below ks_max[blk] need the pre-pass
    int ks_min = 0, ks_max[2] = {0, 0};
    if (kstart) {
        ks_min = -(int)wave_max((float)-(ks_q[0] < ks_q[1] ? ks_q[0] : ks_q[1]));     // exact below 2^24
        ks_max[0] = (int)wave_max((float)ks_q[0]);
        ks_max[1] = (int)wave_max((float)ks_q[1]);
        ks_min = __builtin_amdgcn_readfirstlane(ks_min);
        ks_max[0] = __builtin_amdgcn_readfirstlane(ks_max[0]);
        ks_max[1] = __builtin_amdgcn_readfirstlane(ks_max[1]);
    }
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");      // K(f), V(f), K(f+1); the liveness words (ds_write)
    __builtin_amdgcn_s_barrier();      // raw: __syncthreads() is a fence and would drain the whole ring (vmcnt(0))
    request_bits(t_first);
    static_for<0, 16>([&](auto i) __attribute__((always_inline)) { lds_read_k<i.value, (i.value >> 3) * 32 * 256>(kaddr[i.value & 7]); });
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 2" ::: "memory");      // also: accvgpr_write (Q fragments, O = 0) -> MFMA operand

#ifdef FWD64_PROBE_TIMING
    const unsigned long long T1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- one tile, ring slot SLOT = (t - t_first) & 1.  Gap n = MFMA n of the tile plus its fillers.
    auto tile = [&](auto slotc, int t, bool active, bool maskA, bool maskB, unsigned long long bits_t) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slotc)::value;
        constexpr int VS = SLOT * TILE;                                         // immediate offset of this tile's V slot
        constexpr int KN = ((SLOT + 1) % NS) * TILE;                            // next tile's K slot
        constexpr int VD = (SLOT + NS - 1) % NS, KD = SLOT;                     // slots the DMA of V(t+NS-1) / K(t+NS) goes to
        const int key0 = t * 64;
        if (active) {
            // Ph1: S_A(t) | B elements 14 .. 31 of the previous tile, its check
            static_for<0, 16>([&](auto i) __attribute__((always_inline)) {
                constexpr int I = i.value, ks = I >> 1, sb = I & 1;
                if constexpr (ks == 0) mfma_s0<sb * 8 + ks, ks>(sA[sb]);
                else mfma_s<sb * 8 + ks, ks>(sA[sb]);
                constexpr int e0 = 14 + (I * 18) / 16, e1 = 14 + ((I + 1) * 18) / 16;
                static_for<e0, e1>([&](auto e) __attribute__((always_inline)) { exp_elem(e, sB, smB, pB); });
                if constexpr (I == 15) exp_finish(sB, smB, pB, std::integral_constant<int, AO + 64>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // barrier(t): V(t) and K(t+1) have landed for every wave (the 16 younger pieces may still fly); everybody is done with V(t-1)'s
        // slot and -- the K fragments of tile t were read a tile ago -- with K(t)'s
#ifndef FWD64_PROBE_NO_VMWAIT
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#endif
#ifndef FWD64_PROBE_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        request_bits(t + 1);
        if (active) {
            // Ph2: O_B += V(t-1)^T.P_B(t-1) | V^T fragments of tile t (fragment j is free once MFMA j has issued) | A elements 0 .. 13
            static_for<0, 16>([&](auto j) __attribute__((always_inline)) {
                constexpr int J = j.value;
                mfma_o<AO + 64 + 16 * (J & 3), J>(pB[J >> 2]);
                lds_read_vt<J, VS + (J >> 2) * 16 * 256>(vaddr[J & 3], vaddr8[J & 3]);
                if constexpr (J == 1) {
                    asm volatile("" : "+v"(sA[0]), "+v"(sA[1]));        // S_A's last MFMA is two gaps back: readable from here on
                    if (__builtin_expect(maskA, 0)) mask_scores(sA, 0, key0, bits_t);
                }
                if constexpr (J >= 2) exp_elem(std::integral_constant<int, J - 2>{}, sA, smA, pA);
                __builtin_amdgcn_sched_barrier(0);
            });
            // Ph3: S_B(t) | the DMA of V(t+NS-1) | A elements 14 .. 31, A's check
            const __amdgpu_buffer_rsrc_t rv = tile_desc(Vb, (int)ldv, t + NS - 1), rk = tile_desc(Kb, (int)ldk, t + NS);
            static_for<0, 16>([&](auto i) __attribute__((always_inline)) {
                constexpr int I = i.value, ks = I >> 1, sb = I & 1;
                if constexpr (ks == 0) mfma_s0<sb * 8 + ks, 8 + ks>(sB[sb]);
                else mfma_s<sb * 8 + ks, 8 + ks>(sB[sb]);
                if constexpr ((I & 3) == 1) dma_piece(rv, voV, OFF_V, VD, std::integral_constant<int, I / 4>{});
                constexpr int e0 = 14 + (I * 18) / 16, e1 = 14 + ((I + 1) * 18) / 16;
                static_for<e0, e1>([&](auto e) __attribute__((always_inline)) { exp_elem(e, sA, smA, pA); });
                if constexpr (I == 15) exp_finish(sA, smA, pA, std::integral_constant<int, AO>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // V^T fragments (issued a phase ago)
            // Ph4: O_A += V(t)^T.P_A(t) | K fragments of tile t+1, then the DMA of K(t+NS) | B elements 0 .. 13
            static_for<0, 16>([&](auto j) __attribute__((always_inline)) {
                constexpr int J = j.value;
                mfma_o<AO + 16 * (J & 3), J>(pA[J >> 2]);
                if constexpr (J < 8) {
                    lds_read_k<2 * J, KN + ((2 * J) >> 3) * 32 * 256>(kaddr[(2 * J) & 7]);
                    lds_read_k<2 * J + 1, KN + ((2 * J + 1) >> 3) * 32 * 256>(kaddr[(2 * J + 1) & 7]);
                }
                if constexpr (J >= 9 && (J & 1) == 1) dma_piece(rk, voK, OFF_K, KD, std::integral_constant<int, (J - 9) / 2>{});
                if constexpr (J == 1) {
                    asm volatile("" : "+v"(sB[0]), "+v"(sB[1]));
                    if (__builtin_expect(maskB, 0)) mask_scores(sB, 1, key0, bits_t);
                }
                if constexpr (J >= 2) exp_elem(std::integral_constant<int, J - 2>{}, sB, smB, pB);
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // K fragments of tile t+1
        } else {
            // this wave has no live key in tile t: keep the ring going
            const __amdgpu_buffer_rsrc_t rv = tile_desc(Vb, (int)ldv, t + NS - 1), rk = tile_desc(Kb, (int)ldk, t + NS);
            static_for<0, 4>([&](auto j) __attribute__((always_inline)) { dma_piece(rv, voV, OFF_V, VD, j); });
            static_for<0, 4>([&](auto j) __attribute__((always_inline)) { dma_piece(rk, voK, OFF_K, KD, j); });
            static_for<0, 16>([&](auto i) __attribute__((always_inline)) { lds_read_k<i.value, KN + (i.value >> 3) * 32 * 256>(kaddr[i.value & 7]); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    auto step = [&](auto slotc, int t) __attribute__((always_inline)) {
        const int key0 = t * 64;
        const unsigned long long bits = collect_bits();
        const bool active = !(CAUSAL && key0 > q0w + 63) && bits != 0ull && key0 + 63 >= ks_min;
        const bool part = bits != ~0ull;
        const bool maskA = part || (CAUSAL && key0 + 63 > q0w) || key0 < ks_max[0];
        const bool maskB = part || (CAUSAL && key0 + 63 > q0w + 32) || key0 < ks_max[1];
        tile(slotc, t, active, maskA, maskB, bits);
    };
    for (int t = t_first; t < ntiles; t += NS) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 >= ntiles) break;
        step(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 >= ntiles) break;
        step(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 >= ntiles) break;
        step(std::integral_constant<int, 3>{}, t + 3);
    }
#ifdef FWD64_PROBE_TIMING
    const unsigned long long T2 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's last (empty) DMA pieces must not outlive the workgroup's LDS
    // ---- drain: the last live tile's block B -- elements 14 .. 31 and the check, then O_B += V^T.P_B
    static_for<14, 32>([&](auto e) __attribute__((always_inline)) { exp_elem(e, sB, smB, pB); });
    exp_finish(sB, smB, pB, std::integral_constant<int, AO + 64>{});
    asm volatile("s_nop 1" : "+v"(pB[0]), "+v"(pB[1]), "+v"(pB[2]), "+v"(pB[3]));
    static_for<0, 16>([&](auto j) __attribute__((always_inline)) { mfma_o<AO + 64 + 16 * (j.value & 3), j.value>(pB[j.value >> 2]); });
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");      // MFMA result -> v_accvgpr_read

    // ---- epilogue: O = O^T / l, LSE = m ln2 + ln l
    auto store_block = [&](auto blkc, const Softmax& sm) __attribute__((always_inline)) {
        constexpr int blk = decltype(blkc)::value;
        const float l_tot = sm.l + __shfl_xor(sm.l, 32, 64);
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
        bf16_t* op = O + ((long)b * L + q[blk]) * ldo + (long)h * HD;
        static_for<0, 16>([&](auto gc) __attribute__((always_inline)) {
            constexpr int d = decltype(gc)::value >> 2, g4 = decltype(gc)::value & 3, R = AO + 64 * blk + 16 * d + 4 * g4;
            const int dd = d * 32 + 8 * g4 + 4 * hh;
            u32x2 w;
            w[0] = pack_bf2(agpr_read<R>() * inv, agpr_read<R + 1>() * inv);
            w[1] = pack_bf2(agpr_read<R + 2>() * inv, agpr_read<R + 3>() * inv);
            if (q[blk] < L) *reinterpret_cast<u32x2*>(op + dd) = w;
        });
        if (q[blk] < L && hh == 0 && LSE) LSE[((long)b * H + h) * L + q[blk]] = l_tot > 0.f ? sm.m * LN2 + logf(l_tot) : INFINITY;
    };
    store_block(std::integral_constant<int, 0>{}, smA);
    store_block(std::integral_constant<int, 1>{}, smB);
#ifdef FWD64_PROBE_TIMING      // per wave: prologue / loop / epilogue cycles and the tile count, over the LSE words of its first rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long T3 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && LSE && q0w + 3 < L) {
        float* w = LSE + ((long)b * H + h) * L + q0w;
        w[0] = (float)(T1 - T0);
        w[1] = (float)(T2 - T1);
        w[2] = (float)(T3 - T2);
        w[3] = (float)(ntiles - t_first);
    }
#endif
}

int mantis_attn_fwd64_launch(bool causal, int B, hipStream_t s, const bf16_t* Q, const bf16_t* K, const bf16_t* V, const int* kmask,
                             bf16_t* O, float* LSE, int L, int Lk, int H, int Hkv, long ldq, long ldk, long ldv, long ldo, float scale,
                             const int* kstart) {
    const dim3 grid(cdiv(L, 256) * H * B);
    if (causal)
        MANTIS_LAUNCH((attn_fwd64_kernel<true>), grid, dim3(256), 0, s, Q, K, V, kmask, O, LSE, L, Lk, H, Hkv, ldq, ldk, ldv, ldo, scale,
                      kstart);
    else
        MANTIS_LAUNCH((attn_fwd64_kernel<false>), grid, dim3(256), 0, s, Q, K, V, kmask, O, LSE, L, Lk, H, Hkv, ldq, ldk, ldv, ldo, scale,
                      kstart);
    return mantis_check_launch();
}
