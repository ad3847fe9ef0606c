// Shared pieces of the attention kernels (attn.hip, attn_fwd64.hip): LDS tile layouts, LDS-DMA staging, transposing fragment
// reads, the XCD-aware workgroup map.
#pragma once
#include "common.h"
#include <type_traits>

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
    union { u32x4 u; bf16x8 b; } c;
    c.u = v;
    return c.b;
}

template <int HD>
struct AttnCfg {
    static constexpr int KP = (HD + 15) / 16 * 16;   // contraction length of Q.K^T, zero padded
    static constexpr int DP = (HD + 31) / 32 * 32;   // output width of P.V, padded to MFMA blocks
    static constexpr int NKS = KP / 16;
    static constexpr int NDB = DP / 32;
    static constexpr int NCK = KP / 8;
    static constexpr int PITCH = DP * 2 + 16;        // row pitch in bytes (rows hold DP columns so tr reads of pad cols stay in-row)
    static constexpr int NCH = DP / 8;               // 16-B chunks staged per row
};

// global -> registers (issue early) and registers -> LDS (write late): a ROWS x NCH-chunk row-major tile, 256 threads.
// rows >= rows_valid and chunks starting at column >= cols_valid are zero.
template <int ROWS, int NCH>
struct TileRegs {
    static constexpr int N = (ROWS * NCH + 255) / 256;
    u32x4 v[N];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ g, long gstride, int rows_valid, int cols_valid) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int idx = threadIdx.x + j * 256;
            const int r = idx / NCH, c = idx - r * NCH;
            v[j] = u32x4{0u, 0u, 0u, 0u};
            if (idx < ROWS * NCH && r < rows_valid && c * 8 < cols_valid)
                v[j] = *reinterpret_cast<const u32x4*>(g + (long)r * gstride + c * 8);
        }
    }
    __device__ __forceinline__ void store(char* lds, int pitch) const {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int idx = threadIdx.x + j * 256;
            const int r = idx / NCH, c = idx - r * NCH;
            if (idx < ROWS * NCH) *reinterpret_cast<u32x4*>(lds + r * pitch + c * 16) = v[j];
        }
    }
};

// LDS tile layouts.  hd == 128: rows are exactly one 256-B bank row, tiles are filled by global_load_lds (fully asynchronous, no
// staging registers) and 16-B chunk c of row r sits at slot c ^ swz(r), swz(r) = ((r & 3) << 2) | ((r >> 2) & 3):
//   * the 16 rows of a ds_read_b128 lane group have 16 distinct r & 15 -> 16 distinct slots = all 64 banks once;
//   * the 4 consecutive rows x 4 consecutive chunks of a ds_read_b64_tr_b16 lane group differ in r & 3 = the slot's upper two
//     bits -> 16 distinct slots as well (the previous c ^ ((r & 7) << 1) was 2-way conflicted for both: rocprofv3
//     SQ_LDS_BANK_CONFLICT = 40-50 % of SQ_LDS_IDX_ACTIVE in all three kernels).
// Other head sizes: padded rows (pitch = row bytes + 16), staged through registers.
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int lds_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int HD>
struct Lay {
    static constexpr bool DMA = (HD == 128);
    static constexpr int PITCH = DMA ? 256 : AttnCfg<HD>::PITCH;
    __device__ static __forceinline__ int chunk_off(int row, int chunk) {
        return row * PITCH + (DMA ? ((chunk ^ lds_swz(row)) << 4) : (chunk << 4));
    }
};

// ROWS x 128 bf16 tile, 256 threads: each wave issues ROWS/16 global_load_lds of 1 KiB (4 rows); rows >= rows_valid are clamped
// (their scores are masked / their outputs never stored).
template <int ROWS>
__device__ __forceinline__ void stage_tile_dma(const bf16_t* __restrict__ g, long gstride, int rows_valid, char* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER = ROWS / 16;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int piece = wave * PER + j;
        int row = piece * 4 + (lane >> 4);
        const int c = (lane & 15) ^ lds_swz(row);
        row = row < rows_valid ? row : rows_valid - 1;
        __builtin_amdgcn_global_load_lds(g + (long)row * gstride + c * 8, (lds_void_t*)(lds + piece * 1024), 16, 0, 0);
    }
}

// A-operand fragment with the contraction index running over ROWS of a row-major LDS tile: lane (i = col0 + (lane & 31), h)
// receives rows {row0 + 4h + 0..3, row0 + 8 + 4h + 0..3} of column i -- two hardware-transposing reads.
template <int HD>
__device__ __forceinline__ bf16x8 read_tr_frag(const char* tile, int row0, int col0, int lane) {
    const int s = lane & 15, g16 = (lane >> 4) & 1, h = lane >> 5;
    const int row = row0 + 4 * h + (s >> 2), col = col0 + 16 * g16 + (s & 3) * 4;
    const int off = Lay<HD>::chunk_off(row, col >> 3) + (col & 7) * 2;
    // row + 8: same r & 3, (r >> 2) & 3 flips its upper bit -> slot ^ 2 -> byte offset ^ 32 (tile bases are 256-B aligned)
    const int off8 = Lay<HD>::DMA ? ((off + 8 * Lay<HD>::PITCH) ^ 32) : off + 8 * Lay<HD>::PITCH;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off8));
    union { s16x4 s2[2]; bf16x8 f; } u;
    u.s2[0] = a;
    u.s2[1] = b;
    return u.f;
}

// The builtin transposing read gets an `s_waitcnt vmcnt(0)` in front of it whenever LDS-DMA pieces are in flight (the compiler cannot tell its
// address from the pieces' destinations): the prefetch of the NEXT K / V (Q / dO) tile is then waited for at the first transposing read of
// the CURRENT tile (profiles/r04_experiments.md 14c).  Hand-issued reads are invisible to that pass; their own completion is waited for by hand.
// Measured (round 5, profiles/r05_experiments.md): on the UNCHANGED instruction order they are 2 - 3 % SLOWER than the builtin (dQ 233 vs 228 us,
// dK/dV 338 vs 328 us: the volatile statements cost the scheduler more than the mid-tile wait costs the DMA), so the dQ kernel keeps the
// builtin; the dK/dV group kernel needs them for its merged MFMA order (ATTN_DKV_MERGED), which requests transposed fragments at the top of a
// step, right behind that step's DMA issue.  Build flags for A/B probe builds: -DATTN_DQ_TR_ASM=1, -DATTN_DKV_TR_ASM=1, -DATTN_DKV_MERGED=0.
#ifndef ATTN_DQ_TR_ASM
#define ATTN_DQ_TR_ASM 0
#endif
#ifndef ATTN_DKV_MERGED
#define ATTN_DKV_MERGED 1
#endif
#ifndef ATTN_DKV_TR_ASM
#define ATTN_DKV_TR_ASM 0
#endif
constexpr bool kDqTrAsm = ATTN_DQ_TR_ASM != 0;
constexpr bool kDkvMerged = ATTN_DKV_MERGED != 0;
constexpr bool kDkvTrAsm = ATTN_DKV_TR_ASM != 0 || kDkvMerged;
#ifndef ATTN_DKV_DMA_IN_SLOTS
#define ATTN_DKV_DMA_IN_SLOTS 1
#endif
constexpr bool kDkvDmaInSlots = ATTN_DKV_DMA_IN_SLOTS != 0 && kDkvMerged;      // the merged step issues its DMA pieces one per MFMA slot
#ifndef ATTN_DKV_ASM_ELEMS
#define ATTN_DKV_ASM_ELEMS 1
#endif
constexpr bool kDkvAsmElems = ATTN_DKV_ASM_ELEMS != 0;      // softmax-backward elements and packs of the pipelined dK/dV step as single asm statements

__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// the four d-block fragments (columns d * 32 ..) of rows row0 .. row0 + 15 of an hd-128 tile: eight transposing reads behind ONE wait
__device__ __forceinline__ void read_tr_frag4_sync(const char* tile, int row0, int lane, bf16x8 (&out)[4]) {
    const int s = lane & 15, g16 = (lane >> 4) & 1, h = lane >> 5;
    const int row = row0 + 4 * h + (s >> 2);
    unsigned a[4], b[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int col = d * 32 + 16 * g16 + (s & 3) * 4;
        const int off = Lay<128>::chunk_off(row, col >> 3) + (col & 7) * 2;
        a[d] = lds_addr_of(tile + off);
        b[d] = lds_addr_of(tile + ((off + 8 * 256) ^ 32));       // row + 8: see read_tr_frag
    }
    union { s16x4 h2[2]; bf16x8 f; } u[4];
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
        "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\tds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(u[0].h2[0]), "=&v"(u[0].h2[1]), "=&v"(u[1].h2[0]), "=&v"(u[1].h2[1]), "=&v"(u[2].h2[0]), "=&v"(u[2].h2[1]), "=&v"(u[3].h2[0]),
          "=&v"(u[3].h2[1])
        : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]));
#pragma unroll
    for (int d = 0; d < 4; ++d) out[d] = u[d].f;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Hand-placed LDS fragment reads for the hd == 128 kernels (the compiler neither counts nor moves them; every use is guarded by an
// explicit counted s_waitcnt lgkmcnt): they sit in the shadow of the MFMAs, 2-3 fragments ahead of their consumer.
template <int OFF>
__device__ __forceinline__ void asm_read_b128(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void asm_read_tr64(s16x4& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
union FragU { s16x4 h[2]; bf16x8 f; };
template <int N> __device__ __forceinline__ void wait_lgkm() {
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
}
// fragments requested ahead of their consumer in the hd-128 forward loop.  Measured on 1x MI355X (round 3, tools/attn_fwd_bench.py, Llama-3
// geometry B = 2, L = 2812, 32/8 x 128): depth 3 191-194 us, 4 201 us, 5 198 us, 6 208 us -- deeper rings cost registers (the kernel sits at
// the 256-VGPR limit of two waves per SIMD: 8-20 B/lane of scratch from depth 5 on) and buy nothing: the loop is not LDS-latency bound
#ifndef ATTN_FRAG_DEPTH
#define ATTN_FRAG_DEPTH 3
#endif
// K-fragment i = sb * 8 + ks of a 64-key tile: row sb * 32 + (lane & 31), chunk ks * 2 + h; kaddr[ks] holds the sb = 0 address
template <int I>
__device__ __forceinline__ void issue_kfrag(bf16x8& dst, const unsigned (&kaddr)[8]) {
    asm_read_b128<(I >> 3) * 32 * 256>(dst, kaddr[I & 7]);
}
// V^T fragment j = (sb * 2 + cp) * 4 + d: rows (sb * 2 + cp) * 16 + ..., d-block d; vaddr / vaddr8 hold the row-block-0 addresses
template <int J>
__device__ __forceinline__ void issue_vfrag(FragU& dst, const unsigned (&vaddr)[4], const unsigned (&vaddr8)[4]) {
    asm_read_tr64<(J >> 2) * 16 * 256>(dst.h[0], vaddr[J & 3]);
    asm_read_tr64<(J >> 2) * 16 * 256>(dst.h[1], vaddr8[J & 3]);
}

__device__ __forceinline__ bf16x8 pack_frag(const f32x16& s, int cp) {
    u32x4 u;
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = pack_bf2(s[8 * cp + 2 * e], s[8 * cp + 2 * e + 1]);
    return as_bf16x8(u);
}

// Workgroup -> (x = tile, y = head, z = batch) for a 1-D launch of gx * H * B workgroups.  Hardware sends workgroup i to XCD i % 8
// (each XCD has a private 4 MiB L2); giving every XCD a CONTIGUOUS range of the (batch, head, tile) order keeps all tiles of a
// head on one XCD, adjacent in time, so the K/V (forward, dQ) or Q/dO (dK/dV) rows they all stream are fetched from HBM once
// instead of once per tile (rocprofv3 FETCH_SIZE of the dK/dV kernel at L = 2812: 2.0 GB per launch with the plain 3-D grid).
// heavy_first (the causal kernels, whose tile x = 0 is the heaviest and x = gx - 1 the lightest): the XCD walks ITS range tile-major
// instead of head-major -- every head's heaviest tile first, then every head's second ... -- which is longest-processing-time-first
// for the XCD's 32 CUs.  Head-major hands the last heads' heavy tiles out when most CUs are already done: list-scheduling the Llama-3
// geometry (8 heads x 22 tiles of 2, 4, .. 44 key tiles per XCD, 64 workgroup slots) ends at 86 key-tile steps head-major and at 70
// tile-major (average load 63); at L = 4096 with 3.5 heads per XCD 98 vs 66.  The set of tiles an XCD owns is the same either way.
__device__ __forceinline__ void xcd_tile_map(int gx, int gy, int& x, int& y, int& z, bool heavy_first = false) {
    const int total = gridDim.x, lin = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = lin & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, i = lin >> 3;
    int head;
    x = (start + i) % gx;
    head = (start + i) / gx;
#ifdef ATTN_PROBE_HEAD_MAJOR      // A/B build (tools/build_probe_lib.sh): the round-2 head-major walk
    heavy_first = false;
#endif
    if (heavy_first) {
        const int last = start + (xcd < r ? q : q - 1);
        const int h0 = start / gx, x0 = start - h0 * gx, h1 = last / gx, x1 = last - h1 * gx;
        if (h0 != h1) {
            // column x of the range holds every head h0 .. h1, minus h0 where x < x0, minus h1 where x > x1: three runs of columns
            const int nh = h1 - h0 + 1;
            const int a = x0 < x1 + 1 ? x0 : x1 + 1, b = x0 < x1 + 1 ? x1 + 1 : x0;
            const int n2 = x0 <= x1 + 1 ? nh : nh - 2;
            const int c1 = a * (nh - 1), c2 = c1 + (b - a) * n2;
            if (i < c1) {
                x = i / (nh - 1);
                head = h0 + 1 + (i - x * (nh - 1));
            } else if (i < c2) {
                const int j = i - c1, xx = j / n2;
                x = a + xx;
                head = (n2 == nh ? h0 : h0 + 1) + (j - xx * n2);
            } else {
                const int j = i - c2, xx = j / (nh - 1);
                x = b + xx;
                head = h0 + (j - xx * (nh - 1));
            }
        }
    }
    y = head % gy;
    z = head / gy;
}
