// MFMA-shape power probe (round 3): does the matrix pipe's random-data roof depend on the MFMA shape?
// The vendor library's gfx950 bf16 kernels are 4 waves x (128 x 128) of v_mfma_f32_16x16x32_bf16 (kernel names / disassembly of the
// installed code objects); this library's ring GEMM is 8 waves x (128 x 64) of v_mfma_f32_32x32x16_bf16.  Same FLOP/clk on paper.
// Variants (no global traffic in the loop; LDS-fed ones read conflict-free ds_read_b128 fragments from a resident random tile):
//   R32  register-only 32x32x16      R16  register-only 16x16x32
//   A32  8 waves x 128x64,  32x32x16 (24 reads / 32 MFMA)      B32  4 waves x 128x128, 32x32x16 (32 reads / 64 MFMA, per K=64: x4)
//   B16  4 waves x 128x128, 16x16x32 (16 reads / 64 MFMA)      A16  8 waves x 128x64,  16x16x32 (12 reads / 32 MFMA)
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o tools/_bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void rd(bf16x8& d, unsigned a) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a)); }
// accumulators pinned in AGPRs, in place (the compiler's own allocation shuffles tuples through temporaries at 512 registers)
__device__ __forceinline__ void mfma32(f32x16& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

__device__ __forceinline__ void fill(char* smem, const unsigned short* src, int zero, int nthreads) {
    for (int i = threadIdx.x; i < 65536 / 16; i += nthreads) {
        uint4 v = reinterpret_cast<const uint4*>(src)[(blockIdx.x * 4096 + i) % (1 << 20)];
        if (zero) v = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(smem)[i] = v;
    }
    __syncthreads();
}

// ---- 32x32x16, WAVES x (TM*32 x TN*32) wave tiles, LDS-fed (LDS = 0: registers only)
template <int WAVES, int TM, int TN, int LDS>
__global__ __launch_bounds__(WAVES * 64) void probe32(const unsigned short* __restrict__ src, float* __restrict__ out, int ksteps, int zero) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fill(smem, src, zero, WAVES * 64);
    constexpr int NWN = 256 / (TN * 32);
    const int wm = wave / NWN, wn = wave % NWN;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned rowoff = (unsigned)(lane & 31) * 128u, f = ((unsigned)(lane & 31) >> 1) & 7u;
    unsigned xo[4];
    for (int ks = 0; ks < 4; ++ks) xo[ks] = rowoff + ((((unsigned)(ks * 2 + (lane >> 5))) ^ f) << 4);
    const unsigned a_base = lds0 + wm * (TM * 4096), b_base = lds0 + 32768 + wn * (TN * 4096);
    f32x16 acc[TN][TM];
    for (int i = 0; i < TN; ++i) for (int j = 0; j < TM; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa[2][TM], fb[2][TN];
    auto rdop = [&](int op, int ks, int buf) {
        if (op < TM) rd(fa[buf][op], a_base + xo[ks] + op * 4096);
        else rd(fb[buf][op - TM], b_base + xo[ks] + (op - TM) * 4096);
    };
    for (int b = 0; b < 2; ++b) for (int op = 0; op < TM + TN; ++op) rdop(op, b, b);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int t = 0; t < ksteps; ++t) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cb = ks & 1, nb = cb ^ 1;
            if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN * TM; ++i) {
                mfma32(acc[i / TM][i % TM], fb[cb][i / TM], fa[cb][i % TM]);
                if (LDS && i < TM + TN) rdop(i, (ks + 1) & 3, nb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < TN; ++i) for (int j = 0; j < TM; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * (WAVES * 64) + tid] = s;
}

// ---- 16x16x32, WAVES x (TM*16 x TN*16) wave tiles.  A K=64 slab holds two K=32 halves; lane (r = lane & 15, kg = lane >> 4) reads the
// 16 B at row r, chunk kh*4 + kg (XOR-swizzled with (row >> 1) & 7 like the 32x32 layout)
template <int WAVES, int TM, int TN, int LDS>
__global__ __launch_bounds__(WAVES * 64) void probe16(const unsigned short* __restrict__ src, float* __restrict__ out, int ksteps, int zero) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fill(smem, src, zero, WAVES * 64);
    constexpr int NWN = 256 / (TN * 16);
    const int wm = wave / NWN, wn = wave % NWN;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned r = lane & 15, kg = lane >> 4;
    const unsigned rowoff = r * 128u, f = (r >> 1) & 7u;
    unsigned xo[2];
    for (int kh = 0; kh < 2; ++kh) xo[kh] = rowoff + ((((unsigned)(kh * 4) + kg) ^ f) << 4);
    const unsigned a_base = lds0 + wm * (TM * 2048), b_base = lds0 + 32768 + wn * (TN * 2048);
    f32x4 acc[TN][TM];
    for (int i = 0; i < TN; ++i) for (int j = 0; j < TM; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa[2][TM], fb[2][TN];
    auto rdop = [&](int op, int kh, int buf) {
        if (op < TM) rd(fa[buf][op], a_base + xo[kh] + op * 2048);
        else rd(fb[buf][op - TM], b_base + xo[kh] + (op - TM) * 2048);
    };
    for (int b = 0; b < 2; ++b) for (int op = 0; op < TM + TN; ++op) rdop(op, b, b);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int STRIDE = (TM * TN) / (TM + TN);        // one fragment read every STRIDE MFMAs
    for (int t = 0; t < ksteps; ++t) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int cb = kh, nb = kh ^ 1;
            if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN * TM; ++i) {
                mfma16(acc[i / TM][i % TM], fb[cb][i / TM], fa[cb][i % TM]);
                // NOTE: reading into the buffer being consumed would be a hazard in a real kernel; here buffer nb was consumed in the
                // previous half, and the MFMAs of this half only read cb
                if (LDS && (i % STRIDE) == 0 && i / STRIDE < TM + TN) rdop(i / STRIDE, nb, nb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < TN; ++i) for (int j = 0; j < TM; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    out[blockIdx.x * (WAVES * 64) + tid] = s;
}

template <typename KFn>
static void run(const char* name, KFn kern, int waves, double flop_per_kstep_per_wave, const unsigned short* src, float* out, int ksteps, int zero) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, src, out, ksteps, zero);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * waves * ksteps * flop_per_kstep_per_wave;
        if (rep) printf("%s %s: %8.1f us  %7.1f TFLOP/s\n", name, zero ? "zeros " : "random", ms * 1e3, flop / ms / 1e9);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  !! %s\n", hipGetErrorString(e));
}

int main() {
    std::vector<unsigned short> h(1 << 23);
    srand(1);
    for (auto& x : h) x = (unsigned short)(((rand() & 1) << 15) | ((120 + (rand() & 7)) << 7) | (rand() & 127));
    unsigned short* src; float* out;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int ksteps = 3000;
    // FLOPs per K=64 step per wave: 32x32x16: 4 k-chunks x TM*TN x 32768; 16x16x32: 2 halves x TM*TN x 16384
    for (int zero = 1; zero >= 0; --zero) {
        run("R32 regs only, 8w 128x64,  32x32x16", probe32<8, 4, 2, 0>, 8, 4.0 * 8 * 32768, src, out, ksteps, zero);
        run("R16 regs only, 4w 128x128, 16x16x32", probe16<4, 8, 8, 0>, 4, 2.0 * 64 * 16384, src, out, ksteps, zero);
        run("R16 regs only, 8w 128x64,  16x16x32", probe16<8, 8, 4, 0>, 8, 2.0 * 32 * 16384, src, out, ksteps, zero);
        run("A32 LDS-fed,   8w 128x64,  32x32x16", probe32<8, 4, 2, 1>, 8, 4.0 * 8 * 32768, src, out, ksteps, zero);
        run("B32 LDS-fed,   4w 128x128, 32x32x16", probe32<4, 4, 4, 1>, 4, 4.0 * 16 * 32768, src, out, ksteps, zero);
        run("B16 LDS-fed,   4w 128x128, 16x16x32", probe16<4, 8, 8, 1>, 4, 2.0 * 64 * 16384, src, out, ksteps, zero);
        run("A16 LDS-fed,   8w 128x64,  16x16x32", probe16<8, 8, 4, 1>, 8, 2.0 * 32 * 16384, src, out, ksteps, zero);
    }
    return 0;
}
