// Power / issue probe for the fp8 GEMM main loop (no global traffic inside the loop): v_mfma_scale_f32_32x32x64_f8f6f4 (unit scales) fed
// by conflict-free ds_read_b128 fragment reads from a resident LDS tile, 8 waves x (128 x 64) -- the shape of csrc/gemm_fp8.hip's 256x256
// kernel.  Question: what does the matrix pipe sustain in fp8 on zero / random operands (the bf16 pipe: 2.1-2.2 / 1.53-1.55 PFLOP/s,
// profiles/r02_experiments.md)?  That number, not the 5 PFLOP/s nominal peak, is the roof the fp8 GEMM can be measured against.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_fp8_probe.hip -o tools/_bin/mfma_fp8_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int OFF>
__device__ __forceinline__ void rd(i32x4& d, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "i"(OFF)); }

__device__ __forceinline__ i32x8 cat(const i32x4& a, const i32x4& b) {
    i32x8 r;
    r[0] = a[0], r[1] = a[1], r[2] = a[2], r[3] = a[3], r[4] = b[0], r[5] = b[1], r[6] = b[2], r[7] = b[3];
    return r;
}

template <int FA>
__global__ __launch_bounds__(512) void probe(const unsigned char* __restrict__ src, float* __restrict__ out, int ksteps, int zero) {
    constexpr int TM = 4, TN = 2;
    __shared__ __attribute__((aligned(16))) char smem[65536];          // A rows 0-255 (32 KiB) | B rows 0-255 (32 KiB), 128 B per row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += 512) {
        uint4 v = reinterpret_cast<const uint4*>(src)[(blockIdx.x * 4096 + i) % (1 << 20)];
        if (zero) v = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(smem)[i] = v;
    }
    __syncthreads();
    const int wm = wave / 4, wn = wave % 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned rowoff = (unsigned)(lane & 31) * 128u, f = ((unsigned)(lane & 31) >> 1) & 7u;
    unsigned xo[2][2];
    for (int ks = 0; ks < 2; ++ks)
        for (int h = 0; h < 2; ++h) xo[ks][h] = rowoff + ((((unsigned)(ks * 4 + (lane >> 5) * 2 + h) + f) & 7u) << 4);
    const unsigned a_base = lds0 + wm * (TM * 4096), b_base = lds0 + 32768 + wn * (TN * 4096);
    f32x16 acc[TN][TM];
    for (int i = 0; i < TN; ++i) for (int j = 0; j < TM; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    i32x4 fa[2][TM][2], fb[2][TN][2];
    auto reads = [&](int ks, int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            rd<0>(fa[buf][0][h], a_base + xo[ks][h]); rd<4096>(fa[buf][1][h], a_base + xo[ks][h]);
            rd<8192>(fa[buf][2][h], a_base + xo[ks][h]); rd<12288>(fa[buf][3][h], a_base + xo[ks][h]);
            rd<0>(fb[buf][0][h], b_base + xo[ks][h]); rd<4096>(fb[buf][1][h], b_base + xo[ks][h]);
        }
    };
    reads(0, 0);
    for (int t = 0; t < ksteps; ++t) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cb = ks & 1, nb = cb ^ 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            reads((ks + 1) & 1, nb);
#pragma unroll
            for (int i = 0; i < TN * TM; ++i)
                acc[i / TM][i % TM] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat(fb[cb][i / TM][0], fb[cb][i / TM][1]),
                                                                                      cat(fa[cb][i % TM][0], fa[cb][i % TM][1]),
                                                                                      acc[i / TM][i % TM], 0, FA, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < TN; ++i) for (int j = 0; j < TM; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 512 + tid] = s;
}

template <int FA>
static void run(const char* name, const unsigned char* src, float* out, int ksteps, int zero) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<FA>), dim3(256), dim3(512), 0, 0, src, out, ksteps, zero);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * ksteps * 2.0 * 8 * (2.0 * 32 * 32 * 64);
        if (rep) printf("%s %s: %8.1f us  %7.1f TFLOP/s\n", name, zero ? "zeros " : "random", ms * 1e3, flop / ms / 1e9);
    }
}

int main() {
    std::vector<unsigned char> h(1 << 24);
    srand(1);
    for (auto& x : h) x = (unsigned char)(((rand() & 1) << 7) | ((4 + rand() % 7) << 3) | (rand() & 7));   // finite e4m3 values, random sign
    unsigned char* src; float* out;
    hipMalloc(&src, h.size()); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice);
    const int ksteps = 3000;
    for (int zero = 1; zero >= 0; --zero) {
        run<0>("fp8 e4m3 x e4m3, 8 waves x 128x64", src, out, ksteps, zero);
        run<1>("fp8 e4m3 x e5m2, 8 waves x 128x64", src, out, ksteps, zero);
    }
    return 0;
}
