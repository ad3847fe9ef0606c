// MFMA-gap filler probe (round 3): with ONE wave per SIMD, how many single-issue instructions hide behind a v_mfma_f32_32x32x16_bf16,
// and does it depend on where the MFMA's operands live?  Each variant runs a long loop of 16-MFMA groups with N fillers after every MFMA
// and reports s_memtime cycles per MFMA (wave 0 of every workgroup, averaged).
//   acc:   'a' = C/D in AGPRs (O-style)          'v' = C/D in arch VGPRs (S-style)
//   ab:    'v' = A/B operands in VGPRs           'a' = A/B operands in AGPRs
//   filler kinds: fma (independent v_fma_f32), chain (v_fma -> v_exp on its result -> v_add: the softmax element), exp (independent v_exp_f32),
//                 nop (s_nop 0), lds (ds_read_b64_tr_b16, counted wait at the group's end)
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_filler_probe.hip -o tools/_bin/mfma_filler_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { F_FMA, F_CHAIN, F_EXP, F_NOP, F_LDS };

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

template <bool ACC_A, bool AB_A>
__device__ __forceinline__ void mfma(f32x16& c, const bf16x8& a, const bf16x8& b) {
    if constexpr (ACC_A && !AB_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else if constexpr (ACC_A && AB_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "a"(a), "a"(b));
    else if constexpr (!ACC_A && !AB_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "a"(b));
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int KIND>
__device__ __forceinline__ void filler(float& x0, float& x1, float& x4, float& y, float& l, s16x4& ld, unsigned lds0) {
    if constexpr (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(y));
    else if constexpr (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x0));
    else if constexpr (KIND == F_NOP) asm volatile("s_nop 0");
    else if constexpr (KIND == F_LDS) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ld) : "v"(lds0));
    else       // the softmax element, 3 instructions per "filler": N counts elements
        asm volatile("v_fma_f32 %0, %1, %2, %2\n\tv_exp_f32 %0, %0\n\tv_add_f32 %3, %3, %4" : "+v"(x0) : "v"(x1), "v"(y), "v"(l), "v"(x4));
}

template <bool ACC_A, bool AB_A, int KIND, int N>
__global__ __launch_bounds__(256, 1) void probe(float* __restrict__ out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) char smem[98304];      // > half the LDS: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 98304 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = seed * (float)(i & 255);
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)lane * 8u;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a[4], b;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) a[j][e] = (__bf16)(seed * (float)((lane + e + j) & 7));
    for (int e = 0; e < 8; ++e) b[e] = (__bf16)(seed * (float)((lane * 3 + e) & 7));
    float x[8], y = seed, l = 0.f;
    for (int e = 0; e < 8; ++e) x[e] = seed * (float)(lane + e);
    s16x4 ld[4];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        sfor<0, 16>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            mfma<ACC_A, AB_A>(acc[j & 3], a[j & 3], b);
            sfor<0, N>([&](auto kc) {
                constexpr int k = decltype(kc)::value, r = (j * N + k) & 7;
                filler<KIND>(x[r], x[(r + 1) & 7], x[(r + 4) & 7], y, l, ld[k & 3], lds0);
            });
        });
        if constexpr (KIND == F_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = l;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    for (int e = 0; e < 8; ++e) s += x[e];
    if constexpr (KIND == F_LDS) s += (float)ld[0][0] + (float)ld[1][0] + (float)ld[2][0] + (float)ld[3][0];
    if (s == 12345.678f) out[gridDim.x * 4 + threadIdx.x] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = (float)(t1 - t0) / (float)(iters * 16);
}

template <bool ACC_A, bool AB_A, int KIND, int N>
static void run(const char* name, float* d_out, int iters) {
    const int grid = 256;
    hipLaunchKernelGGL((probe<ACC_A, AB_A, KIND, N>), dim3(grid), dim3(256), 0, 0, d_out, 16, 0.f);
    hipLaunchKernelGGL((probe<ACC_A, AB_A, KIND, N>), dim3(grid), dim3(256), 0, 0, d_out, iters, 0.f);
    hipDeviceSynchronize();
    std::vector<float> h(grid * 4);
    hipMemcpy(h.data(), d_out, grid * 4 * sizeof(float), hipMemcpyDeviceToHost);
    double m = 0, mx = 0;
    for (float v : h) { m += v; mx = v > mx ? v : mx; }
    printf("%-34s N=%d  %6.1f cycles per MFMA (max %6.1f)\n", name, N, m / h.size(), mx);
    fflush(stdout);
}

template <bool ACC_A, bool AB_A, int KIND>
static void sweep(const char* name, float* d_out, int iters) {
    run<ACC_A, AB_A, KIND, 0>(name, d_out, iters);
    run<ACC_A, AB_A, KIND, 2>(name, d_out, iters);
    run<ACC_A, AB_A, KIND, 4>(name, d_out, iters);
    run<ACC_A, AB_A, KIND, 5>(name, d_out, iters);
    run<ACC_A, AB_A, KIND, 6>(name, d_out, iters);
    run<ACC_A, AB_A, KIND, 8>(name, d_out, iters);
}

int main() {
    float* d_out;
    hipMalloc(&d_out, (256 * 4 + 256) * sizeof(float));
    const int iters = 2000;
    sweep<true, false, F_FMA>("acc AGPR, A/B VGPR, v_fma", d_out, iters);
    sweep<false, false, F_FMA>("acc VGPR, A/B VGPR, v_fma", d_out, iters);
    sweep<false, true, F_FMA>("acc VGPR, A/B AGPR, v_fma", d_out, iters);
    sweep<true, true, F_FMA>("acc AGPR, A/B AGPR, v_fma", d_out, iters);
    sweep<true, false, F_EXP>("acc AGPR, A/B VGPR, v_exp", d_out, iters);
    sweep<true, false, F_NOP>("acc AGPR, A/B VGPR, s_nop 0", d_out, iters);
    sweep<true, false, F_LDS>("acc AGPR, A/B VGPR, ds_read_tr", d_out, iters);
    run<true, false, F_CHAIN, 1>("acc AGPR, A/B VGPR, 1 softmax element", d_out, iters);
    run<false, true, F_CHAIN, 1>("acc VGPR, A/B AGPR, 1 softmax element", d_out, iters);
    run<true, false, F_CHAIN, 2>("acc AGPR, A/B VGPR, 2 softmax elements", d_out, iters);
    run<false, true, F_CHAIN, 2>("acc VGPR, A/B AGPR, 2 softmax elements", d_out, iters);
    return 0;
}
