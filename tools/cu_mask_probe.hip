// Probe for "AdamW beside the frozen tower" (DESIGN.md section 7, item 4): how much HBM bandwidth does a streaming kernel get when its
// stream is restricted to N CUs by hipExtStreamCreateWithCUMask, and do two masked streams (streaming kernel | MFMA kernel) overlap?
// Build: hipcc -O3 --offload-arch=gfx950 tools/cu_mask_probe.hip -o tools/_bin/cu_mask_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void stream_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ c, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint4 x = a[i], y = b[i];
        c[i] = make_uint4(x.x + y.x, x.y ^ y.y, x.z + y.z, x.w ^ y.w);
    }
}
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    for (int t = 0; t < iters; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static hipStream_t masked_stream(int first, int count, int total) {
    std::vector<uint32_t> m((total + 31) / 32, 0u);
    for (int i = first; i < first + count; ++i) m[i / 32] |= 1u << (i % 32);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()) != hipSuccess) { printf("cu mask stream failed\n"); exit(1); }
    return s;
}

int main() {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const long n = (1L << 30) / 16 * 2;          // 2 GiB per array
    uint4 *a, *b, *c; float* out;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&out, 4096 * 256 * 4);
    hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("CUs: %d\n", ncu);
    for (int cus : {256, 192, 128, 96, 64, 32}) {
        hipStream_t s = masked_stream(0, cus, ncu);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(stream_kernel, dim3(131072), dim3(256), 0, s, a, b, c, n);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("streaming kernel on the first %3d CU bits: %7.2f ms  %5.2f TB/s\n", cus, best, 3.0 * n * 16 / best / 1e9);
        hipStreamDestroy(s);
    }
    // overlap: streaming kernel on the LAST 64 bits, MFMA kernel on the first 192 bits
    hipStream_t sa = masked_stream(ncu - 64, 64, ncu), sm = masked_stream(0, 192, ncu);
    const int iters = 400000;
    float t_stream, t_mfma, t_both;
    hipEventRecord(e0, sa); hipLaunchKernelGGL(stream_kernel, dim3(131072), dim3(256), 0, sa, a, b, c, n); hipEventRecord(e1, sa);
    hipEventSynchronize(e1); hipEventElapsedTime(&t_stream, e0, e1);
    hipEventRecord(e0, sm); hipLaunchKernelGGL(mfma_kernel, dim3(192 * 4), dim3(256), 0, sm, out, iters); hipEventRecord(e1, sm);
    hipEventSynchronize(e1); hipEventElapsedTime(&t_mfma, e0, e1);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipStreamWaitEvent(sa, e0, 0); hipStreamWaitEvent(sm, e0, 0);
    hipLaunchKernelGGL(mfma_kernel, dim3(192 * 4), dim3(256), 0, sm, out, iters);
    hipLaunchKernelGGL(stream_kernel, dim3(131072), dim3(256), 0, sa, a, b, c, n);
    hipEvent_t ea, em; hipEventCreate(&ea); hipEventCreate(&em);
    hipEventRecord(ea, sa); hipEventRecord(em, sm);
    hipStreamWaitEvent(0, ea, 0); hipStreamWaitEvent(0, em, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&t_both, e0, e1);
    printf("streaming on 64 CUs alone %.2f ms, MFMA on 192 CUs alone %.2f ms, both at once %.2f ms (sum %.2f)\n", t_stream, t_mfma, t_both,
           t_stream + t_mfma);
    return 0;
}
