// AdamW state layout probe (round 3): does the optimizer pass stream faster when the three fp32 state arrays (master, exp_avg, exp_avg_sq)
// are interleaved in 6 KiB chunks (512 elements: [master 2 KiB | m 2 KiB | v 2 KiB]) instead of three separate arrays?  The pass is
// HBM-bound at 28 B per element with reads and writes 1 : 1; fewer concurrent DRAM streams could mean fewer page conflicts.
// Same arithmetic as csrc/optim.hip:adamw_kernel.  Build: hipcc -O3 --offload-arch=gfx950 tools/adamw_layout_probe.hip -o tools/_bin/adamw_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;
__device__ __forceinline__ float lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pk(float a, float b) {      // truncating pack is enough for a bandwidth probe
    return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
}
__device__ __forceinline__ void upd(const u32x4 g, f32x4& pa, f32x4& pb, f32x4& ma, f32x4& mb, f32x4& va, f32x4& vb, u32x4& o) {
    const float lr = 1e-5f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, bc1 = 0.1f, bc2 = 0.001f;
    float gf[8], pf[8], mf[8], vf[8];
    for (int e = 0; e < 4; ++e) {
        gf[2 * e] = lo(g[e]); gf[2 * e + 1] = hi(g[e]);
        pf[e] = pa[e]; pf[4 + e] = pb[e]; mf[e] = ma[e]; mf[4 + e] = mb[e]; vf[e] = va[e]; vf[4 + e] = vb[e];
    }
    for (int e = 0; e < 8; ++e) {
        mf[e] = b1 * mf[e] + (1.f - b1) * gf[e];
        vf[e] = b2 * vf[e] + (1.f - b2) * gf[e] * gf[e];
        pf[e] -= (lr / bc1) * (mf[e] / (sqrtf(vf[e] / bc2) + eps));
    }
    for (int e = 0; e < 4; ++e) {
        o[e] = pk(pf[2 * e], pf[2 * e + 1]);
        pa[e] = pf[e]; pb[e] = pf[4 + e]; ma[e] = mf[e]; mb[e] = mf[4 + e]; va[e] = vf[e]; vb[e] = vf[4 + e];
    }
}
__global__ void adamw_split(bf16_t* p16, const bf16_t* g16, float* p32, float* m, float* v, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const u32x4 g = *reinterpret_cast<const u32x4*>(g16 + i * 8);
        f32x4 pa = *reinterpret_cast<f32x4*>(p32 + i * 8), pb = *reinterpret_cast<f32x4*>(p32 + i * 8 + 4);
        f32x4 ma = *reinterpret_cast<f32x4*>(m + i * 8), mb = *reinterpret_cast<f32x4*>(m + i * 8 + 4);
        f32x4 va = *reinterpret_cast<f32x4*>(v + i * 8), vb = *reinterpret_cast<f32x4*>(v + i * 8 + 4);
        u32x4 o;
        upd(g, pa, pb, ma, mb, va, vb, o);
        *reinterpret_cast<f32x4*>(p32 + i * 8) = pa; *reinterpret_cast<f32x4*>(p32 + i * 8 + 4) = pb;
        *reinterpret_cast<f32x4*>(m + i * 8) = ma;   *reinterpret_cast<f32x4*>(m + i * 8 + 4) = mb;
        *reinterpret_cast<f32x4*>(v + i * 8) = va;   *reinterpret_cast<f32x4*>(v + i * 8 + 4) = vb;
        *reinterpret_cast<u32x4*>(p16 + i * 8) = o;
    }
}
// state: chunks of 512 elements = 1536 floats: [master 512 | m 512 | v 512]
__global__ void adamw_packed(bf16_t* p16, const bf16_t* g16, float* st, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long chunk = i >> 6, w = i & 63;
        float* base = st + chunk * 1536 + w * 8;
        const u32x4 g = *reinterpret_cast<const u32x4*>(g16 + i * 8);
        f32x4 pa = *reinterpret_cast<f32x4*>(base), pb = *reinterpret_cast<f32x4*>(base + 4);
        f32x4 ma = *reinterpret_cast<f32x4*>(base + 512), mb = *reinterpret_cast<f32x4*>(base + 516);
        f32x4 va = *reinterpret_cast<f32x4*>(base + 1024), vb = *reinterpret_cast<f32x4*>(base + 1028);
        u32x4 o;
        upd(g, pa, pb, ma, mb, va, vb, o);
        *reinterpret_cast<f32x4*>(base) = pa; *reinterpret_cast<f32x4*>(base + 4) = pb;
        *reinterpret_cast<f32x4*>(base + 512) = ma; *reinterpret_cast<f32x4*>(base + 516) = mb;
        *reinterpret_cast<f32x4*>(base + 1024) = va; *reinterpret_cast<f32x4*>(base + 1028) = vb;
        *reinterpret_cast<u32x4*>(p16 + i * 8) = o;
    }
}
// everything of a 512-element chunk in one 8 KiB record: [master | m | v | grad (bf16) | param (bf16)]
__global__ void adamw_record(char* rec, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long chunk = i >> 6, w = i & 63;
        char* cb = rec + chunk * 8192;
        float* base = reinterpret_cast<float*>(cb) + w * 8;
        const u32x4 g = *reinterpret_cast<const u32x4*>(cb + 6144 + w * 16);
        f32x4 pa = *reinterpret_cast<f32x4*>(base), pb = *reinterpret_cast<f32x4*>(base + 4);
        f32x4 ma = *reinterpret_cast<f32x4*>(base + 512), mb = *reinterpret_cast<f32x4*>(base + 516);
        f32x4 va = *reinterpret_cast<f32x4*>(base + 1024), vb = *reinterpret_cast<f32x4*>(base + 1028);
        u32x4 o;
        upd(g, pa, pb, ma, mb, va, vb, o);
        *reinterpret_cast<f32x4*>(base) = pa; *reinterpret_cast<f32x4*>(base + 4) = pb;
        *reinterpret_cast<f32x4*>(base + 512) = ma; *reinterpret_cast<f32x4*>(base + 516) = mb;
        *reinterpret_cast<f32x4*>(base + 1024) = va; *reinterpret_cast<f32x4*>(base + 1028) = vb;
        *reinterpret_cast<u32x4*>(cb + 7168 + w * 16) = o;
    }
}
int main() {
    const long n = 1L << 30, n8 = n / 8;      // 1 Gi elements: 28 GiB of traffic per pass
    bf16_t *p16, *g16; float *p32, *m, *v, *st; char* rec;
    hipMalloc(&p16, n * 2); hipMalloc(&g16, n * 2); hipMalloc(&p32, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4);
    hipMalloc(&st, n * 12); hipMalloc(&rec, n * 16);
    hipMemset(p16, 0, n * 2); hipMemset(g16, 0x3c, n * 2); hipMemset(p32, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    hipMemset(st, 0, n * 12); hipMemset(rec, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {65536, 131072, 262144}) {
        for (int kind = 0; kind < 3; ++kind) {
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(adamw_split, dim3(grid), dim3(256), 0, 0, p16, g16, p32, m, v, n8);
                else if (kind == 1) hipLaunchKernelGGL(adamw_packed, dim3(grid), dim3(256), 0, 0, p16, g16, st, n8);
                else hipLaunchKernelGGL(adamw_record, dim3(grid), dim3(256), 0, 0, rec, n8);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it && ms < best) best = ms;
            }
            printf("grid %6d  %-28s %7.3f ms  %5.2f TB/s\n", grid, kind == 0 ? "five separate arrays" : kind == 1 ? "state packed per 512 elems" : "one record per 512 elems",
                   best, 28.0 * n / best / 1e9);
            fflush(stdout);
        }
    }
    return 0;
}
