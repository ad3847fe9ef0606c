// Probe for the next step on `rmsnorm_bwd_kernel` (DESIGN.md section 7 / profiles/r02_experiments.md): the shipped kernel gives one wave a
// whole row and reads dy / x twice (58 us for 5624 x 4096 = 3.2 TB/s; 5624 rows over 4096 resident waves = 2 serial rounds of
// [HBM round trip -> reduce -> L2 re-read + dres -> store]).  Variant Q here:
//   * a row is split over NQ = 4 waves (2 chunks of 8 per lane), two row slots per 512-thread workgroup;
//   * dy, x, dres of a row stay in registers between the two passes (no second read), the weight chunks stay in registers for good;
//   * the NEXT row's dy / x / rstd are loaded before the current row is reduced (software prefetch: the HBM round trip of row i+1
//     overlaps the reduce + store of row i); the current row's dres is requested at the top of the iteration and used after the barrier;
//   * the four partial dots of a row meet in a double-buffered LDS cell: one workgroup barrier per row pair.
// Self-check against the shipped kernel (same inputs; dx to bf16 rounding, dW to fp32 summation order), then timing of both.
// NEVER RUN YET (written at the end of round 2 without GPU minutes left).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -I mantis_amd/csrc tools/rmsnorm_bwd_probe.hip -o tools/_bin/rmsnorm_bwd_probe && tools/_bin/rmsnorm_bwd_probe
// Offline resource check (no GPU needed): add  -Rpass-analysis=kernel-resource-usage
#include "../mantis_amd/csrc/norm.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define Q_NQ 4        // waves per row
#define Q_SLOTS 2     // rows per workgroup iteration
#define Q_MAXC 2      // chunks of 8 bf16 per lane: d = 8 * 64 * Q_NQ * Q_MAXC = 4096

// Wave-wide sum without LDS traffic or index registers: `wave_sum` of common.h is six dependent ds_bpermute (LDS-latency each, and their
// lane-index registers are what the shipped kernel spills: 7 VGPRs to scratch at 128).  DPP: xor 1, xor 2 inside a quad, mirror inside 8
// and 16 lanes (every lane of a 16-lane row then holds the row's sum), the four rows meet through v_readlane.  Result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);       // row_half_mirror
    v = dpp_add<0x140>(v);       // row_mirror
    const int i = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
    return (r0 + r1) + (r2 + r3);
}

struct RowRegs {
    u32x4 vd[Q_MAXC], vx[Q_MAXC];
    float rstd;
};

// Row r of the three inputs through buffer loads: descriptor over the whole tensor, the row's byte offset as the SCALAR offset (r is
// wave-uniform), the lane's byte offset in one 32-bit register -- no per-lane 64-bit pointers for the loop optimiser to multiply.  Rows
// past the end are out of the descriptor's range: they load zeros (xhat = 0: nothing added to the dot or to dW) and their store is dropped.
typedef unsigned int u32x4v __attribute__((vector_size(16)));
__device__ __forceinline__ u32x4 q_ld(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const u32x4v t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    u32x4 o;
    o[0] = t[0], o[1] = t[1], o[2] = t[2], o[3] = t[3];
    return o;
}
__device__ __forceinline__ void q_load(RowRegs& t, __amdgpu_buffer_rsrc_t rsD, __amdgpu_buffer_rsrc_t rsX,
                                       const float* __restrict__ rstd_in, long r, long rows, int d, unsigned lane_bytes) {
    const unsigned soff = (unsigned)(r * d * 2);
#pragma unroll
    for (int k = 0; k < Q_MAXC; ++k) {
        t.vd[k] = q_ld(rsD, lane_bytes + 1024u * k, soff);
        t.vx[k] = q_ld(rsX, lane_bytes + 1024u * k, soff);
    }
    t.rstd = r < rows ? rstd_in[r] : 0.f;
}

__global__ __launch_bounds__(64 * Q_NQ * Q_SLOTS, 4) void rmsnorm_bwd_q_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
    const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, float* __restrict__ dw_partial, long rows, int d) {
    __shared__ float dotbuf[2][Q_SLOTS][Q_NQ];
    __shared__ float fold[Q_NQ * Q_MAXC * 8 * 64];            // one fp32 row of d columns
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: row index and row offsets live in SGPRs
    const int slot = wv / Q_NQ, q = wv % Q_NQ;
    const int c0 = q * (Q_MAXC * 64) + lane;                   // chunk index of k = 0; k-th chunk = c0 + 64 k
    u32x4 vw[Q_MAXC];
#pragma unroll
    for (int k = 0; k < Q_MAXC; ++k) vw[k] = *reinterpret_cast<const u32x4*>(w + (long)(c0 + 64 * k) * 8);
    float dwacc[Q_MAXC][8];
#pragma unroll
    for (int k = 0; k < Q_MAXC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) dwacc[k][e] = 0.f;
    const long stride = (long)gridDim.x * Q_SLOTS;
    const int niter = (int)((rows + stride - 1) / stride);    // the same for every wave of the grid: barriers stay uniform
    long r = (long)blockIdx.x * Q_SLOTS + slot;
    RowRegs cur, nxt;
    const unsigned lane_bytes = (unsigned)c0 * 16u;
    const int nbytes = (int)(unsigned)(rows * d * 2);          // < 4 GiB (the product entry would check)
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)dres, 0, dres ? nbytes : 0, 0x00020000);   // no residual: zeros
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)dx, 0, nbytes, 0x00020000);
    q_load(cur, rsD, rsX, rstd_in, r, rows, d, lane_bytes);
#pragma unroll 1
    for (int it = 0; it < niter; ++it) {
        q_load(nxt, rsD, rsX, rstd_in, r + stride, rows, d, lane_bytes);
        // the residual gradient of the CURRENT row is needed only after the barrier: its round trip hides behind the first pass
        u32x4 vr[Q_MAXC];
#pragma unroll
        for (int k = 0; k < Q_MAXC; ++k) vr[k] = q_ld(rsR, lane_bytes + 1024u * k, (unsigned)(r * d * 2));
        // keep the weight chunks PACKED across iterations (the compiler would hoist their 16 unpacked floats out of the loop)
#pragma unroll
        for (int k = 0; k < Q_MAXC; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(vw[k][e]));
        const float rstd = cur.rstd;
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < Q_MAXC; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = bf2f_lo(cur.vd[k][e]), d1 = bf2f_hi(cur.vd[k][e]);
                const float x0 = bf2f_lo(cur.vx[k][e]) * rstd, x1 = bf2f_hi(cur.vx[k][e]) * rstd;
                dot += d0 * bf2f_lo(vw[k][e]) * x0 + d1 * bf2f_hi(vw[k][e]) * x1;
                dwacc[k][2 * e] += d0 * x0;
                dwacc[k][2 * e + 1] += d1 * x1;
            }
        dot = wave_sum_dpp(dot);
        if (lane == 0) dotbuf[it & 1][slot][q] = dot;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < Q_NQ; ++j) tot += dotbuf[it & 1][slot][j];     // fixed order: deterministic
        tot /= (float)d;
        // the second pass unpacks the SAME packed registers again: without this fence the compiler keeps the 32 unpacked floats of the first
        // pass alive across the barrier (common sub-expressions) and spills
#pragma unroll
        for (int k = 0; k < Q_MAXC; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(cur.vd[k][e]), "+v"(cur.vx[k][e]));
        {
#pragma unroll
            for (int k = 0; k < Q_MAXC; ++k) {
                u32x4v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = bf2f_lo(cur.vx[k][e]) * rstd, x1 = bf2f_hi(cur.vx[k][e]) * rstd;
                    const float a = rstd * (bf2f_lo(cur.vd[k][e]) * bf2f_lo(vw[k][e]) - x0 * tot) + bf2f_lo(vr[k][e]);
                    const float b = rstd * (bf2f_hi(cur.vd[k][e]) * bf2f_hi(vw[k][e]) - x1 * tot) + bf2f_hi(vr[k][e]);
                    o[e] = pack_bf2(a, b);
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, rsO, lane_bytes + 1024u * k, (unsigned)(r * d * 2), 0);   // rows past the end: dropped
            }
        }
        cur = nxt;
        r += stride;
    }
    if (dw_partial) {
        // the two row slots own the same columns: slot 0 writes, slot 1 adds (fixed order), then the row goes out coalesced
        for (int s = 0; s < Q_SLOTS; ++s) {
            if (slot == s) {
#pragma unroll
                for (int k = 0; k < Q_MAXC; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float* f = fold + ((q * Q_MAXC + k) * 8 + e) * 64 + lane;
                        *f = (s == 0) ? dwacc[k][e] : *f + dwacc[k][e];
                    }
            }
            __syncthreads();
        }
        float* out = dw_partial + (long)blockIdx.x * d;
        for (int j = threadIdx.x; j < d; j += 64 * Q_NQ * Q_SLOTS) {
            const int c = j >> 3, e = j & 7;                   // column j = chunk c = (q * MAXC + k) * 64 + lane
            out[j] = fold[((c >> 6) * 8 + e) * 64 + (c & 63)];
        }
    }
}

// Variant S: the shipped kernel line by line, only the wave-wide sum swapped for the DPP one (isolates what the six ds_bpermute and their
// spilled index registers cost).
template <int MAXC>
__global__ __launch_bounds__(64 * RMSB_WAVES, MAXC <= 8 ? 4 : 2) void rmsnorm_bwd_s_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
    const float* __restrict__ rstd_in, const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
    float* __restrict__ dw_partial, long rows, int d, float* __restrict__ amax_parts) {
    __shared__ float fold[MAXC * 64 * 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long wave = (long)blockIdx.x * RMSB_WAVES + wv;
    const long nwaves = (long)gridDim.x * RMSB_WAVES;
    const int cpr = d >> 3;
    float dwacc[MAXC][8];
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) dwacc[k][e] = 0.f;
    for (long r = wave; r < rows; r += nwaves) {
        const float rstd = rstd_in[r];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            const int c = lane + 64 * k;
            if (c < cpr) {
                const u32x4 vd = *reinterpret_cast<const u32x4*>(dy + r * d + c * 8);
                const u32x4 vx = *reinterpret_cast<const u32x4*>(x + r * d + c * 8);
                const u32x4 vw = *reinterpret_cast<const u32x4*>(w + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = bf2f_lo(vd[e]), d1 = bf2f_hi(vd[e]);
                    const float x0 = bf2f_lo(vx[e]) * rstd, x1 = bf2f_hi(vx[e]) * rstd;
                    dot += d0 * bf2f_lo(vw[e]) * x0 + d1 * bf2f_hi(vw[e]) * x1;
                    dwacc[k][2 * e] += d0 * x0;
                    dwacc[k][2 * e + 1] += d1 * x1;
                }
            }
        }
        dot = wave_sum_dpp(dot) / (float)d;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            const int c = lane + 64 * k;
            if (c < cpr) {
                const u32x4 vd = *reinterpret_cast<const u32x4*>(dy + r * d + c * 8);
                const u32x4 vx = *reinterpret_cast<const u32x4*>(x + r * d + c * 8);
                const u32x4 vw = *reinterpret_cast<const u32x4*>(w + c * 8);
                u32x4 vr = {0u, 0u, 0u, 0u};
                if (dres) vr = *reinterpret_cast<const u32x4*>(dres + r * d + c * 8);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = bf2f_lo(vx[e]) * rstd, x1 = bf2f_hi(vx[e]) * rstd;
                    const float a = rstd * (bf2f_lo(vd[e]) * bf2f_lo(vw[e]) - x0 * dot) + bf2f_lo(vr[e]);
                    const float b = rstd * (bf2f_hi(vd[e]) * bf2f_hi(vw[e]) - x1 * dot) + bf2f_hi(vr[e]);
                    o[e] = pack_bf2(a, b);
                }
                *reinterpret_cast<u32x4*>(dx + r * d + c * 8) = o;
            }
        }
    }
    if (dw_partial) {
        // fold[k][e][lane]: lanes hit consecutive banks; waves add in index order
        for (int wq = 0; wq < RMSB_WAVES; ++wq) {
            if (wv == wq) {
#pragma unroll
                for (int k = 0; k < MAXC; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float* f = fold + (k * 8 + e) * 64 + lane;
                        *f = (wq == 0) ? dwacc[k][e] : *f + dwacc[k][e];
                    }
            }
            __syncthreads();
        }
        float* out = dw_partial + (long)blockIdx.x * d;
        for (int j = threadIdx.x; j < d; j += 64 * RMSB_WAVES) {
            const int c = j >> 3, e = j & 7;            // column j = chunk c (lane c & 63, k = c >> 6), element e
            out[j] = fold[((c >> 6) * 8 + e) * 64 + (c & 63)];
        }
    }
}


static unsigned short f2bf_host(float f) {
    unsigned int u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf2f_host(unsigned short h) {
    unsigned int u = (unsigned int)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const long rows = 5624;
    const int d = 4096;
    const size_t n = (size_t)rows * d;
    std::vector<unsigned short> h(n), hw(d);
    std::vector<float> hr(rows);
    srand(1);
    auto rnd = []() { return ((rand() & 0xffff) / 32768.f - 1.f) + ((rand() & 0xffff) / 32768.f - 1.f); };
    bf16_t *dy, *x, *dres, *dx0, *dx1, *w, *gw0, *gw1;
    float *rstd, *ws;
    CK(hipMalloc(&dy, n * 2)); CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&dres, n * 2)); CK(hipMalloc(&dx0, n * 2)); CK(hipMalloc(&dx1, n * 2));
    CK(hipMalloc(&w, d * 2)); CK(hipMalloc(&gw0, d * 2)); CK(hipMalloc(&gw1, d * 2)); CK(hipMalloc(&rstd, rows * 4));
    CK(hipMalloc(&ws, (size_t)1024 * d * 4));
    for (bf16_t* p : {dy, x, dres}) {
        for (size_t i = 0; i < n; ++i) h[i] = f2bf_host(rnd());
        CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < d; ++i) hw[i] = f2bf_host(1.f + 0.1f * rnd());
    CK(hipMemcpy(w, hw.data(), d * 2, hipMemcpyHostToDevice));
    for (long i = 0; i < rows; ++i) hr[i] = 0.5f + 0.25f * (rnd() + 2.f);
    CK(hipMemcpy(rstd, hr.data(), rows * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 50, PQ = 512;                              // 2 workgroups of 8 waves per CU at 128 VGPRs
    auto run_q = [&]() {
        hipLaunchKernelGGL(rmsnorm_bwd_q_kernel, dim3(PQ), dim3(64 * Q_NQ * Q_SLOTS), 0, 0, dy, x, w, rstd, dres, dx1, ws, rows, d);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(d, 64)), dim3(1024), 0, 0, ws, PQ, d, gw1, 0);
    };
    auto run_s = [&]() {
        const int P = mantis_rmsnorm_bwd_partials(rows);
        hipLaunchKernelGGL(rmsnorm_bwd_s_kernel<8>, dim3(P), dim3(64 * RMSB_WAVES), 0, 0, dy, x, w, rstd, dres, dx1, ws, rows, d, (float*)nullptr);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(d, 64)), dim3(1024), 0, 0, ws, P, d, gw1, 0);
    };
    // correctness first
    if (mantis_rmsnorm_bwd(dy, x, w, rstd, dres, dx0, gw0, 0, ws, rows, d, nullptr, nullptr) != MANTIS_OK) { printf("baseline launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    run_q();
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> a(n), b(n), ga(d), gb(d);
    CK(hipMemcpy(a.data(), dx0, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dx1, n * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ga.data(), gw0, d * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(gb.data(), gw1, d * 2, hipMemcpyDeviceToHost));
    double num = 0, den = 0, gnum = 0, gden = 0;
    size_t differ = 0;
    for (size_t i = 0; i < n; ++i) {
        const double p = bf2f_host(a[i]), qv = bf2f_host(b[i]);
        num += (p - qv) * (p - qv); den += p * p; differ += a[i] != b[i];
    }
    for (int i = 0; i < d; ++i) {
        const double p = bf2f_host(ga[i]), qv = bf2f_host(gb[i]);
        gnum += (p - qv) * (p - qv); gden += p * p;
    }
    printf("dx: rel L2 %.3e, %zu of %zu bf16 values differ (the dot is summed in another order);  dW: rel L2 %.3e\n", sqrt(num / den), differ, n,
           sqrt(gnum / gden));
    // variant S against the shipped kernel as well
    run_s();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(b.data(), dx1, n * 2, hipMemcpyDeviceToHost));
    num = den = 0;
    for (size_t i = 0; i < n; ++i) {
        const double p = bf2f_host(a[i]), qv = bf2f_host(b[i]);
        num += (p - qv) * (p - qv); den += p * p;
    }
    printf("variant S dx: rel L2 %.3e\n", sqrt(num / den));
    // timing
    float ms0 = 0, ms1 = 0, ms2 = 0;
    CK(hipEventRecord(e0));
    for (int i = 0; i < REP; ++i) mantis_rmsnorm_bwd(dy, x, w, rstd, dres, dx0, gw0, 0, ws, rows, d, nullptr, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms0, e0, e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < REP; ++i) run_q();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < REP; ++i) run_s();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms2, e0, e1));
    const double bytes = 4.0 * n * 2;                          // dy, x, dres in; dx out
    printf("shipped  (1 wave / row, two reads)      : %.1f us per call incl. dW reduce = %.2f TB/s\n", 1e3 * ms0 / REP, bytes / (1e-3 * ms0 / REP) / 1e12);
    printf("variant S (shipped + DPP wave sum)      : %.1f us per call incl. dW reduce = %.2f TB/s\n", 1e3 * ms2 / REP, bytes / (1e-3 * ms2 / REP) / 1e12);
    printf("variant Q (4 waves / row, regs, prefetch): %.1f us per call incl. dW reduce = %.2f TB/s\n", 1e3 * ms1 / REP, bytes / (1e-3 * ms1 / REP) / 1e12);
    return 0;
}
