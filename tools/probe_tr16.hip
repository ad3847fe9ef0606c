// Hardware probe (run on the GPU box): prints the lane -> element mapping of ds_read_b64_tr_b16 and of the
// v_mfma_f32_32x32x16_bf16 operand/accumulator layout, to pin the assumptions documented in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe_tr(unsigned short* out, int stride_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // 16-lane group g reads a [4][16] block: lane s of the group points at row (s>>2), columns 4*(s&3)..
    const int g = l >> 4, s = l & 15;
    const int addr = g * 1024 + (s >> 2) * stride_elems + (s & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
__global__ void probe_mfma(float* out) {
    // A[i][k] = i + 0.01*k? use exactly representable: A[i][k] = (i==I0 && k==K0), sweep via B = identity-like
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = (l >> 5) * 8 + e;            // assumed: lane (i = l&31, kg = l>>5) holds A[i][8kg+e]
        a[e] = (__bf16)(float)((l & 31) + 1);      // A[i][k] = i+1  (all k)
        b[e] = (__bf16)((k == 3) ? (float)((l & 31) * 2 + 1) : 0.f);   // B[k][j] = (k==3) * (2j+1)
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];   // expect D[i][j] = (i+1)*(2j+1)
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {16, 64}) {
        probe_tr<<<1, 64>>>(d, stride);
        std::vector<unsigned short> h(256);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("tr16 stride=%d\n", stride);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    float* f; hipMalloc(&f, 64 * 16 * 4);
    probe_mfma<<<1, 64>>>(f);
    std::vector<float> hf(1024);
    hipMemcpy(hf.data(), f, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            if (hf[l * 16 + r] != (float)((i + 1) * (2 * j + 1))) ++bad;
        }
    printf("mfma 32x32x16 layout check: %d mismatches (0 = assumed layout holds)\n", bad);
    if (bad) for (int l = 0; l < 64; l += 9) { printf("  lane %d:", l); for (int r = 0; r < 16; ++r) printf(" %g", hf[l * 16 + r]); printf("\n"); }
    return 0;
}
