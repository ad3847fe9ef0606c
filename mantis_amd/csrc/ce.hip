// Masked, shifted next-token cross-entropy: forward statistics + in-place logits gradient, gfx950.
//
// Replaces (reference): the loss tail of LlavaForConditionalGeneration.forward
//   /root/reference/mantis/models/mllava/modeling_llava.py:521-537  (shift, attention-mask filter, CrossEntropyLoss(mean,
//   ignore_index=-100)) and its autograd backward.  The shift/mask filter is resolved by the packing plan into
//   (row, target) pairs (pack.hip: ce_row / ce_tgt), so only rows that can carry a label reach the lm_head GEMM and the
//   [B, L, V] logits tensor of the reference is never materialised.
//
// One 1024-thread workgroup per row; V = 128258 bf16 logits = 256 KB per row, read twice (second pass from L2) and
// overwritten with the gradient.  Algorithmic bytes per row: 2*V read + 2*V write.  Softmax math in fp32.
#include "common.h"

#define CE_THREADS 1024

// count[0] = rows with a usable target (0 <= t < V): the mean's denominator.  count[1] = rows whose target is >= V: the
// reference (torch CrossEntropyLoss) raises on those; here they contribute neither loss nor gradient nor to the denominator and
// are reported so the host can raise.
__global__ void ce_count_kernel(const int* __restrict__ tgt, int R, int V, int* __restrict__ count) {
    __shared__ float red[16];
    float c = 0.f, bad = 0.f;
    for (int i = threadIdx.x; i < R; i += blockDim.x) {
        const int t = tgt[i];
        c += (t >= 0 && t < V) ? 1.f : 0.f;
        bad += (t >= V) ? 1.f : 0.f;
    }
    c = block_sum(c, red);
    bad = block_sum(bad, red);
    if (threadIdx.x == 0) { count[0] = (int)(c + 0.5f); count[1] = (int)(bad + 0.5f); }
}

// logits: [R, ld] bf16, overwritten with dlogits = (softmax - onehot) * gscale / count (zeros for ignored rows)
__global__ __launch_bounds__(CE_THREADS) void ce_fwd_bwd_kernel(bf16_t* __restrict__ logits, const int* __restrict__ tgt,
                                                                const int* __restrict__ count, float* __restrict__ row_loss,
                                                                float* __restrict__ row_lse, int V, long ld, float gscale,
                                                                int write_grad) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const int t = tgt[row];
    bf16_t* x = logits + (long)row * ld;
    const int nchunk = (V + 7) >> 3;
    if (t < 0 || t >= V) {
        if (threadIdx.x == 0) { row_loss[row] = 0.f; if (row_lse) row_lse[row] = 0.f; }
        if (write_grad) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int c = threadIdx.x; c < nchunk; c += CE_THREADS) *reinterpret_cast<u32x4*>(x + c * 8) = z;
        }
        return;
    }
    // pass 1: per-thread online (max, sum)
    float m = -INFINITY, s = 0.f;
    for (int c = threadIdx.x; c < nchunk; c += CE_THREADS) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + c * 8);
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[2 * e] = bf2f_lo(v[e]); f[2 * e + 1] = bf2f_hi(v[e]); }
        float cm = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (c * 8 + e < V) cm = fmaxf(cm, f[e]);
        const float nm = fmaxf(m, cm);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (c * 8 + e < V) acc += __expf(f[e] - nm);
        s = s * __expf(m - nm) + acc;
        m = nm;
    }
    const float gm = block_max(m, red);
    const float part = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
    const float gs = block_sum(part, red);
    const float lse = gm + logf(gs);
    const float xt = bf2f(x[t]);
    __syncthreads();  // every thread has read x[t] before pass 2 overwrites it
    if (threadIdx.x == 0) { row_loss[row] = lse - xt; if (row_lse) row_lse[row] = lse; }
    if (!write_grad) return;
    const float sc = gscale / (float)(*count);
    for (int c = threadIdx.x; c < nchunk; c += CE_THREADS) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + c * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j0 = c * 8 + 2 * e, j1 = j0 + 1;
            float g0 = (j0 < V) ? __expf(bf2f_lo(v[e]) - lse) : 0.f;
            float g1 = (j1 < V) ? __expf(bf2f_hi(v[e]) - lse) : 0.f;
            if (j0 == t) g0 -= 1.f;
            if (j1 == t) g1 -= 1.f;
            o[e] = pack_bf2(g0 * sc, g1 * sc);
        }
        *reinterpret_cast<u32x4*>(x + c * 8) = o;
    }
}

// loss = gscale_loss * sum(row_loss) / count  (fixed-order sum; 0/0 -> NaN like torch's mean over an empty set)
__global__ void ce_finish_kernel(const float* __restrict__ row_loss, const int* __restrict__ count, int R,
                                 float* __restrict__ loss_out, float lscale) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < R; i += blockDim.x) s += row_loss[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) loss_out[0] = lscale * s / (float)(*count);
}

extern "C" {

// logits [R, ld] bf16 (ld % 8 == 0, columns >= V are scratch); targets int32 (<0 = ignored).  Writes
//   loss_out[0] = loss_scale * mean_over_valid(lse - logit[target]),   count_out[0] = number of valid rows (0 <= t < V),
//   count_out[1] = number of rows with an out-of-range target (t >= V; the reference raises on those),
//   and (write_grad) overwrites logits with d(loss_scale_grad * mean CE)/dlogits.
int mantis_ce_fwd_bwd(void* logits, const int32_t* targets, int R, int V, int64_t ld, float grad_scale, float loss_scale,
                      int write_grad, float* row_loss_ws, float* row_lse_out, int32_t* count_out, float* loss_out,
                      void* stream) {
    if (R <= 0 || V <= 0 || ld % 8 || ld < V) return MANTIS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    MANTIS_LAUNCH(ce_count_kernel, dim3(1), dim3(1024), 0, s, targets, R, V, count_out);
    MANTIS_LAUNCH(ce_fwd_bwd_kernel, dim3(R), dim3(CE_THREADS), 0, s, (bf16_t*)logits, targets, count_out, row_loss_ws,
                       row_lse_out, V, (long)ld, grad_scale, write_grad);
    MANTIS_LAUNCH(ce_finish_kernel, dim3(1), dim3(1024), 0, s, row_loss_ws, count_out, R, loss_out, loss_scale);
    return mantis_check_launch();
}

}  // extern "C"
