// Multi-image token packing for gfx950: the integer plan and the bf16 row movers.
//
// Replaces (reference): LlavaForConditionalGeneration._merge_input_ids_with_image_features
//   /root/reference/mantis/models/mllava/modeling_llava.py:293-360 (plan + scatter),
//   the token-embedding gather at :427, and their autograd backward (index_put / embedding backward).
//
// Plan kernel: ONE workgroup of 1024 threads; every array is <= B*L ints (a few K elements), so the work is
// latency- not bandwidth-bound and a single CU doing block-wide scans in LDS beats ~10 ATen launches.
// Row kernels: pure HBM-bound 16 B/lane copies (algorithmic bytes: read B*T*d + I*N*d, write B*L*d bf16).
#include "common.h"
#include <climits>

#define IMGBIT (1 << 30)
#define PLAN_THREADS 1024

// inclusive block scan of one int per thread (1024 threads = 16 waves); returns inclusive prefix, *total = sum.
__device__ __forceinline__ int block_scan_incl(int v, int* lds, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    __syncthreads();
    if (lane == 63) lds[w] = x;
    __syncthreads();
    if (w == 0) {
        int t = (lane < 16) ? lds[lane] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            int y = __shfl_up(t, o, 64);
            if (lane >= o) t += y;
        }
        if (lane < 16) lds[16 + lane] = t;  // inclusive wave totals
    }
    __syncthreads();
    const int base = (w == 0) ? 0 : lds[16 + w - 1];
    *total = lds[16 + 15];
    return x + base;
}

// status[0] = 0 ok | 1 image-slot count mismatch (reference raises ValueError, modeling_llava.py:347-351)
//             | 2 L passed by the host differs from max_b k[b]*(N-1)+T (:301)
// status[1] = number of image slots found, status[2] = number of <image> tokens, status[3] = left_padding
// FIXED = false: the reference's slot search (:344-345: unwritten rows minus the first pad_b of them), which mis-places the image
//   rows of a right-padded sample holding fewer images than the batch maximum (masked in the reference by the
//   bs = 1 assert of processing_llava.py:277-285).  FIXED = true (`fix_unequal_counts`): appendix A's index-only
//   formulation -- image j of sample b occupies [p[b,t_j] - (N-1), p[b,t_j]] whatever the padding side, every other unwritten slot
//   stays padding; identical to the reference for equal counts, and sample by sample to the reference run at B = 1.
template <bool FIXED>
__global__ __launch_bounds__(PLAN_THREADS) void pack_plan_kernel(
    const long* __restrict__ ids, const long* __restrict__ attn, const long* __restrict__ labels, int B, int T, int N,
    int num_images, long IMG, long PAD, long IGN, int L, int* __restrict__ src, long* __restrict__ out_mask,
    long* __restrict__ out_labels, long* __restrict__ out_pos, int* __restrict__ kmask, int* __restrict__ text_pos,
    int* __restrict__ img_slot, int* __restrict__ ce_row, int* __restrict__ ce_tgt, int* __restrict__ status) {
    __shared__ int lds[64];
    __shared__ int sh_flag, sh_kmax, sh_base;
    const int tid = threadIdx.x;
    if (tid == 0) { sh_flag = 0; sh_kmax = 0; sh_base = 0; }
    __syncthreads();
    // :296 left_padding = not any(ids[:, -1] == pad)
    for (int b = tid; b < B; b += PLAN_THREADS)
        if (ids[(long)b * T + T - 1] == PAD) atomicOr(&sh_flag, 1);
    // :298-301 k[b], kmax
    int total_img_tokens = 0;
    for (int b = 0; b < B; ++b) {
        int c = 0;
        for (int t = tid; t < T; t += PLAN_THREADS) c += (ids[(long)b * T + t] == IMG);
        int tot;
        block_scan_incl(c, lds, &tot);
        if (tid == 0) sh_kmax = max(sh_kmax, tot);
        total_img_tokens += tot;
        __syncthreads();
    }
    __syncthreads();
    const int left = !sh_flag;
    const int Lc = sh_kmax * (N - 1) + T;
    if (tid == 0) { status[2] = total_img_tokens; status[3] = left; status[1] = 0; status[0] = 0; }
    if (Lc != L) {
        // host/device disagreement on the merged length: leave every plan output in a SAFE state (all padding, nothing to
        // gather, no CE rows) so that downstream gather/scatter kernels cannot index out of bounds; the host sees status 2.
        if (tid == 0) status[0] = 2;
        for (long i = tid; i < (long)B * L; i += PLAN_THREADS) {
            src[i] = -1; out_mask[i] = 0; out_labels[i] = IGN; out_pos[i] = 1; kmask[i] = 0;
        }
        for (long i = tid; i < (long)B * T; i += PLAN_THREADS) { text_pos[i] = -1; ce_row[i] = -1; ce_tgt[i] = -100; }
        for (long i = tid; i < (long)num_images * N; i += PLAN_THREADS) img_slot[i] = -1;
        return;
    }
    const int perT = (T + PLAN_THREADS - 1) / PLAN_THREADS;
    const int perL = (L + PLAN_THREADS - 1) / PLAN_THREADS;
    const long total_rows = (long)num_images * N;
    for (int b = 0; b < B; ++b) {
        const long* idb = ids + (long)b * T;
        // :309 p = cumsum(m*(N-1)+1) - 1 over a contiguous chunk per thread
        const int t0 = tid * perT, t1 = min(T, t0 + perT);
        int local = 0;
        for (int t = t0; t < t1; ++t) local += (idb[t] == IMG) ? N : 1;
        int tot;
        int incl = block_scan_incl(local, lds, &tot);
        const int pad_b = L - 1 - (tot - 1);  // :310 nb_image_pad
        const int shift = left ? pad_b : 0;   // :311-312
        int img_before = 0, k_b = 0;          // FIXED: <image> tokens of this sample in front of this thread's chunk / in the sample
        if constexpr (FIXED) {
            int li = 0;
            for (int t = t0; t < t1; ++t) li += (idb[t] == IMG);
            const int ii = block_scan_incl(li, lds, &k_b);
            img_before = ii - li;
        }
        // :316-326 initialise the merged row
        for (int p = tid; p < L; p += PLAN_THREADS) {
            src[(long)b * L + p] = -1;
            out_mask[(long)b * L + p] = 0;
            out_labels[(long)b * L + p] = IGN;
        }
        __syncthreads();
        // :338-341 scatter text
        int run = incl - local;
        for (int t = t0; t < t1; ++t) {
            const bool is_img = idb[t] == IMG;
            run += is_img ? N : 1;
            const int p = run - 1 + shift;
            const int flat = b * T + t;
            if (!is_img) {
                src[(long)b * L + p] = t;
                const long a = attn[(long)b * T + t];
                out_mask[(long)b * L + p] = a;
                const long lab = labels ? labels[(long)b * T + t] : IGN;
                out_labels[(long)b * L + p] = lab;
                text_pos[flat] = p;
                // row I: the hidden state at p-1 predicts this token (shift + mask filter, :523-527)
                if (p >= 1) {
                    ce_row[flat] = b * L + p - 1;
                    ce_tgt[flat] = (a != 0 && lab != IGN) ? (int)lab : -100;
                } else {
                    ce_row[flat] = -1;
                    ce_tgt[flat] = -100;
                }
            } else {
                text_pos[flat] = -1;
                ce_row[flat] = -1;
                ce_tgt[flat] = -100;
                if constexpr (FIXED) {
                    // the N slots that end at p are this image's, feature rows in batch-major, in-sample order (:353)
                    const long g0 = ((long)sh_base + img_before) * N;
                    for (int e = 0; e < N; ++e) {
                        const long g = g0 + e;
                        const int q = p - (N - 1) + e;
                        if (g < total_rows) {
                            src[(long)b * L + q] = IMGBIT | (int)g;
                            img_slot[g] = b * L + q;
                        }
                        out_mask[(long)b * L + q] = 1;  // :354
                    }
                    ++img_before;
                }
            }
        }
        __syncthreads();
        const int p0 = tid * perL, p1 = min(L, p0 + perL);
        if constexpr (FIXED) {
            if (tid == 0) sh_base += k_b;
        } else {
            // :344-345 image slots = unwritten rows minus the first pad_b of them; :353 filled in row-major order
            int lu = 0;
            for (int p = p0; p < p1; ++p) lu += (src[(long)b * L + p] == -1);
            int totu;
            int inclu = block_scan_incl(lu, lds, &totu);
            int rank = inclu - lu;  // unwritten rows before p0
            const int base = sh_base;
            for (int p = p0; p < p1; ++p) {
                if (src[(long)b * L + p] == -1) {
                    if (rank >= pad_b) {
                        const long g = (long)base + (rank - pad_b);
                        if (g < total_rows) {
                            src[(long)b * L + p] = IMGBIT | (int)g;
                            img_slot[g] = b * L + p;
                        }
                        out_mask[(long)b * L + p] |= 1;  // :354
                    }
                    ++rank;
                }
            }
            __syncthreads();
            if (tid == 0) sh_base = base + max(0, totu - pad_b);
        }
        __syncthreads();
        // :355 position_ids = cumsum(mask) - 1, 1 where mask == 0
        int lm = 0;
        for (int p = p0; p < p1; ++p) lm += (int)out_mask[(long)b * L + p];
        int totm;
        int inclm = block_scan_incl(lm, lds, &totm);
        long c = inclm - lm;
        for (int p = p0; p < p1; ++p) {
            const long mk = out_mask[(long)b * L + p];
            c += mk;
            out_pos[(long)b * L + p] = (mk == 0) ? 1 : (c - 1);
            kmask[(long)b * L + p] = mk != 0;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const long found = FIXED ? (long)sh_base * N : (long)sh_base;     // FIXED counts <image> tokens, the search counts slots
        status[1] = (int)found;
        if (found != total_rows) status[0] = 1;
    }
}

// out[row] = embW[ids[t]] | feats[r] | 0, 16 B per lane; d % 8 == 0.
__global__ void pack_rows_fwd_kernel(const int* __restrict__ src, const long* __restrict__ ids,
                                     const bf16_t* __restrict__ embW, const bf16_t* __restrict__ feats,
                                     bf16_t* __restrict__ out, int B, int T, int L, int d, long V) {
    const int cpr = d >> 3;
    const long total = (long)B * L * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cpr;
        const int c = (int)(i - row * cpr);
        const int s = src[row];
        u32x4 v = {0u, 0u, 0u, 0u};
        if (s >= 0) {
            const bf16_t* p;
            if (s & IMGBIT) {
                p = feats + (long)(s & (IMGBIT - 1)) * d;
            } else {
                const int b = (int)(row / L);
                long id = ids[(long)b * T + s];
                id = id < 0 ? 0 : (id >= V ? V - 1 : id);
                p = embW + id * d;
            }
            v = *reinterpret_cast<const u32x4*>(p + c * 8);
        }
        *reinterpret_cast<u32x4*>(out + row * d + c * 8) = v;
    }
}

// out[r] = idx[r] >= 0 ? in[idx[r]] : 0
__global__ void gather_rows_kernel(const bf16_t* __restrict__ in, const int* __restrict__ idx, bf16_t* __restrict__ out,
                                   long nrows, int d) {
    const int cpr = d >> 3;
    const long total = nrows * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cpr;
        const int c = (int)(i - row * cpr);
        const int s = idx[row];
        u32x4 v = {0u, 0u, 0u, 0u};
        if (s >= 0) v = *reinterpret_cast<const u32x4*>(in + (long)s * d + c * 8);
        *reinterpret_cast<u32x4*>(out + row * d + c * 8) = v;
    }
}

// out[idx[r]] = in[r] for idx[r] >= 0 (indices are unique by construction -> plain stores, no atomics)
__global__ void scatter_rows_kernel(const bf16_t* __restrict__ in, const int* __restrict__ idx, bf16_t* __restrict__ out,
                                    long nrows, int d) {
    const int cpr = d >> 3;
    const long total = nrows * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cpr;
        const int c = (int)(i - row * cpr);
        const int s = idx[row];
        if (s >= 0)
            *reinterpret_cast<u32x4*>(out + (long)s * d + c * 8) = *reinterpret_cast<const u32x4*>(in + row * d + c * 8);
    }
}

// Deterministic embedding backward.  chain kernel: for text token i, leader[i] = no earlier text token with the same
// id; next[i] = the next later text token with the same id (or -1).  n = B*T is ~1K, O(n^2) id compares from L2.
__global__ void embed_chain_kernel(const long* __restrict__ ids, const int* __restrict__ text_pos, int n,
                                   int* __restrict__ leader, int* __restrict__ next) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (text_pos[i] < 0) { leader[i] = 0; next[i] = -1; return; }
    const long id = ids[i];
    int lead = 1;
    for (int j = 0; j < i; ++j)
        if (ids[j] == id && text_pos[j] >= 0) { lead = 0; break; }
    int nx = -1;
    for (int j = i + 1; j < n; ++j)
        if (ids[j] == id && text_pos[j] >= 0) { nx = j; break; }
    leader[i] = lead;
    next[i] = nx;
}

// one workgroup per token; leaders sum their chain in token order (fp32) and add it to gradW[id] (bf16, +=).
__global__ void embed_grad_kernel(const bf16_t* __restrict__ dout, const long* __restrict__ ids,
                                  const int* __restrict__ text_pos, const int* __restrict__ leader,
                                  const int* __restrict__ next, bf16_t* __restrict__ gradW, int T, int L, int d, long V,
                                  int accumulate) {
    const int i = blockIdx.x;
    if (!leader[i]) return;
    const long id = ids[i];
    if (id < 0 || id >= V) return;
    for (int c = threadIdx.x; c < (d >> 3); c += blockDim.x) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (accumulate) {
            const u32x4 g = *reinterpret_cast<const u32x4*>(gradW + id * d + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[2 * e] = bf2f_lo(g[e]); acc[2 * e + 1] = bf2f_hi(g[e]); }
        }
        for (int j = i; j >= 0; j = next[j]) {
            const int b = j / T;
            const long row = (long)b * L + text_pos[j];
            const u32x4 v = *reinterpret_cast<const u32x4*>(dout + row * d + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[2 * e] += bf2f_lo(v[e]); acc[2 * e + 1] += bf2f_hi(v[e]); }
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(acc[2 * e], acc[2 * e + 1]);
        *reinterpret_cast<u32x4*>(gradW + id * d + c * 8) = o;
    }
}

static inline int row_grid(long total_chunks) {
    long g = (total_chunks + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}


// ---- packed samples (sample packing, /root/reference/mantis/train/data.py:1546-1671: several samples in ONE row) --------------
// Runs after pack_plan_kernel on the same arrays.  seg[B,T]: sample index of every input token (non-decreasing along a row).
// Produces, per merged position p of a row:   kstart[p] = first merged position of p's sample (the O(L) form of the reference's
// block-diagonal 4-D mask: a causal query attends keys >= kstart only),   qend[p] = one past the last position of p's sample,
// and rewrites   position_ids   to restart at 0 in every sample (data.py:1641-1648; pads keep 1 as in modeling_llava.py:355) and
// ce_row / ce_tgt so that the first token of a sample is NOT predicted from the last hidden state of the previous sample.
// A token's span in the merged row is [tok_start, tok_end]: one slot for text, num_patches slots for <image>; spans are recovered
// from an inclusive scan of the span lengths (+ the row's left-padding shift, which is 0 for packed rows).  One workgroup per row.
__device__ __forceinline__ int block_scan_max_incl(int v, int* lds) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x = max(x, y);
    }
    __syncthreads();
    if (lane == 63) lds[w] = x;
    __syncthreads();
    if (w == 0) {
        int t = (lane < 16) ? lds[lane] : INT_MIN;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int y = __shfl_up(t, o, 64);
            if (lane >= o) t = max(t, y);
        }
        if (lane < 16) lds[16 + lane] = t;
    }
    __syncthreads();
    return (w == 0) ? x : max(x, lds[16 + w - 1]);
}

__global__ __launch_bounds__(PLAN_THREADS) void pack_segments_kernel(
    const long* __restrict__ ids, const int* __restrict__ seg, const long* __restrict__ out_mask, int B, int T, int N, long IMG, int L,
    long* __restrict__ out_pos, int* __restrict__ ce_row, int* __restrict__ ce_tgt, int* __restrict__ kstart,
    int* __restrict__ qend, int* __restrict__ ws /* [B, L] scratch */) {
    __shared__ int lds[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long* idb = ids + (long)b * T;
    const int* sgb = seg + (long)b * T;
    int* wsb = ws + (long)b * L;
    int* ksb = kstart + (long)b * L;
    int* qeb = qend + (long)b * L;
    const int perT = (T + PLAN_THREADS - 1) / PLAN_THREADS, perL = (L + PLAN_THREADS - 1) / PLAN_THREADS;
    // 1. span ends by an inclusive scan of span lengths; the row's shift = L - total (left padding, modeling_llava.py:310-312)
    const int t0 = tid * perT, t1 = min(T, t0 + perT);
    int local = 0;
    for (int t = t0; t < t1; ++t) local += (idb[t] == IMG) ? N : 1;
    int tot;
    const int incl = block_scan_incl(local, lds, &tot);
    const int shift = L - tot;          // packed rows: 0.  (right-padded rows never reach here: the host refuses pad + segments)
    for (int p = tid; p < L; p += PLAN_THREADS) wsb[p] = 0;
    __syncthreads();
    // 2. mark the first merged position of every sample with (position + 1); sample starts also lose their CE row
    int run = incl - local;
    for (int t = t0; t < t1; ++t) {
        const int len = (idb[t] == IMG) ? N : 1;
        const int start = run + shift;
        run += len;
        if (t == 0 || sgb[t] != sgb[t - 1]) {
            if (start >= 0 && start < L) wsb[start] = start + 1;
            ce_row[(long)b * T + t] = -1;
            ce_tgt[(long)b * T + t] = -100;
        }
    }
    __syncthreads();
    // 3. kstart = running maximum of the marks (minus 1); positions before the first mark (left padding) belong to "sample" 0
    const int p0 = tid * perL, p1 = min(L, p0 + perL);
    int lm = 0;
    for (int p = p0; p < p1; ++p) lm = max(lm, wsb[p]);
    const int pre = block_scan_max_incl(lm, lds);      // inclusive over chunks
    // exclusive prefix for this chunk = max over earlier chunks: recompute from the inclusive value of the previous thread
    __shared__ int chunkmax[PLAN_THREADS];
    chunkmax[tid] = pre;
    __syncthreads();
    int cur = tid == 0 ? 0 : chunkmax[tid - 1];
    for (int p = p0; p < p1; ++p) {
        cur = max(cur, wsb[p]);
        ksb[p] = cur > 0 ? cur - 1 : 0;
    }
    __syncthreads();
    // 4. qend[p] = start of the next sample (or L): the smallest mark position > p  ->  reverse running minimum
    int ln = L;
    for (int p = p1 - 1; p >= p0; --p) if (wsb[p] > 0 && p > 0) ln = p;      // smallest marked position inside the chunk (p = 0 is no "next")
    // suffix minimum over later chunks: small serial pass by warp 0 over the 1024 chunk values (once per step, latency irrelevant)
    chunkmax[tid] = ln;
    __syncthreads();
    if (tid == 0) {
        int m = L;
        for (int i = PLAN_THREADS - 1; i >= 0; --i) { const int v = chunkmax[i]; chunkmax[i] = m; m = min(m, v); }   // exclusive suffix min
    }
    __syncthreads();
    int nxt = chunkmax[tid];
    for (int p = p1 - 1; p >= p0; --p) {
        qeb[p] = nxt;
        if (wsb[p] > 0 && p > 0) nxt = p;
    }
    __syncthreads();
    // 5. position ids restart per sample: cumsum(mask) - cumsum(mask)[kstart - 1] - 1; pads keep 1 (modeling_llava.py:355)
    int lc = 0;
    for (int p = p0; p < p1; ++p) lc += (int)out_mask[(long)b * L + p];
    int totm;
    const int inclm = block_scan_incl(lc, lds, &totm);
    int c = inclm - lc;
    for (int p = p0; p < p1; ++p) {
        c += (int)out_mask[(long)b * L + p];
        wsb[p] = c;                                    // inclusive cumsum of the merged attention mask
    }
    __syncthreads();
    for (int p = p0; p < p1; ++p) {
        const int ks = ksb[p];
        const int base = ks > 0 ? wsb[ks - 1] : 0;
        out_pos[(long)b * L + p] = out_mask[(long)b * L + p] == 0 ? 1 : (long)(wsb[p] - base - 1);
    }
}

extern "C" {

// mode 0 = the reference's placement (mantis_pack_plan), 1 = `fix_unequal_counts` (see pack_plan_kernel)
int mantis_pack_plan_mode(const int64_t* input_ids, const int64_t* attention_mask, const int64_t* labels, int B, int T,
                          int num_patches, int num_images, int64_t image_token_index, int64_t pad_token_id,
                          int64_t ignore_index, int L, int mode, int32_t* src, int64_t* out_mask, int64_t* out_labels,
                          int64_t* out_pos, int32_t* kmask, int32_t* text_pos, int32_t* img_slot, int32_t* ce_row,
                          int32_t* ce_tgt, int32_t* status, void* stream) {
    if (B <= 0 || T <= 0 || L < T || num_patches <= 0 || num_images < 0 || (mode != 0 && mode != 1)) return MANTIS_EINVAL;
    if ((long)num_images * num_patches >= IMGBIT) return MANTIS_EUNSUPPORTED;
#define PLAN_ARGS dim3(1), dim3(PLAN_THREADS), 0, (hipStream_t)stream, (const long*)input_ids, (const long*)attention_mask, \
                  (const long*)labels, B, T, num_patches, num_images, (long)image_token_index, (long)pad_token_id, (long)ignore_index, L, \
                  src, (long*)out_mask, (long*)out_labels, (long*)out_pos, kmask, text_pos, img_slot, ce_row, ce_tgt, status
    if (mode == 1) MANTIS_LAUNCH(pack_plan_kernel<true>, PLAN_ARGS);
    else MANTIS_LAUNCH(pack_plan_kernel<false>, PLAN_ARGS);
#undef PLAN_ARGS
    return mantis_check_launch();
}

int mantis_pack_plan(const int64_t* input_ids, const int64_t* attention_mask, const int64_t* labels, int B, int T,
                     int num_patches, int num_images, int64_t image_token_index, int64_t pad_token_id,
                     int64_t ignore_index, int L, int32_t* src, int64_t* out_mask, int64_t* out_labels,
                     int64_t* out_pos, int32_t* kmask, int32_t* text_pos, int32_t* img_slot, int32_t* ce_row,
                     int32_t* ce_tgt, int32_t* status, void* stream) {
    return mantis_pack_plan_mode(input_ids, attention_mask, labels, B, T, num_patches, num_images, image_token_index, pad_token_id,
                                 ignore_index, L, 0, src, out_mask, out_labels, out_pos, kmask, text_pos, img_slot, ce_row, ce_tgt, status,
                                 stream);
}

int mantis_pack_segments(const int64_t* input_ids, const int32_t* segment_ids, const int64_t* merged_mask, int B, int T,
                         int num_patches, int64_t image_token_index, int L, int64_t* out_pos, int32_t* ce_row, int32_t* ce_tgt,
                         int32_t* kstart, int32_t* qend, int32_t* workspace, void* stream) {
    if (B <= 0 || T <= 0 || L < T || num_patches <= 0) return MANTIS_EINVAL;
    MANTIS_LAUNCH(pack_segments_kernel, dim3(B), dim3(PLAN_THREADS), 0, (hipStream_t)stream, (const long*)input_ids,
                       segment_ids, (const long*)merged_mask, B, T, num_patches, (long)image_token_index, L, (long*)out_pos, ce_row,
                       ce_tgt, kstart, qend, workspace);
    return mantis_check_launch();
}

int mantis_pack_rows_fwd(const int32_t* src, const int64_t* input_ids, const void* embed_weight, const void* image_features,
                         void* out, int B, int T, int L, int d, int64_t vocab, void* stream) {
    if (d % 8) return MANTIS_EUNSUPPORTED;
    const long chunks = (long)B * L * (d / 8);
    MANTIS_LAUNCH(pack_rows_fwd_kernel, dim3(row_grid(chunks)), dim3(256), 0, (hipStream_t)stream, src,
                       (const long*)input_ids, (const bf16_t*)embed_weight, (const bf16_t*)image_features, (bf16_t*)out,
                       B, T, L, d, (long)vocab);
    return mantis_check_launch();
}

int mantis_gather_rows(const void* in, const int32_t* idx, void* out, int64_t nrows, int d, void* stream) {
    if (d % 8) return MANTIS_EUNSUPPORTED;
    if (nrows == 0) return MANTIS_OK;
    MANTIS_LAUNCH(gather_rows_kernel, dim3(row_grid(nrows * (d / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, idx, (bf16_t*)out, (long)nrows, d);
    return mantis_check_launch();
}

int mantis_scatter_rows(const void* in, const int32_t* idx, void* out, int64_t nrows, int d, void* stream) {
    if (d % 8) return MANTIS_EUNSUPPORTED;
    if (nrows == 0) return MANTIS_OK;
    MANTIS_LAUNCH(scatter_rows_kernel, dim3(row_grid(nrows * (d / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, idx, (bf16_t*)out, (long)nrows, d);
    return mantis_check_launch();
}

// grad_weight[ids[b,t]] (+)= sum over text tokens with that id of dmerged[b, text_pos[b,t]]   (deterministic order)
int mantis_embed_grad(const void* dmerged, const int64_t* input_ids, const int32_t* text_pos, int32_t* leader_ws,
                      int32_t* next_ws, void* grad_weight, int B, int T, int L, int d, int64_t vocab, int accumulate,
                      void* stream) {
    if (d % 8) return MANTIS_EUNSUPPORTED;
    const int n = B * T;
    MANTIS_LAUNCH(embed_chain_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const long*)input_ids,
                       text_pos, n, leader_ws, next_ws);
    MANTIS_LAUNCH(embed_grad_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dmerged,
                       (const long*)input_ids, text_pos, leader_ws, next_ws, (bf16_t*)grad_weight, T, L, d, (long)vocab,
                       accumulate);
    return mantis_check_launch();
}

}  // extern "C"
