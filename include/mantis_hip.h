/* libmantis_hip.so -- C-ABI of the MI355X (gfx950) hot path for Mantis' multi-image LLaVA training step.
 *
 * Boundary.  The reference (TIGER-AI-Lab/Mantis) is 100 % Python; its hot path -- transformers.Trainer.training_step
 * driving mantis/models/mllava/modeling_llava.py:364-549 -- reaches native code only through torch/ATen, flash-attn and
 * NCCL.  There is no reference FFI to mirror, so this header DEFINES the operator boundary a maintainer binds from Python
 * (ctypes): plain device pointers, sizes, element strides and a hipStream_t passed as void*.  No torch
 * types, no device allocation inside (every workspace is passed in by the caller), no mutable global state, re-entrant per stream.
 * Every function returns 0 on success, MANTIS_EINVAL (-1) for invalid arguments, MANTIS_EUNSUPPORTED (-2) for a shape or
 * alignment the kernels do not cover, MANTIS_ELAUNCH (-3) if the HIP launch failed.  All tensors are bf16 (raw 16-bit)
 * unless stated, row-major, 16-byte aligned; "ld*" are row strides in ELEMENTS.
 *
 * Each entry cites the reference interface (file:line) it replaces; `HF:` = the transformers copy the reference runs on.
 */
#ifndef MANTIS_HIP_H
#define MANTIS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MANTIS_OK 0
#define MANTIS_EINVAL (-1)
#define MANTIS_EUNSUPPORTED (-2)
#define MANTIS_ELAUNCH (-3)

/* ---- multi-image token packing --------------------------------------------------------------------------------------
 * replaces LlavaForConditionalGeneration._merge_input_ids_with_image_features
 *   /root/reference/mantis/models/mllava/modeling_llava.py:293-360 (integer plan, bit-exact) and the loss row filter
 *   of :521-527.  L = max_b (#<image> in row b) * (num_patches-1) + T is computed by the caller (:301).
 * src[B,L]: -1 padding slot | t (text token column) | (1<<30)|r (image-feature row r).
 * status[0]: 0 ok, 1 image-slot count mismatch (reference raises ValueError :347-351), 2 L mismatch (every plan
 * output is then left in a safe all-padding state: nothing to gather, no CE rows);
 * status[1] = slots found, status[2] = #<image> tokens, status[3] = left_padding (:296). */
int mantis_pack_plan(const int64_t* input_ids, const int64_t* attention_mask, const int64_t* labels /*nullable*/, int B,
                     int T, int num_patches, int num_images, int64_t image_token_index, int64_t pad_token_id,
                     int64_t ignore_index, int L, int32_t* src, int64_t* out_mask, int64_t* out_labels, int64_t* out_pos,
                     int32_t* kmask, int32_t* text_pos, int32_t* img_slot, int32_t* ce_row, int32_t* ce_tgt,
                     int32_t* status, void* stream);
/* The same plan with a placement mode.  mode 0 = mantis_pack_plan (the reference's slot search, including its mis-placement of the
 * image rows of a right-padded sample that holds fewer images than the batch maximum: modeling_llava.py:343-345, hidden in the
 * reference by the bs = 1 assert of processing_llava.py:277-285).  mode 1 = `fix_unequal_counts`: image j of sample b
 * occupies [p[b,t_j] - (N-1), p[b,t_j]] whatever the padding side (SURVEY appendix A's index-only formulation); identical to mode 0
 * when every sample holds the same number of images, and sample by sample identical to the reference run at B = 1. */
int mantis_pack_plan_mode(const int64_t* input_ids, const int64_t* attention_mask, const int64_t* labels /*nullable*/, int B,
                          int T, int num_patches, int num_images, int64_t image_token_index, int64_t pad_token_id,
                          int64_t ignore_index, int L, int mode, int32_t* src, int64_t* out_mask, int64_t* out_labels,
                          int64_t* out_pos, int32_t* kmask, int32_t* text_pos, int32_t* img_slot, int32_t* ce_row,
                          int32_t* ce_tgt, int32_t* status, void* stream);
/* Packed samples (/root/reference/mantis/train/data.py:1546-1671, PackingDataset.pack_batch: several samples concatenated into one
 * row with a block-diagonal 4-D attention mask and position ids restarting per sample).  Runs after mantis_pack_plan on the same
 * arrays; segment_ids int32 [B,T] = sample index of every input token (non-decreasing along a row, rows not padded).  Writes
 * kstart / qend int32 [B,L] (first / one-past-last merged position of each position's sample: the O(L) form of the block-diagonal
 * mask, consumed by mantis_attn_fwd / mantis_attn_bwd), rewrites out_pos to restart at 0 per sample and removes the CE row of every
 * sample's first token (no prediction across a sample boundary).  workspace: int32 [B,L]. */
int mantis_pack_segments(const int64_t* input_ids, const int32_t* segment_ids, const int64_t* merged_mask, int B, int T,
                         int num_patches, int64_t image_token_index, int L, int64_t* out_pos, int32_t* ce_row, int32_t* ce_tgt,
                         int32_t* kstart, int32_t* qend, int32_t* workspace, void* stream);
/* merged[b,p,:] = embed_weight[ids[b,src]] | image_features[src & ~(1<<30)] | 0     (modeling_llava.py:427,338,353) */
int mantis_pack_rows_fwd(const int32_t* src, const int64_t* input_ids, const void* embed_weight, const void* image_features,
                         void* out, int B, int T, int L, int d, int64_t vocab, void* stream);
/* out[r] = idx[r] >= 0 ? in[idx[r]] : 0   /   out[idx[r]] = in[r]  (unique idx) -- autograd of the index_put at :338,:353 */
int mantis_gather_rows(const void* in, const int32_t* idx, void* out, int64_t nrows, int d, void* stream);
int mantis_scatter_rows(const void* in, const int32_t* idx, void* out, int64_t nrows, int d, void* stream);
/* embedding backward of modeling_llava.py:427 (nn.Embedding), deterministic: grad_weight[id] (+)= sum dmerged[b,text_pos] */
int mantis_embed_grad(const void* dmerged, const int64_t* input_ids, const int32_t* text_pos, int32_t* leader_ws,
                      int32_t* next_ws, void* grad_weight, int B, int T, int L, int d, int64_t vocab, int accumulate,
                      void* stream);

/* ---- norms: HF:models/llama/modeling_llama.py:53-67 (LlamaRMSNorm) + autograd; HF:models/siglip/modeling_siglip.py:325-358 */
/* amax_parts (nullable; rmsnorm_fwd, rmsnorm_bwd, swiglu_fwd): mantis_fp8_quantize_ws_floats() floats <- per-workgroup maxima of
 * |output|, for the fp8 quantiser that consumes the output next (it then skips its own amax pass) */
int mantis_rmsnorm_fwd(const void* x, const void* weight, void* y, float* rstd /*[rows], nullable*/, int64_t rows, int d,
                       float eps, float* amax_parts, void* stream);
int mantis_rmsnorm_bwd_partials(int64_t rows); /* workspace rows: floats needed = partials * d */
int mantis_rmsnorm_bwd(const void* dy, const void* x, const void* weight, const float* rstd, const void* dres /*nullable*/,
                       void* dx, void* grad_weight /*nullable*/, int accumulate, float* workspace, int64_t rows, int d,
                       float* amax_parts, void* stream);
int mantis_layernorm_fwd(const void* x, const void* weight, const void* bias, void* y, int64_t rows, int d, float eps,
                         void* stream);

/* ---- activations: HF:models/llama/modeling_llama.py:163-176 (SwiGLU), modeling_llava.py:106-118 (GELU), siglip MLP acts.
 * kind: 0 gelu(erf) 1 gelu(tanh) 2 quick_gelu 3 silu.  gate_up is [M, 2I] = (gate | up). */
int mantis_swiglu_fwd(const void* gate_up, void* out, int64_t M, int I, int64_t ld_gate_up, float* amax_parts, void* stream);
int mantis_swiglu_bwd(const void* dact, const void* gate_up, void* dgate_up, int64_t M, int I, int64_t ld_gate_up,
                      void* stream);
int mantis_act_fwd(const void* x, void* y, int64_t n, int kind, void* stream);
int mantis_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int kind, void* stream);
int mantis_add(const void* a, const void* b, void* y, int64_t n, void* stream);
int mantis_colsum_partials(int64_t M);
int mantis_colsum(const void* x, void* grad, int accumulate, float* workspace, int64_t M, int N, int64_t ld, void* stream);

/* ---- RoPE: HF:models/llama/modeling_llama.py:113-160 on the position_ids of modeling_llava.py:355 */
int mantis_rope_table(const int64_t* position_ids, const float* inv_freq, void* cos_out, void* sin_out, int64_t R,
                      int half_dim, void* stream);
/* Sectioned table (Qwen2-VL): frequency j uses row section_of_freq[j] of position_ids[S, R].  Multimodal RoPE
 * HF:models/qwen2_vl/modeling_qwen2_vl.py:156-170,207-213 (S = 3) and the vision tower's 2-D rotary embedding :239-248 (S = 2). */
int mantis_rope_table_sections(const int64_t* position_ids, const float* inv_freq, const int32_t* section_of_freq, void* cos_out,
                               void* sin_out, int64_t R, int half_dim, void* stream);
int mantis_rope_apply(void* x, const void* cos_tab, const void* sin_tab, int64_t R, int nheads, int head_dim, int64_t ld,
                      int backward, void* stream);
/* out[b][h][c][r] = in[b][h][r][c], zero for R <= r < Rpad (operand layouts for dX/dW GEMMs and attention) */
int mantis_transpose(const void* in, void* out, int R, int C, int Rpad, int64_t ld_in, int64_t ld_out, int nb, int nh,
                     int64_t in_stride_b, int64_t in_stride_h, int64_t out_stride_b, int64_t out_stride_h, void* stream);

/* ---- GEMM: every nn.Linear of the path (see csrc/gemm.hip).  C = epi(A[M,K] . B[N,K]^T).
 * flags: 1 bias | act<<1 (1 gelu-erf, 2 gelu-tanh, 3 quick-gelu) | 16 residual add | 32 accumulate into C
 *        | 64 SwiGLU backward fused behind dact = A.B^T: residual = [gate | up][M, 2N], C = [dgate | dup][M, 2N]
 *        | bits 8-11 tile variant (0 = auto) | 4096 A is K-major ([K,M], row stride lda) | 8192 B is K-major ([K,N]):
 *        dX = dY.W uses B K-major (the weight as stored), dW = dY^T.X uses both K-major -- no transposed copies.
 *        | bits 16-27 CU budget this launch is planned for (0 = default; see mantis_gemm_cu_budget)
 *        | 32768 the launch shares the GPU with long-running kernels of other queues (RCCL collectives of a data-parallel step): the 176-row
 *          kernel then runs one tile per workgroup instead of persistent workgroups (same results; a persistent workgroup that waits for a
 *          CU someone else holds walks its whole static tile list late).  mantis_gemm_bf16_nt_fused: bit 7 (128) of `variant`.
 *        | 16384 remainder tiles of a ring16 launch reduced by their last arriver inside the GEMM kernel instead of by the finishing
 *          kernel (round-4 behaviour; same results bit for bit; tests / A-B measurements; process default: MANTIS_GEMM_SK_FINISH=0).
 * Remainder rounds: tiles of an incomplete last round of 256x256 tiles are split along K (deterministic); the ring16 kernels' split
 * units store fp32 partial slabs into the workspace and a second, small kernel on the same stream (gemm_ring16_finish_kernel) sums
 * them and runs the epilogue on all compute units -- no tickets, no spinning, nothing that can deadlock under CU masks.
 * Tile variants: 1 = 128x128 generic kernel, 2 = 256x256 generic kernel, 12 = 256x256 ring kernel with 8 waves x (128 x 64) of
 * 32x32x16 MFMAs, 13 = 256x256 ring16 kernel with 4 waves x (128 x 128) of 16x16x32 MFMAs, 14 = 256x256 ring16 kernel with 8 waves x
 * (128 x 64) of 16x16x32 MFMAs, 15 = the 176x256 kernel (round 6, csrc/gemm176.hip: 4 waves x (176 x 64) of 16x16x32 MFMAs, A row-major only -- forward and
 * dX layouts --, whole tiles without a K split, PERSISTENT workgroups that prefetch their next tile's first K-steps under the epilogue when a
 * launch has more tiles than CUs; MANTIS_EUNSUPPORTED with flag 4096).  Automatic choice (variant 0): a fitted cost model picks the 128x128 kernel or a ring kernel
 * (mantis_gemm_pick_variant: 1 or 12 = "a ring kernel"); among the ring kernels, 14 for every K-major layout and for short per-CU K
 * walks, 13 for row-major (NT) operands with >= 400 K-steps per CU (rounds x K/64), unless the process was started with
 * MANTIS_GEMM_RING = 12 | 13 | 14 (forces that ring kernel wherever a ring kernel is chosen; A/B measurements); where a ring kernel is chosen, A is
 * row-major and a predicted-time model (rounds x (K-steps x loop time + fixed cost) per tiling, for the launch's CU budget) favours it by >= 3 %,
 * the 176x256 kernel: it turns the 352- / 528-tile grids of M = 5624 into whole rounds (MANTIS_GEMM_176 = 0 never | 1 model (default) | 2 always;
 * MANTIS_GEMM_176P = 0: one tile per workgroup).  The ring kernels address operands through 32-bit buffer descriptors: an
 * operand of >= 4 GiB is routed to the generic kernel when the variant is auto and returns MANTIS_EUNSUPPORTED when a ring variant (or
 * the SwiGLU epilogue) was forced. */
int mantis_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                        const void* bias, const void* residual, int64_t ldr, int flags, void* workspace /*nullable*/,
                        int64_t workspace_bytes, void* stream);
/* bytes of caller-owned, 256-B aligned, ZERO-INITIALISED device workspace a launch of C[M,N] over K may need (0 = none; only the
 * ring kernel's deterministic split-K remainder round uses it; every launch leaves it zeroed-for-reuse, so one buffer per stream
 * serves all stream-ordered launches).  M = N = K = 0: the largest requirement of any shape on the current device (~128 MB on 256 CUs since
 * round 5: 2 x #CU slabs of 256 KiB for the balanced remainder round; size buffers with this call, never with a constant). */
int mantis_gemm_workspace_bytes(int M, int N, int K);
/* CU budget of the GEMM tile scheduler (rounds of tiles, the K split of an incomplete last round, the tile variant).  PER CALL since round 5:
 * bits 16-27 of mantis_gemm_bf16_nt's / _sumsq's `flags` and of mantis_gemm_bf16_nt_fused's `variant` (0 = the default: the environment
 * constant MANTIS_GEMM_CUS, read once, else the whole device) -- no process-wide state, two models / gradient reducers in one process do
 * not interfere.  This entry is a pure query: the budget a launch asking for `cus` plans for (cus > 0: clamped to [8, #CU of the device];
 * cus <= 0: the default).  For data-parallel runs: every RCCL channel is a workgroup that cannot share a CU with a 160-KiB-LDS ring
 * workgroup, so with C channels active plan for #CU - C.  Deterministic for a given budget; a different budget changes which tiles are
 * K-split, i.e. their fp32 summation order (results agree to bf16 rounding).  The split-K workspace size does not depend on it. */
int mantis_gemm_cu_budget(int cus);
/* Forward projections with a two-column epilogue fused in (16x16x32 ring kernels 13 / 14 and the 176x256 kernel 15, NT layout).
 *   mode 1 (SwiGLU): B = [gate | up] weight [2 I, K], N = 2 I: C = A . B^T [M, 2 I] exactly as mantis_gemm_bf16_nt writes it AND
 *                    aux0 = silu(gate) * up [M, I] (bf16, row stride aux_ld) exactly as mantis_swiglu_fwd computes it (replaces
 *                    transformers LlamaMLP's act_fn(gate_proj(x)) * up_proj(x), modeling_llama.py:163-176, as ONE launch)
 *   mode 2 (RoPE):   q|k|v projection, heads of 128 columns, optional bias: columns [0, aux_n) leave with the rotary embedding applied
 *                    (aux0 = cos, aux1 = sin, bf16 [M, 64], row stride aux_ld), exactly as mantis_rope_apply(backward = 0) would
 *                    rotate them in a second pass (apply_rotary_pos_emb, modeling_llama.py:138-160)
 * variant: bits 0-3 0 = per-shape choice, 13 / 14 / 15 forced | 64 = in-kernel remainder reduction (flag 16384 of mantis_gemm_bf16_nt) | bits
 * 16-27 CU budget (as in mantis_gemm_bf16_nt's flags).  MANTIS_EUNSUPPORTED for shapes outside the fused kernels' conditions (N % 256, I % 128,
 * aux_n % 128, 16-B alignment, operands < 4 GiB): the caller then issues the two launches. */
int mantis_gemm_bf16_nt_fused(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                              const void* bias /*nullable*/, int mode, void* aux0, const void* aux1 /*mode 2*/, int64_t aux_ld, int aux_n,
                              int variant, void* workspace, int64_t workspace_bytes, void* stream);

/* Weight-gradient GEMM that also leaves the squared norm of its result.  C (+)= A . B^T exactly as mantis_gemm_bf16_nt computes it
 * (flags: 32 accumulate | 4096 / 8192 K-major operands | bits 8-11 variant 13 / 14, or 0 = automatic | 16384 | bits 16-27 CU budget) and, for every 256 x 256 output
 * tile t (the kernel's tile order), tile_sumsq[t] = the sum of the squares of the bf16 values stored in that tile -- after the
 * accumulation when flag 32 is set.  tile_sumsq holds cdiv(M,256) * cdiv(N,256) floats; every entry is written exactly once.  Replaces,
 * for the weight gradients, the pass of clip_grad_norm_ (HF trainer.py:2535-2545) over the gradient: the optimizer sums the tile values
 * (mantis_sum_f32).  Deterministic.  Ring16 kernels only: N % 256 == 0, ldc % 8 == 0, 16-B aligned pointers, no bias / activation /
 * residual; otherwise MANTIS_EUNSUPPORTED (-2) and the caller runs mantis_gemm_bf16_nt followed by mantis_sumsq. */
int mantis_gemm_bf16_nt_sumsq(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K, int flags,
                              float* tile_sumsq, void* workspace, int64_t workspace_bytes, void* stream);
/* Two weight-gradient GEMMs in ONE launch (round 6): C1[M1,N1] (+)= A1^T . B1 and C2[M2,N2] (+)= A2^T . B2 with A given [K, M], B given [K, N] (the
 * TN form of mantis_gemm_bf16_nt_sumsq: dW = dY^T . X reads both activations as stored) for two problems that share K.  Replaces two launches
 * whose 256 x 256 grids each end in a K-split remainder round + finishing pass by one grid of whole tiles -- dW(down_proj) 4096 x 14336 (896 tiles)
 * + dW(q|k|v) 6144 x 4096 (384 tiles) of a Llama-3-8B / Mistral-7B decoder layer = 1280 tiles = 5.0 rounds on 256 CUs -- in one XCD-contiguous
 * tile order over the union; no workspace.  flags: 32 accumulate | bits 16-27 CU budget.  ts1 / ts2: per-tile sums of squares of what was stored
 * (cdiv(M,256) * cdiv(N,256) floats each, as mantis_gemm_bf16_nt_sumsq) or both NULL.  Results: the 8-wave ring16 kernel's whole-tile
 * arithmetic (= variant 14 of mantis_gemm_bf16_nt on a shape without a K-split remainder, bit for bit); against a K-split launch of the same
 * shape the remainder tiles' fp32 summation order differs (bf16 rounding).  MANTIS_EUNSUPPORTED: N % 256, unaligned or >= 4 GiB operands.
 * mantis_gemm_tn_pair_wins: 1 when the cost model predicts the paired launch >= 3 % ahead of two launches for `cus` CUs (<= 0: default). */
int mantis_gemm_bf16_tn_pair(const void* A1, int64_t lda1, const void* B1, int64_t ldb1, void* C1, int64_t ldc1, int M1, int N1, float* ts1,
                             const void* A2, int64_t lda2, const void* B2, int64_t ldb2, void* C2, int64_t ldc2, int M2, int N2, float* ts2,
                             int K, int flags, void* stream);
int mantis_gemm_tn_pair_wins(int M1, int N1, int M2, int N2, int K, int cus);
/* tile family the auto heuristic (flags bits 8-11 == 0) picks: 12 = a 256x256 ring kernel (which of 12 / 13 / 14: see above), 1 = the 128x128 generic kernel */
int mantis_gemm_pick_variant(int M, int N, int K);
/* the same for a launch planned for `cus` compute units (cus <= 0: the default budget) */
int mantis_gemm_pick_variant_cus(int M, int N, int K, int cus);
/* The remainder-round plan of a ring16 launch of C[M,N] over K planned for `cus` compute units (<= 0: default), for tests and tools (no device
 * work).  out[0..7] = {tiles, full tiles, remainder tiles, S of the equal split (1 = none), balanced units (0 = equal split), tail slots,
 * remainder workgroups, K-steps}; then 4 ints per remainder workgroup in grid order, while they fit `cap`: {remainder tile (-1 = empty tail
 * slot), first K-step, end K-step, slab}.  Returns the number of ints of the full description (tests/test_gemm_remainder_plan.py checks
 * that every K-step of every remainder tile is covered exactly once, slabs are unique and tails sit on their range's XCD). */
int mantis_gemm_remainder_plan(int M, int N, int K, int cus, int32_t* out, int cap);

/* ---- fp8 linears (BASELINE configs[4] "fp8 MFMA"): an accelerated variant of the bf16 nn.Linear of the Qwen2
 * decoder (HF:models/qwen2_vl/modeling_qwen2_vl.py:453-466,501-504); the reference has no fp8, tolerance is stated against its bf16 /
 * fp32 path.  OCP e4m3 (fmt 0, max 448) / e5m2 (fmt 1, max 57344), per-tensor just-in-time scaling.  See csrc/gemm_fp8.hip.
 * quantize: x bf16 [rows, cols] (stride ld elements, cols % 16 == 0) -> q [rows, cols] (stride ldq bytes) and, if qt != NULL, the
 * transposed copy qt [cols, rows_pad] (stride ldt bytes, rows_pad = rows rounded up to 16, zero tail);
 * state float[3] <- {amax, FMAX / amax, amax / FMAX}; workspace: mantis_fp8_quantize_ws_floats() floats.
 * gemm: C[M,N] bf16 = epi(dequant_a * dequant_b * A8[M,K] . B8[N,K]^T), lda / ldb bytes, K % 16 == 0, fmt_a 0|1, B e4m3;
 * flags 1 bias | 16 residual | 32 accumulate | variant << 8 (0 auto, 1 128x128, 2 256x256). */
int mantis_fp8_quantize_ws_floats(void);
int mantis_fp8_quantize(const void* x, int64_t rows, int cols, int64_t ld, int fmt, void* q, int64_t ldq, void* qt, int64_t ldt,
                        float* state, float* workspace, const float* amax_in /*nullable: max|x| already taken by x's producer*/,
                        int amax_in_count /*1 (mantis_gemm_fp8_dx_swiglu) or mantis_fp8_quantize_ws_floats() per-workgroup maxima*/,
                        void* stream);
int mantis_gemm_fp8_nt(const void* A8, int64_t lda, const void* B8, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                       const float* dequant_a, const float* dequant_b, int fmt_a, const void* bias, const void* residual, int64_t ldr,
                       int flags, void* stream);
/* Opt-in finer scaling (set_precision("fp8_rowwise")): q (nullable) with one scale per ROW of x, row_dequant float[rows]; qt (nullable)
 * with one scale per COLUMN of x, col_dequant float[cols]; workspace rows + cols floats.  mantis_gemm_fp8_nt flag 128: dequant_a /
 * dequant_b are such vectors (float[M] / float[N]) and the epilogue multiplies out[m, n] by dequant_a[m] * dequant_b[n]. */
int mantis_fp8_quantize_2d(const void* x, int64_t rows, int cols, int64_t ld, int fmt, void* q, int64_t ldq, float* row_dequant, void* qt,
                           int64_t ldt, float* col_dequant, float* workspace, void* stream);
/* dgu[M, 2N] = swiglu_backward(dequant * A8[M,K] . B8[N,K]^T, gate_up[M, 2N]): dX of down_proj with the SwiGLU backward (autograd of
 * HF:models/qwen2_vl/modeling_qwen2_vl.py:453-466) in the epilogue; amax_out (nullable) float[1] <- max |dgu| for the next quantiser. */
int mantis_gemm_fp8_dx_swiglu(const void* A8, int64_t lda, const void* B8, int64_t ldb, void* dgu, int64_t ld_dgu, int M, int N, int K,
                              const float* dequant_a, const float* dequant_b, int fmt_a, const void* gate_up, int64_t ld_gu,
                              float* amax_out, void* stream);

/* ---- attention: HF:models/llama/modeling_llama.py:191-214,262-276; HF:models/siglip/modeling_siglip.py:227-247 */
/* kmask int32 [B,L] (nullable): 1 = key may be attended (key padding).  kstart int32 [B,L] (nullable): packed samples -- query q
 * attends keys >= kstart[b,q] only (the start of its own sample; non-decreasing in q), i.e. the block-diagonal mask of
 * /root/reference/mantis/train/data.py:1627-1638 in O(L) form. */
int mantis_attn_fwd(const void* Q, const void* K, const void* V, const int32_t* kmask, const int32_t* kstart, void* O, float* LSE,
                    int B, int L, int H, int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, int causal,
                    void* stream);
int mantis_attn_dsum(const void* dO, const void* O, float* Dsum, int B, int L, int H, int hd, int64_t ldo, void* stream);
/* Cross attention (non-causal): Lq query rows per batch entry (Q, O, dO, dQ: [B*Lq, ...]; LSE, Dsum fp32 [B, H, Lq]) over Lk keys (K, V, dK,
 * dV: [B*Lk, ...]; kmask int32 [B, Lk] or NULL).  The Idefics2 perceiver resampler: 64 latent queries over concat[context, latents]
 * (/root/reference/mantis/models/idefics2/modeling_idefics2.py:812-912).  Same kernels, head dims, strides and status codes as
 * mantis_attn_fwd / mantis_attn_bwd; the backward's workspace is 2 * B*Lk*H*hd bf16 when H > Hkv. */
int mantis_attn_fwd_cross(const void* Q, const void* K, const void* V, const int32_t* kmask, void* O, float* LSE, int B, int Lq, int Lk,
                          int H, int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, void* stream);
int mantis_attn_bwd_cross(const void* Q, const void* K, const void* V, const void* O, const void* dO, const int32_t* kmask,
                          const float* LSE, float* Dsum, void* dQ, void* dK, void* dV, void* workspace, int B, int Lq, int Lk, int H,
                          int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ld_out, int64_t ldo, int64_t lddq,
                          int64_t lddk, int64_t lddv, float scale, void* stream);
/* workspace: 2*B*L*H*hd bf16 (per-query-head dK/dV partials, reduced over the GQA group) when
 * mantis_attn_bwd_needs_workspace(H, Hkv, hd) says so, else unused/NULL: not for H == Hkv, and not for hd 128 with H = 4 Hkv
 * (Llama-3), which always runs the GQA-aware dK/dV kernel (one workgroup per (64-key block, KV head) walks the group's query heads
 * as two streams, partials meet in LDS).  Other group sizes (even, or odd >= 5; Qwen2-7B's 7:1) run that kernel when the grid is
 * tiny or at least two rounds deep, and the per-query-head path otherwise -- they are handed the workspace either way.
 * O (forward output, row stride ld_out) optional: if given, Dsum = rowsum(dO * O) is computed inside the dQ kernel and written to
 * Dsum ([B,H,L] fp32); if NULL, Dsum must already hold it (mantis_attn_dsum).
 * kstart / qend (int32 [B,L], both or neither, nullable): segment bounds for packed samples (data.py:1609-1671 block-diagonal mask):
 * kstart[b,q] = first key position query q attends, qend[b,k] = one past the last query position that attends key k. */
int mantis_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const int32_t* kmask,
                    const int32_t* kstart, const int32_t* qend, const float* LSE, float* Dsum, void* dQ, void* dK, void* dV,
                    void* workspace, int B, int L, int H, int Hkv, int hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ld_out,
                    int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, float scale, int causal, void* stream);
int mantis_attn_bwd_needs_workspace(int H, int Hkv, int hd);

/* ---- loss: modeling_llava.py:521-537 (shift + mask filter resolved by mantis_pack_plan into ce_row / ce_tgt).
 * count_out is int32[2]: [0] rows with 0 <= target < V (the mean's denominator), [1] rows with target >= V (torch's
 * CrossEntropyLoss raises on those; here they are excluded from loss, gradient and denominator and reported). */
int mantis_ce_fwd_bwd(void* logits, const int32_t* targets, int R, int V, int64_t ld, float grad_scale, float loss_scale,
                      int write_grad, float* row_loss_ws, float* row_lse_out /*nullable*/, int32_t* count_out,
                      float* loss_out, void* stream);

/* ---- ViT front end: modeling_llava.py:434-435 + HF Siglip/CLIP VisionEmbeddings; feature select modeling_llava.py:460-461 */
int mantis_im2col(const float* pixels, void* patches, int I, int C, int H, int W, int P, int Kp, void* stream);
/* Qwen2-VL PatchEmbed input (HF:models/qwen2_vl/modeling_qwen2_vl.py:268-274): fp32 flattened patches -> bf16 rows zero-padded to Kp */
int mantis_cast_pad_rows(const float* in, void* out, int64_t rows, int K, int64_t ld_in, int Kp, void* stream);
int mantis_vit_assemble(const void* patch_out, const void* pos_emb, const void* cls_emb /*nullable*/, void* out, int I, int N,
                        int d, void* stream);
int mantis_drop_cls(const void* in, void* out, int I, int N, int d, void* stream);
/* NaViT image preparation on the device (replaces the host loop of /root/reference/mantis/models/idefics2/modeling_idefics2.py:1636-1639
 * padding-image detection, :1653-1658 pixel mask -> patch mask, :190-210 bucketised position ids), one workgroup per image slot:
 * pixels fp32 [n, C, H, W] (16-B aligned, C*H*W % 4 == 0), pixel_mask uint8 [n, H, W] or NULL (= all attended); bucket int32
 * [tab_n, tab_n]: bucket[m][j] = bucketize(arange(0, 1 - 1e-6, 1 / m), boundaries, right = True)[j], built by the caller with the
 * reference's own float arithmetic.  Outputs int32: real[n] (1 = has a non-zero pixel), patch_mask[n, (H/P)*(W/P)], pos_ids[same],
 * status[n] (1 = attended patches do not form an nh x nw grid, where the reference raises). */
int mantis_navit_prepare(const float* pixels, const uint8_t* pixel_mask, int n_images, int C, int H, int W, int P, int side,
                         const int32_t* bucket, int tab_n, int32_t* real, int32_t* patch_mask, int32_t* pos_ids, int32_t* status,
                         void* stream);

/* ---- optimizer: HF:trainer.py:1785-1796 + clip :2535-2545, AdamW over flat buffers */
int mantis_adamw(void* param_bf16, const void* grad_bf16, float* master, float* exp_avg, float* exp_avg_sq, int64_t n,
                 float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                 const float* grad_scale_dev /*nullable: multiply grads by *grad_scale_dev (clip)*/, void* stream);
/* The same step with the fp32 master stored SPLIT: its round-to-nearest-even upper half is the bf16 parameter itself (param_bf16, read AND
 * written), master_lo (uint16 [n]) holds its low 16 bits, and the sign bit of exp_avg_sq (a value that is never negative) records the one case
 * those 32 bits leave open -- an exact tie that rounded UP to the even neighbour.  26 instead of 28 bytes of HBM traffic and 2 instead of 4
 * bytes of state per parameter; bit-identical to mantis_adamw on the joined master (DeepSpeed's fp32 master weights,
 * HF:trainer.py:1785-1796).  exp_avg_sq as stored: |x| is Adam's second moment. */
int mantis_adamw_split(void* param_bf16, const void* grad_bf16, void* master_lo, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                       const float* grad_scale_dev, void* stream);
/* master_out[i] = the fp32 master of element i, from (param_bf16, master_lo, sign of exp_avg_sq): checkpoints, tests */
int mantis_master_join(const void* param_bf16, const void* master_lo, const float* exp_avg_sq, float* master_out, int64_t n, void* stream);
/* the inverse (resume from a checkpoint that holds fp32 masters): param_bf16 = bf16(master), master_lo = its low half, tie bit -> sign of
 * exp_avg_sq (its magnitude is kept) */
int mantis_master_split(const float* master, void* param_bf16, void* master_lo, float* exp_avg_sq, int64_t n, void* stream);
/* out[0] (+)= sum of x[0 .. n): one workgroup, fixed summation order (the tile partials of mantis_gemm_bf16_nt_sumsq) */
int mantis_sum_f32(const float* x, int64_t n, float* out, int accumulate, void* stream);
/* partials[r] = sum of squares of the bf16 elements x[off_len[2r] .. off_len[2r] + off_len[2r+1]) for r < n_ranges (device array of
 * int64 pairs, offsets and lengths multiples of 8), one workgroup per range: the many small gradient ranges no fused dW GEMM covers. */
int mantis_sumsq_ranges(const void* x_bf16, const int64_t* off_len, int n_ranges, float* partials, void* stream);
int mantis_sumsq_partials(int64_t n);
int mantis_sumsq(const void* x_bf16, int64_t n, float* partials_ws, float* out /*[1], += */, int accumulate, void* stream);
int mantis_clip_scale(const float* sumsq, float max_norm, float* scale_out, float* norm_out, void* stream);

/* library / device info */
/* ---- CU-partitioned streams: the HBM-bound optimizer pass on one share of the compute units, the next batch's frozen vision tower on
 * the rest (software pipelining across HF:trainer.py's training_step -> optimizer.step boundary; same arithmetic, only earlier) */
int mantis_stream_create_cu_mask(int first_cu, int n_cus, void** stream_out);
/* A stream with the highest (level < 0), default (0) or lowest (level > 0) hardware-queue priority the device offers (hipStreamCreateWithPriority,
 * non-blocking); levels_out (nullable) receives {greatest, least}.  Work off the critical path (the backward's weight-gradient GEMMs) queued
 * on a lowest-priority stream is dispatched where the critical path leaves compute units idle.  The caller owns the stream. */
int mantis_stream_create_priority(int level, void** stream_out, int* levels_out);
int mantis_stream_destroy(void* stream);

int mantis_version(void);

#ifdef __cplusplus
}
#endif
#endif
