"""fp8 variant of decoder.py's layer loop (BASELINE.json configs[4]: "Qwen2-VL-7B ... fp8 MFMA"): the seven
linears of every decoder layer (q|k|v, o, gate|up, down: forward, dX and dW) run on the fp8 MFMA GEMM of csrc/gemm_fp8.hip; norms, RoPE,
attention, SwiGLU, residual stream, lm_head and the loss stay exactly as in decoder.py (bf16 / fp32).

Recipe (per tensor, just-in-time scales, no history): activations and weights in e4m3, output gradients in e5m2; the quantiser hands
back the transposed copy together with the row-major one, so forward (X8 . W8^T), dX (dY8 . W8T^T) and dW (dY8T . X8T^T) are all the one
"NT" kernel.  The backward keeps the TRANSPOSED fp8 activations (1 B/element) instead of decoder.py's bf16 n1 / n2 / a.
Opt-in `rowwise` recipe (set_precision("fp8_rowwise")): the row-major copy carries one scale per row and the transposed copy one per
column of the source, so both operands of each of the three GEMMs are scaled per row and the epilogue multiplies by sa[m] * sb[n].

The reference has no fp8 (its linears are bf16 nn.Linear); parity is therefore stated in two steps (tests/gpu_checks.py fp8_*):
HIP == the oracle's exact restatement of this arithmetic (oracle/ops_ref.py fp8_quantize / gemm_fp8_nt), and that restatement vs the
fp32 oracle of the reference within the fp8 tolerance written in the tests."""
import os

from . import decoder as D

# A/B switch (measurements only): 0 = dX(down_proj), SwiGLU backward and the amax pass as three launches
FUSE_SWIGLU_BWD = os.environ.get("MANTIS_FP8_FUSE_SWIGLU", "1") == "1"
PRODUCER_AMAX = os.environ.get("MANTIS_FP8_PRODUCER_AMAX", "1") == "1"      # 0 = every quantiser runs its own amax pass

E4M3, E5M2 = 0, 1
_NAMES = ("qkv", "o", "gu", "down")


class Fp8Weights:
    """e4m3 copies (row-major for the forward, transposed for dX) of the decoder's linear weights.  `refresh()` drops them; they are
    re-quantised lazily, layer by layer, at first use."""

    def __init__(self, lm, rowwise=False):
        self.lm = lm
        self.rowwise = rowwise
        self.cache = {}

    def refresh(self):
        self.cache.clear()

    def get(self, K, i, name):
        key = (i, name)
        w = self.cache.get(key)
        if w is None:
            w = self.cache[key] = _quant(K, self.lm["layers"][i][name], E4M3, True, None, self.rowwise)
        return w


def _quant(K, x, fmt, transposed, amax, rowwise):
    if rowwise:
        return K.fp8_quantize(x, fmt, transposed=transposed, rowwise=True)
    return K.fp8_quantize(x, fmt, transposed=transposed, amax=amax)


def _lin(K, xq, wq, bias=None, residual=None):
    return K.gemm_fp8_nt(xq.q, xq.dequant, wq.q, wq.dequant, E4M3, bias=bias, residual=residual, rowwise=xq.rowwise)


def layer_forward(K, lw, w8, i, tc, x, B, L, cos, sin, kmask, kstart, scale, parts, keep):
    """One fp8 decoder block: x -> (x_out, what its backward reads).  `keep`: make the transposed quantised copies the weight-gradient GEMMs
    read.  Deterministic (amax is a maximum, the GEMMs' K-split order is fixed): a second run on the same x reproduces the first bit for
    bit -- what activation checkpointing relies on."""
    H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    eps = tc.rms_norm_eps
    rw = w8.rowwise
    n1, rstd1 = K.rmsnorm_fwd(x, lw["ln1"], eps, amax_parts=parts)
    n1q = _quant(K, n1, E4M3, keep, parts, rw)
    qkv = _lin(K, n1q, w8.get(K, i, "qkv"), bias=lw.get("qkv_b"))
    K.rope_apply_(qkv, cos, sin, H + Hkv, hd)
    o, lse = K.attn_fwd(qkv, B, L, H, Hkv, hd, kmask, scale, True, kstart=kstart)
    oq = _quant(K, o, E4M3, keep, None, rw)
    x_mid = _lin(K, oq, w8.get(K, i, "o"), residual=x)
    n2, rstd2 = K.rmsnorm_fwd(x_mid, lw["ln2"], eps, amax_parts=parts)
    n2q = _quant(K, n2, E4M3, keep, parts, rw)
    gu = _lin(K, n2q, w8.get(K, i, "gu"))
    a = K.swiglu_fwd(gu, amax_parts=parts)
    aq = _quant(K, a, E4M3, keep, parts, rw)
    x_out = _lin(K, aq, w8.get(K, i, "down"), residual=x_mid)
    if keep:
        for t in (n1q, oq, n2q, aq):
            t.q = None                                  # the backward reads only the transposed copies
    return x_out, (x, rstd1, qkv, o, lse, x_mid, rstd2, gu, n1q, oq, n2q, aq)


def decoder_forward(K, lm, w8, tc, x, B, L, kmask, compute_grads=True, record=None, rope=None, kstart=None, checkpoint=False):
    """Same contract as decoder.decoder_forward (rope tables given by the caller; `checkpoint`: keep only every layer's input and run
    `layer_forward` again in the backward)."""
    hd = tc.head_dim
    scale = hd ** -0.5
    cos, sin = rope
    saved = []
    # producer-side amax: RMSNorm and SwiGLU take the maximum |value| of what they write, so the quantiser that follows skips its own
    # pass over the tensor (one scratch buffer, reused: producer and consumer are adjacent on the stream)
    parts = K.amax_parts_buffer(x.device) if PRODUCER_AMAX and not w8.rowwise else None
    for i in range(tc.num_hidden_layers):
        keep_copies = compute_grads and not checkpoint
        x_out, keep = layer_forward(K, lm["layers"][i], w8, i, tc, x, B, L, cos, sin, kmask, kstart, scale, parts, keep_copies)
        if compute_grads:
            saved.append((x,) if checkpoint else keep)
        del keep
        x = x_out
        if record is not None:
            record[f"llm_layer{i}_out"] = x.view(B, L, -1)
    return x, dict(saved=saved, cos=cos, sin=sin, scale=scale, parts=parts)


def _dw(K, dyq, xq, grad, acc):
    """grad[out, in] (+)= dY^T . X over the (zero-padded) token axis: both operands are the quantiser's transposed copies."""
    if grad is not None:
        K.gemm_fp8_nt(dyq.qt, dyq.dequant_t, xq.qt, xq.dequant_t, E5M2, out=grad, accumulate=acc, rowwise=dyq.rowwise)


def _dx(K, dyq, wq):
    return K.gemm_fp8_nt(dyq.q, dyq.dequant, wq.qt, wq.dequant_t, E5M2, rowwise=dyq.rowwise)


def decoder_backward(K, lm, w8, grads, grads_layers, tc, ctx, hctx, plan, B, L, kmask, accumulate=False, on_bucket_ready=None,
                     kstart=None, qend=None):
    """Same contract as decoder.decoder_backward: head + loss backward (bf16, shared), then the fp8 layer loop."""
    H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    acc = accumulate
    saved, cos, sin, scale = ctx["saved"], ctx["cos"], ctx["sin"], ctx["scale"]
    dx = D.head_backward(K, lm, grads, hctx, plan, B, L, acc, on_bucket_ready)
    rw = w8.rowwise
    # The weight-gradient GEMMs are off the critical path (nothing in the backward reads a dW).  The fp8 GEMM has no K split for the tiles of an
    # incomplete last round (Qwen2-7B: N = 3584 = 14 tiles of 256 -- 448-tile grids = 1.75 rounds on 256 CUs on five of the twelve shapes), so
    # on a side stream (hip_ops.side_stream) the dW launches queue behind the quantiser that made their operands, and their workgroups take
    # the compute units the dX chain's launches leave idle (one 160-KiB workgroup per CU either way).  Bucket hooks follow them onto that
    # stream, as in decoder.decoder_backward; the caller's stream joins at the end.  Bit-identical.  Default: the lowest-priority queue on a
    # single GPU (same box, Qwen2-VL-7B step: 332.7 -> 327.9 ms; a plain stream 328.7), the calling stream under data parallelism (the
    # bucket exchanges then start exactly where they are signalled); MANTIS_DW_STREAM = 0 | 1 | low overrides either.
    side = K.side_stream(default="low" if on_bucket_ready is None else "0")

    def off_path(fn, *inputs):
        if side is None:
            fn()
        else:
            side.run(fn, *inputs)

    def dw(dyq, xq, grad):
        if grad is not None:
            off_path(lambda: _dw(K, dyq, xq, grad, acc), dyq.qt, dyq.col_dequant, dyq.state, xq.qt, xq.col_dequant, xq.state)

    def bucket(key):
        if on_bucket_ready is not None:
            off_path(lambda: on_bucket_ready(key))
    parts_dx = K.amax_parts_buffer(dx.device) if PRODUCER_AMAX and not rw else None
    parts_mid = K.amax_parts_buffer(dx.device) if PRODUCER_AMAX and not rw else None
    dx_amax = None                     # the top layer's dx comes from the loss head (no producer-side amax); below: rmsnorm_bwd's
    for i in reversed(range(tc.num_hidden_layers)):
        lw = lm["layers"][i]
        lg_ = grads_layers[i]
        entry = saved.pop()
        if len(entry) == 1:          # activation checkpointing: only the layer input was kept -- run the layer's forward again
            _, entry = layer_forward(K, lw, w8, i, tc, entry[0], B, L, cos, sin, kmask, kstart, scale, ctx.get("parts"), True)
        x_in, rstd1, qkv, o, lse, x_mid, rstd2, gu, n1q, oq, n2q, aq = entry
        del entry
        dxq = _quant(K, dx, E5M2, lg_["down"] is not None, dx_amax, rw)
        dw(dxq, aq, lg_["down"])
        bucket(("layer", i, "down"))
        wd = w8.get(K, i, "down")
        # dact = dx . W_down, the SwiGLU backward and max |dgu| in ONE launch (the [M, I] activation gradient never goes to HBM, the
        # quantiser's amax pass over the 2I-wide gradient is skipped)
        if FUSE_SWIGLU_BWD and not rw:
            dgu, dgu_amax = K.gemm_fp8_dx_swiglu(dxq.q, dxq.dequant, wd.qt, wd.dequant, gu, E5M2)
        else:
            dgu, dgu_amax = K.swiglu_bwd(_dx(K, dxq, wd), gu), None
        del aq, gu, dxq
        dguq = _quant(K, dgu, E5M2, lg_["gu"] is not None, dgu_amax, rw)
        del dgu
        dw(dguq, n2q, lg_["gu"])
        bucket(("layer", i, "gu"))
        dn2 = _dx(K, dguq, w8.get(K, i, "gu"))
        del dguq, n2q
        dx_mid = K.rmsnorm_bwd(dn2, x_mid, lw["ln2"], rstd2, dx, lg_["ln2"], acc, amax_parts=parts_mid)
        del dn2, dx
        dmq = _quant(K, dx_mid, E5M2, lg_["o"] is not None, parts_mid, rw)
        dw(dmq, oq, lg_["o"])
        do = _dx(K, dmq, w8.get(K, i, "o"))
        del dmq, oq
        dqkv = K.attn_bwd(qkv, o, do, lse, B, L, H, Hkv, hd, kmask, scale, True, kstart=kstart, qend=qend)
        del do, o
        K.rope_apply_(dqkv, cos, sin, H + Hkv, hd, backward=True)
        dqq = _quant(K, dqkv, E5M2, lg_["qkv"] is not None, None, rw)
        dw(dqq, n1q, lg_["qkv"])
        if lg_.get("qkv_b") is not None:
            K.colsum(dqkv, lg_["qkv_b"], acc)
        dn1 = _dx(K, dqq, w8.get(K, i, "qkv"))
        del dqkv, dqq, n1q, qkv
        dx = K.rmsnorm_bwd(dn1, x_in, lw["ln1"], rstd1, dx_mid, lg_["ln1"], acc, amax_parts=parts_dx)
        dx_amax = parts_dx
        del dn1, dx_mid, x_in
        bucket(("layer", i, "attn"))
    if side is not None:
        side.join()
    return dx


# ---------------------------------------------------------------------------------------------------------------- engine-side dispatch
def forward(K, eng, lm, tc, x, B, L, position_ids, kmask, kstart, compute_grads, record, rope=None, checkpoint=False):
    """Decoder forward on the engine's precision: `eng.w8` (Fp8Weights, set by ArenaModule.set_precision("fp8")) selects the fp8
    layer loop, None the bf16 one of decoder.py.  The e4m3 weight copies are re-made unless the trainer flagged this micro-batch as
    following another one of the same accumulation window (`eng.weights_unchanged`)."""
    w8 = getattr(eng, "w8", None)
    if w8 is None:
        return D.decoder_forward(K, lm, tc, x, B, L, position_ids, kmask, kstart, compute_grads, record, rope=rope, checkpoint=checkpoint)
    if not getattr(eng, "weights_unchanged", False):
        w8.refresh()
    eng.weights_unchanged = False
    if rope is None:
        rope = K.rope_table(position_ids.reshape(-1), D.inv_freq(tc.head_dim, tc.rope_theta).to(x.device))
    return decoder_forward(K, lm, w8, tc, x, B, L, kmask, compute_grads, record, rope=rope, kstart=kstart, checkpoint=checkpoint)


def backward(K, eng, lm, grads, grads_layers, tc, ctx, hctx, plan, B, L, kmask, kstart, qend, accumulate, on_bucket_ready):
    w8 = getattr(eng, "w8", None)
    if w8 is None:
        return D.decoder_backward(K, lm, grads, grads_layers, tc, ctx, hctx, plan, B, L, kmask, kstart, qend, accumulate, on_bucket_ready)
    return decoder_backward(K, lm, w8, grads, grads_layers, tc, ctx, hctx, plan, B, L, kmask, accumulate, on_bucket_ready, kstart=kstart,
                            qend=qend)
