"""Explicit forward / backward of the decoder-only text model shared by the LLaVA path (Llama-3), the Idefics2 path (Mistral-7B) and
the Qwen2-VL path (Qwen2-7B):
RMSNorm -> fused q|k|v (+ bias on the Qwen2-VL path) -> RoPE -> causal GQA attention (+ key mask, + sample-packing segment bounds) -> o_proj -> +res -> RMSNorm ->
SwiGLU MLP -> +res, then final RMSNorm, lm_head on the rows that can carry a label, masked shifted cross-entropy.

Reference control flow: HF LlamaModel / MistralModel.forward (transformers/models/llama/modeling_llama.py:284-325,367-418; Mistral is
the same block), invoked from /root/reference/mantis/models/mllava/modeling_llava.py:510-537 and
/root/reference/mantis/models/idefics2/modeling_idefics2.py:1700-1708,1880-1899.  `K` is the operator backend (mantis_amd.hip_ops; the
host-logic tests pass the oracle's operator restatement instead).  `lm` = dict(embed, layers=[dict(qkv, o, gu, down, ln1, ln2)], norm,
head) of arena views; `grads` / `grads_layers` = the same shape over the gradient arena (None where frozen)."""
import torch


def inv_freq(head_dim, theta):
    # transformers/models/llama/modeling_llama.py:95-110 (default rope), computed on the host in fp32 like the reference
    return 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))


def layer_forward(K, lw, tc, x, B, L, cos, sin, kmask, kstart, scale):
    """One decoder block: x [B*L, d] -> (x_out, what its backward reads).  Deterministic (fixed summation orders everywhere): running it
    again on the same x gives the same bits, which is what activation checkpointing relies on."""
    H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    eps = tc.rms_norm_eps
    n1, rstd1 = K.rmsnorm_fwd(x, lw["ln1"], eps)
    qkv = K.linear_qkv_rope(n1, lw["qkv"], lw.get("qkv_b"), cos, sin, H + Hkv, hd)      # RoPE in the projection's epilogue (hd 128)
    o, lse = K.attn_fwd(qkv, B, L, H, Hkv, hd, kmask, scale, True, kstart=kstart)
    x_mid = K.gemm_nt(o, lw["o"], residual=x)
    n2, rstd2 = K.rmsnorm_fwd(x_mid, lw["ln2"], eps)
    gu, a = K.linear_gu_swiglu(n2, lw["gu"])                                            # SwiGLU in the projection's epilogue
    x_out = K.gemm_nt(a, lw["down"], residual=x_mid)
    return x_out, (x, rstd1, qkv, o, lse, x_mid, rstd2, gu, n1, n2, a)


def decoder_forward(K, lm, tc, x, B, L, position_ids, kmask, kstart=None, compute_grads=True, record=None, rope=None, checkpoint=False):
    """x [B*L, d] merged input embeddings -> (hidden states before the final norm, ctx for decoder_backward).
    `rope` = (cos, sin) tables [B*L, hd/2] built by the caller (Qwen2-VL's multimodal RoPE); default: 1-D RoPE of position_ids.
    A layer dict may carry `qkv_b`, the fused q|k|v bias (Qwen2: HF:models/qwen2_vl/modeling_qwen2_vl.py:501-503).
    checkpoint: activation checkpointing per decoder layer (`--gradient_checkpointing True`,
    /root/reference/mantis/train/scripts/train_mllava.sh:168; HF wraps every LlamaDecoderLayer in torch.utils.checkpoint,
    transformers/models/llama/modeling_llama.py:284-325): keep only each layer's INPUT and run the layer's forward again in the backward."""
    hd = tc.head_dim
    scale = hd ** -0.5
    cos, sin = rope if rope is not None else K.rope_table(position_ids.reshape(-1), inv_freq(hd, tc.rope_theta).to(x.device))
    saved = []
    for i in range(tc.num_hidden_layers):
        x_out, keep = layer_forward(K, lm["layers"][i], tc, x, B, L, cos, sin, kmask, kstart, scale)
        if compute_grads:
            # 288 GB of HBM: by default keep the cheap-to-recompute tensors too (n1, n2, a: +370 MB per layer) instead of re-running
            # RMSNorm / SwiGLU in the backward; with `checkpoint` only the layer input stays
            saved.append((x,) if checkpoint else keep)
        del keep
        x = x_out
        if record is not None:
            record[f"llm_layer{i}_out"] = x.view(B, L, -1)
    return x, dict(saved=saved, cos=cos, sin=sin, scale=scale)


def no_padding(attention_mask_cpu, grown, L):
    """True when every row of the batch fills the whole (merged) sequence length L: attention_mask_cpu [B, T] on the host, `grown` the
    rows each sample gains in the merge (LLaVA path: (#image placeholders) * (patches - 1); 0 where placeholders are pre-expanded).
    Then the merged key mask is all ones and the attention kernels can run without one (kmask = None is DEFINED as all keys live)."""
    return bool(((attention_mask_cpu != 0).sum(-1) + grown == L).all())


def compact_ce_rows(plan, input_ids_cpu, attention_mask_cpu, labels, image_token, ignore_index, dev, vocab_size=None,
                    refuse_image_targets=False):
    """Shrink the plan's cross-entropy lists (one entry per (b, t) token) to the entries that can actually carry a label, so the final
    norm, lm_head forward / dX / dW and the loss kernel run on the labelled rows only.  Exact: an ignored target contributes neither loss
    nor gradient (its dlogits row is zero).  The selection is taken on the HOST copy of the batch (no device sync, the GEMM row count must
    be known at launch) and is a superset of the device-side validity -- the kernel's own -100 entries inside it stay ignored.  The list
    is padded to a multiple of 8 rows with (row -1, target -100) entries.  Call it after pack_plan / pack_segments.
    vocab_size: raise IndexError for a countable target outside [0, V), as torch does, on every step.  refuse_image_targets: raise
    where a target sits on an image placeholder (paths with one slot per placeholder, where HF would count it)."""
    import torch
    if labels is None:
        return
    lab = labels.detach().to("cpu") if labels.device.type != "cpu" else labels
    # host-side label checks, EVERY step (the labels are on the host here anyway; only the device-side status words are read back on
    # the first step alone): torch's CrossEntropyLoss raises on a target outside [0, V) on every call, and so does this
    if refuse_image_targets and bool(((lab != ignore_index) & (input_ids_cpu == image_token) & (attention_mask_cpu != 0)).any()):
        raise NotImplementedError("a label other than the ignore index sits on an image placeholder token: HF's loss would count it, "
                                  "this path drops targets on image slots (the reference's collators never produce them)")
    valid = (attention_mask_cpu != 0) & (lab != ignore_index) & (input_ids_cpu != image_token)
    if vocab_size is not None:
        bad = valid & ((lab < 0) | (lab >= vocab_size))
        if bool(bad.any()):
            raise IndexError(f"{int(bad.sum())} label(s) are outside [0, vocab_size = {vocab_size}) "
                             f"(torch.nn.CrossEntropyLoss: 'Target out of bounds')")
    sel = torch.nonzero(valid.reshape(-1)).reshape(-1)
    n = int(sel.numel())
    pad = 8 if n == 0 else (-n) % 8
    sel_d = sel.to(dev, non_blocking=True)
    row, tgt = plan.ce_row.index_select(0, sel_d), plan.ce_tgt.index_select(0, sel_d)
    if pad:
        row = torch.cat([row, torch.full((pad,), -1, dtype=row.dtype, device=row.device)])
        tgt = torch.cat([tgt, torch.full((pad,), -100, dtype=tgt.dtype, device=tgt.device)])
    plan.ce_row, plan.ce_tgt = row.contiguous(), tgt.contiguous()


def head_and_loss(K, lm, tc, x, plan, B, L, labels_given, grad_scale, loss_scale, compute_grads, need_logits, record=None):
    """Final norm + lm_head + masked shifted CE.  Only the rows that can carry a label (plan.ce_row) reach lm_head for the loss; the
    full [B, L, V] logits are produced only on request.  Returns (loss, count, logits_full, ctx)."""
    eps = tc.rms_norm_eps
    V = tc.vocab_size
    Vp = K.pad8(V)
    logits_full = None
    if need_logits:
        nf_all, _ = K.rmsnorm_fwd(x, lm["norm"], eps, want_rstd=False)
        if record is not None:
            record["llm_final_norm"] = nf_all.view(B, L, -1)
        lg = K.gemm_nt(nf_all, lm["head"], ldc=Vp)
        logits_full = lg.view(B, L, Vp)[:, :, :V]
    loss = count = ctx = None
    if labels_given or compute_grads:
        h_ce = K.gather_rows(x, plan.ce_row)                       # [n, d] rows that carry a label (compact_ce_rows), else all B*T
        nf, rstdf = K.rmsnorm_fwd(h_ce, lm["norm"], eps)
        logits = K.gemm_nt(nf, lm["head"], ldc=Vp)                # [B*T, Vp]
        loss, count = K.ce_fwd_bwd(logits, plan.ce_tgt, V, grad_scale, loss_scale, write_grad=compute_grads)
        ctx = dict(dlogits=logits, nf=nf, h_ce=h_ce, rstdf=rstdf, Vp=Vp)      # logits were overwritten in place with dlogits
    return loss, count, logits_full, ctx


def head_backward(K, lm, grads, hctx, plan, B, L, acc, on_bucket_ready=None):
    """Backward of head_and_loss: lm_head dW / dX, final RMSNorm, scatter to the sequence rows.  Returns dx [B*L, d]."""
    dlogits, nf, h_ce, rstdf = hctx["dlogits"], hctx["nf"], hctx["h_ce"], hctx["rstdf"]
    if grads.get("head") is not None:
        K.linear_dw(dlogits, nf, grads["head"], acc)      # pad columns [V, Vp) are zero
    dnf = K.linear_dx(dlogits, lm["head"], k=hctx["Vp"])
    dh_ce = K.rmsnorm_bwd(dnf, h_ce, lm["norm"], rstdf, None, grads.get("norm"), acc)
    dx = K.scatter_rows(dh_ce, plan.ce_row, B * L)
    hctx.clear()
    if on_bucket_ready is not None:
        on_bucket_ready("head")
    return dx


def decoder_backward(K, lm, grads, grads_layers, tc, ctx, hctx, plan, B, L, kmask, kstart=None, qend=None, accumulate=False,
                     on_bucket_ready=None):
    """Backward of head_and_loss + decoder_forward.  Returns dx [B*L, d], the gradient w.r.t. the merged input embeddings.
    Fires on_bucket_ready("head") and (("layer", i, "down" | "gu" | "attn")) as each gradient bucket completes."""
    H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    acc = accumulate
    saved, cos, sin, scale = ctx["saved"], ctx["cos"], ctx["sin"], ctx["scale"]
    dx = head_backward(K, lm, grads, hctx, plan, B, L, acc, on_bucket_ready)
    # The weight-gradient GEMMs are off the critical path (nothing in the backward reads a dW): with a side stream (opt-in,
    # hip_ops.side_stream) they queue there, behind the kernel that produced their dY, and their workgroups fill the compute units the
    # dX chain's kernels leave idle in incomplete tile rounds.  The bucket hooks follow them onto that stream (a bucket is ready when
    # its dW GEMMs AND the main-stream kernels before the hook are done); the caller's stream joins at the end.
    side = K.side_stream()

    def off_path(fn, *inputs):
        if side is None:
            fn()
        else:
            side.run(fn, *inputs)

    def bucket(key):
        if on_bucket_ready is not None:
            off_path(lambda: on_bucket_ready(key))

    # Round 6: on a single stream without bucket hooks (no gradient exchange to start early) dW(down_proj) waits for the layer's dW(q|k|v) and the
    # two run as ONE grid where the library predicts a gain (hip_ops.linear_dw_pair: 896 + 384 tiles = 5.0 whole rounds on 256 CUs for the
    # Llama-3-8B / Mistral-7B geometry instead of two launches with a K-split remainder round and a finishing pass each).  Its operands -- the
    # layer's incoming gradient and the SwiGLU output -- are held until then (+ one [rows, d] and one [rows, I] buffer of lifetime).
    pair_dw = None
    for i in reversed(range(tc.num_hidden_layers)):
        lw = lm["layers"][i]
        lg_ = grads_layers[i]
        entry = saved.pop()
        if len(entry) == 1:          # activation checkpointing: only the layer input was kept -- run the layer's forward again
            _, entry = layer_forward(K, lw, tc, entry[0], B, L, cos, sin, kmask, kstart, scale)
        x_in, rstd1, qkv, o, lse, x_mid, rstd2, gu, n1, n2, a = entry
        del entry
        if pair_dw is None:
            pair_dw = bool(side is None and on_bucket_ready is None and lg_["down"] is not None and lg_["qkv"] is not None and
                           hasattr(K, "dw_pair_wins") and
                           K.dw_pair_wins(lg_["down"].shape[0], lg_["down"].shape[1], lg_["qkv"].shape[0], lg_["qkv"].shape[1], dx.shape[0]))
        held = None
        if pair_dw and lg_["down"] is not None and lg_["qkv"] is not None:
            held = (dx, a)                                   # dW(down) runs with dW(q|k|v) below
        elif lg_["down"] is not None:
            off_path(lambda dx=dx, a=a: K.linear_dw(dx, a, lg_["down"], acc), dx, a)
        bucket(("layer", i, "down"))
        dgu = K.linear_dx_swiglu(dx, lw["down"], gu)     # dact = dx . W_down and the SwiGLU backward in one launch
        del a, gu
        if lg_["gu"] is not None:
            off_path(lambda dgu=dgu, n2=n2: K.linear_dw(dgu, n2, lg_["gu"], acc), dgu, n2)
        bucket(("layer", i, "gu"))
        dn2 = K.linear_dx(dgu, lw["gu"])
        del dgu, n2
        dx_mid = K.rmsnorm_bwd(dn2, x_mid, lw["ln2"], rstd2, dx, lg_["ln2"], acc)
        del dn2, dx
        if lg_["o"] is not None:
            off_path(lambda dx_mid=dx_mid, o=o: K.linear_dw(dx_mid, o, lg_["o"], acc), dx_mid, o)
        do = K.linear_dx(dx_mid, lw["o"])
        dqkv = K.attn_bwd(qkv, o, do, lse, B, L, H, Hkv, hd, kmask, scale, True, kstart=kstart, qend=qend)
        del do, o
        K.rope_apply_(dqkv, cos, sin, H + Hkv, hd, backward=True)
        if held is not None:
            K.linear_dw_pair(held[0], held[1], lg_["down"], dqkv, n1, lg_["qkv"], acc)
            held = None
        elif lg_["qkv"] is not None:
            off_path(lambda dqkv=dqkv, n1=n1: K.linear_dw(dqkv, n1, lg_["qkv"], acc), dqkv, n1)
        if lg_.get("qkv_b") is not None:
            K.colsum(dqkv, lg_["qkv_b"], acc)
        dn1 = K.linear_dx(dqkv, lw["qkv"])
        del dqkv, n1, qkv
        dx = K.rmsnorm_bwd(dn1, x_in, lw["ln1"], rstd1, dx_mid, lg_["ln1"], acc)
        del dn1, dx_mid, x_in
        bucket(("layer", i, "attn"))
    if side is not None:
        side.join()
    return dx
