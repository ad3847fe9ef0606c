"""Operator-level hook: the gfx950 attention kernels behind HuggingFace's attention registry.

The reference selects its attention kernel through `--attn_implementation` (/root/reference/mantis/train/train_mllava.py:79-82,
flash_attention_2 by default): HF looks the name up in `ALL_ATTENTION_FUNCTIONS` / `AttentionInterface` and calls
    fn(module, query[B,H,S,D], key[B,Hkv,S,D], value[B,Hkv,S,D], attention_mask, scaling=..., dropout=..., **kw)
        -> (attn_output[B,S,H,D], attn_weights | None)                       (HF:models/llama/modeling_llama.py:262-276)
`register()` adds the name "mantis_hip" to that registry, so a user who keeps the stock HF modules (instead of the fused
`mantis_amd.modeling_llava` step) still runs the hand-written CDNA4 forward AND backward for attention:
    import mantis_amd.hf_attention as A; A.register();  ... --attn_implementation mantis_hip
Supported masks are the ones the Mantis path produces: causal, or causal + key padding (the 4-D additive mask HF's
`create_causal_mask` builds from the 2-D attention mask, or the 2-D mask itself).  Anything else (sliding window, arbitrary
4-D masks, attention dropout, returned attention weights) raises NotImplementedError rather than computing something else."""
import torch

from . import hip_ops as K


def _rows(x):
    """[B, H, S, D] (any strides) -> ([B*S, H*D] row view with heads contiguous inside a row, B, H, S, D); copies only if the
    memory is not already [B, S, H, D]-contiguous (HF produces q/k/v as .view(B,S,H,D).transpose(1,2): no copy)."""
    B, H, S, D = x.shape
    t = x.transpose(1, 2)
    if not t.is_contiguous():
        t = t.contiguous()
    return t.view(B * S, H * D), B, H, S, D


def key_mask_from_hf(attention_mask, B, S):
    """HF mask -> int32 key mask [B, S] (1 = attend) for a causal attention, or None.  Accepts None, the 2-D padding mask, or the
    4-D causal(+padding) mask (bool: True = attend; float: 0 = attend, large negative = masked)."""
    if attention_mask is None:
        return None
    m = attention_mask
    if m.dim() == 2:
        if m.shape != (B, S):
            raise NotImplementedError(f"2-D attention mask of shape {tuple(m.shape)} for B={B}, S={S}")
        return (m != 0).to(torch.int32).contiguous()
    if m.dim() != 4 or m.shape[0] != B or m.shape[1] != 1 or m.shape[2] != S or m.shape[3] != S:
        raise NotImplementedError(f"attention mask of shape {tuple(m.shape)}: only [B,1,S,S] causal(+padding) masks are supported")
    allowed = m if m.dtype == torch.bool else (m == 0)
    last = allowed[:, 0, -1, :]                       # the last query row sees every non-padded key under a causal mask
    # cheap structural check (O(S) per sample, not O(S^2)): the first query row may see at most key 0, the diagonal is allowed
    # wherever the key itself is allowed
    if S > 1 and bool(allowed[:, 0, 0, 1:].any()):
        raise NotImplementedError("4-D attention mask is not causal")
    diag = torch.diagonal(allowed[:, 0], dim1=-2, dim2=-1)
    if not bool((diag | ~last).all()):
        raise NotImplementedError("4-D attention mask is not a causal + key-padding mask")
    return last.to(torch.int32).contiguous()


class _HipAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, kmask, scale, causal):
        q2, B, H, S, D = _rows(q)
        k2, _, Hkv, _, _ = _rows(k)
        v2 = _rows(v)[0]
        o, lse = K.attn_fwd_qkv(q2, k2, v2, B, S, H, Hkv, D, kmask, scale, causal)
        ctx.save_for_backward(q2, k2, v2, o, lse, kmask if kmask is not None else torch.empty(0))
        ctx.dims = (B, H, Hkv, S, D, scale, causal, kmask is not None)
        return o.view(B, S, H, D)

    @staticmethod
    def backward(ctx, do):
        q2, k2, v2, o, lse, kmask = ctx.saved_tensors
        B, H, Hkv, S, D, scale, causal, has_mask = ctx.dims
        do2 = do.contiguous().view(B * S, H * D)
        dq = torch.empty_like(q2)
        dk, dv = torch.empty_like(k2), torch.empty_like(v2)
        K.attn_bwd_qkv(q2, k2, v2, o, do2, lse, dq, dk, dv, B, S, H, Hkv, D, kmask if has_mask else None, scale, causal)
        back = lambda t, h: t.view(B, S, h, D).transpose(1, 2)          # gradient w.r.t. the [B, H, S, D] argument
        return back(dq, H), back(dk, Hkv), back(dv, Hkv), None, None, None


def mantis_hip_attention(module, query, key, value, attention_mask, scaling=None, dropout=0.0, is_causal=None, **kwargs):
    """HF attention-interface function (see module docstring)."""
    if dropout and getattr(module, "training", False):
        raise NotImplementedError("attention dropout is not implemented on the gfx950 path (the Mantis Llama configs use 0.0)")
    if kwargs.get("sliding_window") is not None or kwargs.get("output_attentions"):
        raise NotImplementedError("sliding-window attention / returned attention weights are not produced by the fused kernels")
    if query.dtype != torch.bfloat16 or (not query.is_cuda and K.__name__.endswith("hip_ops")):
        raise NotImplementedError(f"mantis_hip attention computes in bf16 on the GPU; got {query.dtype} on {query.device}")
    B, H, S, D = query.shape
    if key.shape[2] != S:
        raise NotImplementedError("KV-cache decoding (key length != query length) is out of scope: training forward only")
    causal = is_causal if is_causal is not None else getattr(module, "is_causal", True)
    kmask = key_mask_from_hf(attention_mask, B, S)
    if attention_mask is not None and attention_mask.dim() == 4:
        causal = True
    scale = float(scaling) if scaling is not None else D ** -0.5
    out = _HipAttention.apply(query, key, value, kmask, scale, bool(causal))
    return out, None


def mantis_hip_mask(batch_size, q_length, kv_length, q_offset=0, kv_offset=0, mask_function=None, attention_mask=None, **kwargs):
    """HF mask-interface function for this implementation (transformers.masking_utils.AttentionMaskInterface): hands the 2-D key-padding
    mask through untouched -- the kernels take causality as a flag and padding as an O(S) key mask, so no [B,1,S,S] tensor is built
    and no device sync is needed (HF's flash-attention mask function tests `mask.all()` on the host).  Everything that is not plain
    causal (+ padding) -- packed-sequence / sliding / chunked mask functions, a KV cache -- is refused instead of being dropped."""
    from transformers import masking_utils as M
    if mask_function is not None and mask_function is not M.causal_mask_function:
        raise NotImplementedError("mantis_hip attention: only the plain causal mask (+ key padding) is implemented; HF composed another "
                                  "mask function (packed sequences / sliding window / chunked attention)")
    if q_length != kv_length or q_offset != 0 or kv_offset != 0:
        raise NotImplementedError("mantis_hip attention: KV-cache decoding (query length != key length) is out of scope")
    if attention_mask is None:
        return None
    return attention_mask[:, -kv_length:]


def register(name="mantis_hip"):
    """Make `--attn_implementation mantis_hip` (config._attn_implementation) resolve to the gfx950 kernels.

    BOTH registries are needed: `AttentionInterface` for the attention function and `AttentionMaskInterface` for the mask that HF's
    `create_causal_mask` hands to it -- for a name missing from the mask registry `create_causal_mask` returns None and the hook would
    run every batch unmasked (a left-padded batch then attends its pad keys: round-2 advisor finding).  If this transformers version has
    no mask registry to add to, registration fails rather than computing unmasked attention."""
    from transformers import AttentionInterface
    try:
        from transformers.masking_utils import AttentionMaskInterface
    except ImportError as e:
        raise RuntimeError("mantis_hip attention needs transformers.masking_utils.AttentionMaskInterface to receive the padding mask; "
                           "this transformers version does not have it") from e
    AttentionMaskInterface.register(name, mantis_hip_mask)
    AttentionInterface.register(name, mantis_hip_attention)
    return name
