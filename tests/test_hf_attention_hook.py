"""The HF attention-registry hook (mantis_amd/hf_attention.py, reference plug-in point train_mllava.py:79-82): a stock HF
LlamaForCausalLM with `attn_implementation="mantis_hip"` must match HF's own eager attention, forward and backward, with a
right-padded batch.  CPU: host logic (layout views, HF mask -> key mask, autograd plumbing) with the oracle operators in place of
the HIP backend; `-m gpu`: the same comparison on the real kernels."""
import pytest
import torch


def _tiny_llama(attn):
    transformers = pytest.importorskip("transformers")
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=320, max_position_embeddings=256, rope_theta=500000.0, rms_norm_eps=1e-5, attention_dropout=0.0)
    cfg._attn_implementation = attn
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg)


def _run(model, ids, mask):
    # a token right behind a pad position is not a target: its prediction would come from a query row whose keys are ALL masked (left
    # padding), which every implementation fills with its own garbage (HF eager: uniform attention over all keys)
    ok = mask.bool() & torch.cat([torch.ones_like(mask[:, :1]), mask[:, :-1]], 1).bool()
    out = model(input_ids=ids, attention_mask=mask, labels=torch.where(ok, ids, torch.full_like(ids, -100)))
    out.loss.backward()
    g = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()}
    return out.logits.detach().float().cpu(), float(out.loss), g


def _compare(device, dtype, left_pad=False):
    import mantis_amd.hf_attention as A
    A.register()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 320, (2, 70), generator=g)
    mask = torch.ones_like(ids)
    if left_pad:
        mask[1, :9] = 0          # pad keys sit BEFORE the valid tokens: a causal mask alone does not hide them
    else:
        mask[1, -9:] = 0
    eager = _tiny_llama("eager").to(device=device, dtype=dtype)
    ours = _tiny_llama("mantis_hip").to(device=device, dtype=dtype)
    ours.load_state_dict(eager.state_dict())
    l1, loss1, g1 = _run(eager, ids.to(device), mask.to(device))
    l2, loss2, g2 = _run(ours, ids.to(device), mask.to(device))
    valid = mask.bool()
    rel = float((l1[valid] - l2[valid]).norm() / l1[valid].norm())
    assert rel < 4e-2, rel
    assert abs(loss1 - loss2) < 2e-2 * abs(loss1)
    for n in g1:
        r = float((g1[n] - g2[n]).norm() / (g1[n].norm() + 1e-12))
        assert r < 8e-2, (n, r)


def test_hook_matches_hf_eager_attention_host_logic(monkeypatch):
    import mantis_amd.hf_attention as A
    from oracle import ops_ref
    monkeypatch.setattr(A, "K", ops_ref)
    _compare("cpu", torch.bfloat16)


def test_hook_receives_the_padding_mask_left_padded_batch(monkeypatch):
    """Round-2 advisor finding: without a mask function registered under the same name HF hands the hook attention_mask=None; a
    left-padded batch then differs from HF eager by ~0.6 relative on the valid tokens."""
    import mantis_amd.hf_attention as A
    from oracle import ops_ref
    monkeypatch.setattr(A, "K", ops_ref)
    seen = []
    real = A.key_mask_from_hf

    def spy(m, B, S):
        seen.append(None if m is None else tuple(m.shape))
        return real(m, B, S)
    monkeypatch.setattr(A, "key_mask_from_hf", spy)
    _compare("cpu", torch.bfloat16, left_pad=True)
    assert seen and all(s == (2, 70) for s in seen), seen


def test_mask_function_refuses_what_the_kernels_do_not_implement():
    import mantis_amd.hf_attention as A
    from transformers import masking_utils as M
    m = torch.ones(2, 8, dtype=torch.bool)
    assert A.mantis_hip_mask(2, 8, 8, mask_function=M.causal_mask_function, attention_mask=m) is not None
    assert A.mantis_hip_mask(2, 8, 8, mask_function=M.causal_mask_function, attention_mask=None) is None
    with pytest.raises(NotImplementedError):
        A.mantis_hip_mask(2, 8, 8, mask_function=M.and_masks(M.causal_mask_function, M.sliding_window_causal_mask_function(4)), attention_mask=m)
    with pytest.raises(NotImplementedError):
        A.mantis_hip_mask(2, 1, 8, mask_function=M.causal_mask_function, attention_mask=m)


def test_unsupported_masks_are_refused(monkeypatch):
    import mantis_amd.hf_attention as A
    full = torch.zeros(1, 1, 5, 5)                      # bidirectional 4-D mask: not causal
    with pytest.raises(NotImplementedError):
        A.key_mask_from_hf(full, 1, 5)
    causal = torch.full((1, 1, 5, 5), float("-inf")).triu(1)
    assert A.key_mask_from_hf(causal, 1, 5).tolist() == [[1, 1, 1, 1, 1]]
    causal[0, 0, :, 4] = float("-inf")                  # key 4 padded
    assert A.key_mask_from_hf(causal, 1, 5).tolist() == [[1, 1, 1, 1, 0]]
    assert A.key_mask_from_hf(torch.tensor([[1, 1, 0]]), 1, 3).tolist() == [[1, 1, 0]]
    assert A.key_mask_from_hf(None, 1, 3) is None


@pytest.mark.gpu
def test_hook_matches_hf_eager_attention_on_hip():
    assert torch.cuda.is_available()
    _compare("cuda", torch.bfloat16)


@pytest.mark.gpu
def test_hook_matches_hf_eager_attention_on_hip_left_padded():
    assert torch.cuda.is_available()
    _compare("cuda", torch.bfloat16, left_pad=True)
