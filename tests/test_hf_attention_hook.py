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
    out = model(input_ids=ids, attention_mask=mask, labels=torch.where(mask.bool(), ids, torch.full_like(ids, -100)))
    out.loss.backward()
    g = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()}
    return out.logits.detach().float().cpu(), float(out.loss), g


def _compare(device, dtype):
    import mantis_amd.hf_attention as A
    A.register()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 320, (2, 70), generator=g)
    mask = torch.ones_like(ids)
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
