"""CPU: the PRODUCT's Idefics2 host logic (mantis_amd/modeling_idefics2.py: image preparation, vision tower sequencing, connector
forward / backward, merger, decoder, gradient plumbing) with the oracle's operator restatement monkeypatched in place of the HIP
backend, against the Idefics2 oracle (pinned to the reference fork) on the golden inputs."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh

CASES = ["idefics2_b1_img2", "idefics2_b1_navit", "idefics2_b2_padimg_rightpad", "idefics2_b1_text_only"]


@pytest.fixture()
def cpu_backend(monkeypatch):
    import mantis_amd.modeling_idefics2 as mod
    from oracle import ops_ref
    monkeypatch.setattr(mod, "K", ops_ref)
    return mod


@pytest.mark.parametrize("case", CASES)
def test_step_matches_oracle(cpu_backend, case):
    z = Hh.load_case(case)
    model = Hh.build_idefics2_product("cpu")
    oracle = Hh.build_idefics2_oracle_bf16()
    assert model._ensure_grad_arena()
    rec = {}
    out = model.engine.step_from_batch(Hh.idefics2_batch(z), compute_grads=True, overwrite_grads=True, need_logits=True, record=rec)
    Hh.check_idefics2_step_against_oracle(model, oracle, z, out, rec)
    assert abs(float(out["loss"]) - float(z["loss"])) < 0.03 * float(z["loss"])          # vs the reference's own fp32 loss
    for n, p in model.named_parameters():
        if n.startswith("model.vision_model."):
            assert p.grad is None


def test_trainer_drives_the_idefics2_engine(cpu_backend):
    from mantis_amd.trainer import MantisHipTrainer
    z = Hh.load_case("idefics2_b2_padimg_rightpad")
    model = Hh.build_idefics2_product("cpu")
    tr = MantisHipTrainer(model, gradient_accumulation_steps=2)
    l1 = tr.training_step(model, Hh.idefics2_batch(z))
    g1 = model.grad_arena.float().clone()
    l2 = tr.training_step(model, Hh.idefics2_batch(z))
    assert l1.dim() == 0 and abs(float(l1) - float(z["loss"]) / 2) < 0.03 * float(z["loss"]) and torch.equal(l1, l2)
    assert Hh.rel_l2(model.grad_arena.float().numpy(), 2 * g1.numpy()) < 2e-2          # accumulation over the GA window


def test_image_token_count_mismatch_raises(cpu_backend):
    z = Hh.load_case("idefics2_b1_img2")
    model = Hh.build_idefics2_product("cpu")
    b = Hh.idefics2_batch(z)
    b["input_ids"] = b["input_ids"].clone()
    b["input_ids"][0, 3] = 5                      # one <image> token fewer than image hidden states
    with pytest.raises(ValueError):
        model.engine.step_from_batch(b, compute_grads=False)


def test_state_dict_names_match_reference():
    model = Hh.build_idefics2_product("cpu")
    _, sd = Hh.golden_cfg_and_weights("idefics2")
    assert {n for n, _ in model.named_parameters()} == set(sd)
    assert sum(b.numel() for b in model.grad_buckets().values()) == model.grad_arena.numel()


def test_packed_row_equals_the_separate_samples(cpu_backend):
    """Long-sequence packing on the Idefics2 path (BASELINE configs[3]): the two samples of the B=2 golden batch (unpadded parts)
    packed into one row -- loss and gradients equal the oracle running them one by one."""
    z = Hh.load_case("idefics2_b2_padimg_rightpad")
    ids, am, lab = z["input_ids"], z["attention_mask"], z["labels"]
    keep = [am[b].astype(bool) for b in range(2)]
    pid = torch.from_numpy(np.concatenate([ids[b][keep[b]] for b in range(2)]))[None]
    plab = torch.from_numpy(np.concatenate([lab[b][keep[b]] for b in range(2)]))[None]
    seg = torch.from_numpy(np.concatenate([np.full(int(keep[b].sum()), b, np.int32) for b in range(2)]))[None]
    pv = torch.from_numpy(np.concatenate([z["pixel_values"][0], z["pixel_values"][1][:1]], 0))[None]        # 3 real images in order
    pm = torch.from_numpy(np.concatenate([z["pixel_attention_mask"][0], z["pixel_attention_mask"][1][:1]], 0))[None]
    model = Hh.build_idefics2_product("cpu")
    oracle = Hh.build_idefics2_oracle_bf16()
    assert model._ensure_grad_arena()
    out = model.engine.step(pid, torch.ones_like(pid), plab, pv, pm, compute_grads=True, overwrite_grads=True, segment_ids=seg)
    oracle.zero_grad()
    oloss = oracle.forward_packed(pid, pv, pm, seg, torch.ones_like(pid), plab)
    oloss.backward()
    assert abs(float(out["loss"]) - float(oloss)) <= 5e-3 * float(oloss), (float(out["loss"]), float(oloss))
    for name, p in model.named_parameters():
        if p.requires_grad:
            g, og = p.grad.float().numpy(), oracle.w[name].grad.numpy()
            assert Hh.cosine(g, og) > 0.995 and Hh.rel_l2(g, og) < 6e-2, (name, Hh.cosine(g, og), Hh.rel_l2(g, og))


def test_loss_backward_through_the_autograd_bridge(cpu_backend):
    """Stock `model(**batch).loss.backward()` callers: same loss and bit-identical gradients as MantisHipTrainer.training_step."""
    import torch
    from mantis_amd.trainer import MantisHipTrainer
    z = Hh.load_case("idefics2_b2_padimg_rightpad")
    ref = Hh.build_idefics2_product("cpu")
    l_ref = MantisHipTrainer(ref, gradient_accumulation_steps=1).training_step(ref, Hh.idefics2_batch(z))
    model = Hh.build_idefics2_product("cpu")
    model.train()
    out = model(**Hh.idefics2_batch(z))
    assert out.logits is None and out.loss.requires_grad
    out.loss.backward()
    assert torch.equal(out.loss.detach(), l_ref) and torch.equal(model.grad_arena, ref.grad_arena)
    for p in model.parameters():
        p.grad = None
    model(**Hh.idefics2_batch(z)).loss.backward()
    assert torch.equal(model.grad_arena, ref.grad_arena)
