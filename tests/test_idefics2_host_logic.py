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
