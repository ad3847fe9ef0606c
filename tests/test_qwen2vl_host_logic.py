"""CPU: the PRODUCT's Qwen2-VL host logic (mantis_amd/modeling_qwen2_vl.py: 3-D rope index, 2-D vision rotary ids, per-image attention
grouping, tower + merger sequencing, image-token merge, decoder with q/k/v bias and multimodal RoPE, gradient plumbing) with the
oracle's operator restatement monkeypatched in place of the HIP backend, against the Qwen2-VL oracle (pinned to the HF class the
reference resolves to) on the golden inputs."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh

CASES = ["qwen2vl_b1_img2", "qwen2vl_b1_img1_tall", "qwen2vl_b2_rightpad", "qwen2vl_b1_text_only"]


@pytest.fixture()
def cpu_backend(monkeypatch):
    import mantis_amd.modeling_qwen2_vl as mod
    from oracle import ops_ref
    monkeypatch.setattr(mod, "K", ops_ref)
    return mod


@pytest.mark.parametrize("case", CASES)
def test_step_matches_oracle(cpu_backend, case):
    z = Hh.load_case(case)
    model = Hh.build_qwen2vl_product("cpu")
    oracle = Hh.build_qwen2vl_oracle_bf16()
    assert model._ensure_grad_arena()
    rec = {}
    out = model.engine.step_from_batch(Hh.qwen2vl_batch(z), compute_grads=True, overwrite_grads=True, need_logits=True, record=rec)
    if "position_ids" in z.files:
        assert np.array_equal(rec["position_ids"].numpy(), z["position_ids"])        # vs the reference's own get_rope_index
    Hh.check_qwen2vl_step_against_oracle(model, oracle, z, out, rec)
    assert abs(float(out["loss"]) - float(z["loss"])) < 0.03 * float(z["loss"])      # vs the reference's own fp32 loss
    for n, p in model.named_parameters():
        if n.startswith("model.visual."):
            assert p.grad is None


def test_trainer_drives_the_qwen2vl_engine(cpu_backend):
    from mantis_amd.trainer import MantisHipTrainer
    z = Hh.load_case("qwen2vl_b2_rightpad")
    model = Hh.build_qwen2vl_product("cpu")
    tr = MantisHipTrainer(model, gradient_accumulation_steps=2)
    l1 = tr.training_step(model, Hh.qwen2vl_batch(z))
    g1 = model.grad_arena.float().clone()
    l2 = tr.training_step(model, Hh.qwen2vl_batch(z))
    assert l1.dim() == 0 and abs(float(l1) - float(z["loss"]) / 2) < 0.03 * float(z["loss"]) and torch.equal(l1, l2)
    assert Hh.rel_l2(model.grad_arena.float().numpy(), 2 * g1.numpy()) < 2e-2          # accumulation over the GA window


def test_image_token_count_mismatch_raises(cpu_backend):
    z = Hh.load_case("qwen2vl_b1_img2")
    model = Hh.build_qwen2vl_product("cpu")
    b = Hh.qwen2vl_batch(z)
    b["input_ids"] = b["input_ids"].clone()
    b["input_ids"][0, 4] = 5                      # one <|image_pad|> token fewer than merged patches
    with pytest.raises(ValueError):
        model.engine.step_from_batch(b, compute_grads=False)


def test_labelled_padding_is_rejected(cpu_backend):
    z = Hh.load_case("qwen2vl_b2_rightpad")
    model = Hh.build_qwen2vl_product("cpu")
    b = Hh.qwen2vl_batch(z)
    b["labels"] = b["labels"].clone()
    b["labels"][1, -1] = 7                        # a label on a padded position: HF would count it, this path refuses
    with pytest.raises(NotImplementedError):
        model.engine.step_from_batch(b, compute_grads=False)


def test_state_dict_names_match_reference_and_hf4_names_load():
    model = Hh.build_qwen2vl_product("cpu")
    _, sd = Hh.golden_cfg_and_weights("qwen2vl")
    assert set(dict(model.named_parameters())) == set(sd)
    # a transformers-4.x checkpoint (visual.*, model.layers.*) lands on the same parameters
    old = {}
    for k, v in sd.items():
        if k.startswith("model.visual."):
            old[k[len("model."):]] = v
        elif k.startswith("model.language_model."):
            old["model." + k[len("model.language_model."):]] = v
        else:
            old[k] = v
    m2 = Hh.build_qwen2vl_product("cpu")
    m2.arena.zero_()
    assert m2.load_reference_state_dict(old) == []
    assert torch.equal(m2.arena, model.arena)


def test_forward_contract(cpu_backend):
    z = Hh.load_case("qwen2vl_b1_img2")
    model = Hh.build_qwen2vl_product("cpu")
    model.eval()
    b = Hh.qwen2vl_batch(z)
    with torch.no_grad():
        out = model(**b)
    assert out.logits.dtype == torch.float32 and tuple(out.logits.shape) == tuple(z["logits"].shape)
    assert abs(float(out.loss) - float(z["loss"])) < 0.03 * float(z["loss"])
    assert out["loss"] is out.loss and out[0] is out.loss
    with pytest.raises(NotImplementedError):
        model(**b, pixel_values_videos=torch.zeros(1, 1))
