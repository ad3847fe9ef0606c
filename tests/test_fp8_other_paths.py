"""CPU: `set_precision("fp8")` on the LLaVA and Idefics2 modules (ArenaModule.set_precision -> decoder_fp8 through the engines' dispatch)
with the oracle's exact restatement of the fp8 arithmetic in place of the HIP backend, against the fp32 oracles of the reference within
the fp8 variant's stated tolerance.  (BASELINE names fp8 for the Qwen2-VL configuration only; the headline stays bf16 -- this is the
same accelerated variant of the decoder linears offered on the other two paths.)"""
import pytest
import torch

from tests import helpers as Hh


PRECISIONS = ["fp8", "fp8_rowwise"]          # per-tensor scales; one scale per token / per feature (opt-in)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_llava_fp8_step(monkeypatch, precision):
    import mantis_amd.engine as eng
    from oracle import ops_ref
    monkeypatch.setattr(eng, "K", ops_ref)
    z = Hh.load_case("siglip_b2_equal_rightpad")
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    model.set_precision(precision)
    assert model.engine.w8.rowwise == (precision == "fp8_rowwise")
    oracle = Hh.build_oracle_bf16_weights("siglip")
    assert model._ensure_grad_arena()
    out = model.engine.step(torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]), torch.from_numpy(z["labels"]),
                            Hh.pixels_list(z), compute_grads=True, overwrite_grads=True)
    oracle.zero_grad()
    oloss, _ = oracle.forward(z["input_ids"], Hh.pixels_list(z), z["attention_mask"], z["labels"])
    oloss.backward()
    assert Hh.check_fp8_grads_against_oracle(model, oracle, out["loss"], oloss) > 0.85
    assert model.set_precision("bf16").engine.w8 is None


@pytest.mark.parametrize("precision", PRECISIONS)
def test_idefics2_fp8_step(monkeypatch, precision):
    import mantis_amd.modeling_idefics2 as mod
    from oracle import ops_ref
    monkeypatch.setattr(mod, "K", ops_ref)
    z = Hh.load_case("idefics2_b2_padimg_rightpad")
    model = Hh.build_idefics2_product("cpu").set_precision(precision)
    oracle = Hh.build_idefics2_oracle_bf16()
    assert model._ensure_grad_arena()
    out = model.engine.step_from_batch(Hh.idefics2_batch(z), compute_grads=True, overwrite_grads=True)
    oracle.zero_grad()
    oloss, _ = oracle.forward(z["input_ids"], z["pixel_values"], z["pixel_attention_mask"], z["attention_mask"], z["labels"])
    oloss.backward()
    assert Hh.check_fp8_grads_against_oracle(model, oracle, out["loss"], oloss) > 0.85
