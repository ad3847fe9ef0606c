"""CPU: the fp8 variant of the Qwen2-VL decoder (mantis_amd/decoder_fp8.py: which tensors are quantised in which format, which copies
the forward / dX / dW GEMMs read, what the backward keeps, weight-copy reuse across an accumulation window) with the oracle's exact
restatement of the fp8 arithmetic (oracle/ops_ref.py fp8_quantize / gemm_fp8_nt) in place of the HIP backend, against the fp32 Qwen2-VL
oracle.  The reference has no fp8: the tolerance of this accelerated variant is stated here -- loss 1e-2 relative, activations 0.15
relative L2, weight-matrix gradient cosine >= 0.95, bias / norm-weight gradient cosine >= 0.85 (the key bias' gradient is a near-cancelling
sum: mostly quantisation noise) -- per-tensor e4m3 activations / weights, e5m2 output gradients."""
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
def test_fp8_step_within_stated_tolerance(cpu_backend, case):
    z = Hh.load_case(case)
    model = Hh.build_qwen2vl_product("cpu").set_precision("fp8")
    assert model._ensure_grad_arena()
    rec = {}
    out = model.engine.step_from_batch(Hh.qwen2vl_batch(z), compute_grads=True, overwrite_grads=True, need_logits=True, record=rec)
    Hh.check_qwen2vl_step_against_oracle(model, Hh.build_qwen2vl_oracle_bf16(), z, out, rec, loss_rtol=1e-2, grad_cos=Hh.FP8_COS_2D,
                                         grad_rel=Hh.FP8_GRAD_REL_2D, act_rel=Hh.FP8_ACT_REL, grad_cos_1d=Hh.FP8_COS_1D, grad_rel_1d=0.6)


@pytest.mark.parametrize("case", ["qwen2vl_b1_img2", "qwen2vl_b2_rightpad"])
def test_fp8_rowwise_step_within_stated_tolerance(cpu_backend, case):
    """set_precision("fp8_rowwise"): the same stated tolerance as the per-tensor recipe (it is at least as fine everywhere)."""
    z = Hh.load_case(case)
    model = Hh.build_qwen2vl_product("cpu").set_precision("fp8_rowwise")
    assert model._ensure_grad_arena() and model.engine.w8.rowwise
    rec = {}
    out = model.engine.step_from_batch(Hh.qwen2vl_batch(z), compute_grads=True, overwrite_grads=True, need_logits=True, record=rec)
    Hh.check_qwen2vl_step_against_oracle(model, Hh.build_qwen2vl_oracle_bf16(), z, out, rec, loss_rtol=1e-2, grad_cos=Hh.FP8_COS_2D,
                                         grad_rel=Hh.FP8_GRAD_REL_2D, act_rel=Hh.FP8_ACT_REL, grad_cos_1d=Hh.FP8_COS_1D, grad_rel_1d=0.6)


def test_rowwise_quantiser_restatement_properties():
    from oracle import ops_ref as R
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(37, 48, generator=g) * 3)
    x[5] *= 1e-6                                    # a row far below the tensor's range: per-tensor e4m3 flushes it to zero
    x[9] = 0.0
    x[:, 7] = 0.0
    x = x.to(torch.bfloat16)
    for fmt, fmax, dt in ((0, 448.0, torch.float8_e4m3fn), (1, 57344.0, torch.float8_e5m2)):
        t = R.fp8_quantize(x, fmt, rowwise=True)
        assert t.rowwise and t.state is None and t.dequant is t.row_dequant and t.dequant_t is t.col_dequant
        assert t.row_dequant.shape == (37,) and t.col_dequant.shape == (48,) and t.qt.shape == (48, 48) and not t.qt[:, 37:].any()
        assert float(t.row_dequant[9]) == 1.0 and float(t.col_dequant[7]) == 1.0 and not t.q[9].any() and not t.qt[7].any()
        deq = t.q.view(dt).float() * t.row_dequant[:, None]
        deq_t = t.qt[:, :37].view(dt).float() * t.col_dequant[:, None]
        # every row's (column's) largest element maps to FMAX exactly
        assert torch.allclose(deq.abs().amax(1)[x.float().abs().amax(1) > 0], x.float().abs().amax(1)[x.float().abs().amax(1) > 0], rtol=1e-6)
        assert torch.allclose(deq_t.abs().amax(1)[x.float().abs().amax(0) > 0], x.float().abs().amax(0)[x.float().abs().amax(0) > 0], rtol=1e-6)
        bound = 0.04 if fmt == 0 else 0.08
        assert Hh.rel_l2(deq.numpy(), x.float().numpy()) < bound
        assert Hh.rel_l2(deq[5].numpy(), x[5].float().numpy()) < bound            # the tiny row keeps full relative precision ...
        if fmt == 0:
            pt = R.fp8_quantize(x, fmt)
            assert not (pt.q[5] & 0x7f).any()                                     # ... which one scale per tensor cannot give it
    # GEMM on the quantised bytes = matmul of the dequantised values, scales applied as an outer product
    a, b = R.fp8_quantize(x, 1, transposed=False, rowwise=True), R.fp8_quantize(x[:20], 0, transposed=False, rowwise=True)
    y = R.gemm_fp8_nt(a.q, a.dequant, b.q, b.dequant, 1, rowwise=True)
    ref = (a.q.view(torch.float8_e5m2).float() * a.row_dequant[:, None]) @ (b.q.view(torch.float8_e4m3fn).float() * b.row_dequant[:, None]).t()
    assert Hh.rel_l2(y.float().numpy(), ref.numpy()) < 5e-3
    # the three GEMMs of a linear see per-row scales on both operands: dW = dY^T X from the transposed copies
    dy = (torch.randn(37, 32, generator=g)).to(torch.bfloat16)
    dq, xq = R.fp8_quantize(dy, 1, rowwise=True), R.fp8_quantize(x, 0, rowwise=True)
    dw = R.gemm_fp8_nt(dq.qt, dq.dequant_t, xq.qt, xq.dequant_t, 1, rowwise=True)
    assert dw.shape == (32, 48) and Hh.rel_l2(dw.float().numpy(), (dy.float().t() @ x.float()).numpy()) < 0.08


def test_fp8_quantiser_restatement_properties():
    from oracle import ops_ref as R
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(37, 48, generator=g) * 3).to(torch.bfloat16)
    for fmt, fmax, dt in ((0, 448.0, torch.float8_e4m3fn), (1, 57344.0, torch.float8_e5m2)):
        t = R.fp8_quantize(x, fmt)
        amax, sc, dq = t.state.tolist()
        assert amax == float(x.float().abs().max()) and abs(sc * dq - 1.0) < 1e-6
        deq = t.q.view(dt).float() * dq
        assert float(deq.abs().max()) == pytest.approx(amax, rel=1e-6)              # the largest element maps to FMAX exactly
        assert Hh.rel_l2(deq.numpy(), x.float().numpy()) < (0.04 if fmt == 0 else 0.08)
        assert t.qt.shape == (48, 48) and torch.equal(t.qt[:, :37], t.q.t()) and not t.qt[:, 37:].any()
    # GEMM on the quantised bytes = matmul of the dequantised values
    a, b = R.fp8_quantize(x, 1, transposed=False), R.fp8_quantize(x[:20], 0, transposed=False)
    y = R.gemm_fp8_nt(a.q, a.dequant, b.q, b.dequant, 1)
    ref = (a.q.view(torch.float8_e5m2).float() * a.state[2]) @ (b.q.view(torch.float8_e4m3fn).float() * b.state[2]).t()
    assert Hh.rel_l2(y.float().numpy(), ref.numpy()) < 5e-3


def test_weight_copies_follow_the_accumulation_window(cpu_backend):
    """The e4m3 weight copies are re-made at the first micro-batch of every accumulation window (an optimizer step may have changed the
    weights) and reused inside it; a caller that is not the trainer always gets fresh copies."""
    from mantis_amd.trainer import MantisHipTrainer
    from oracle import ops_ref as R
    z = Hh.load_case("qwen2vl_b2_rightpad")
    model = Hh.build_qwen2vl_product("cpu").set_precision("fp8")
    calls = []
    orig = R.fp8_quantize

    def counting(x, *a, **kw):
        if x.data_ptr() >= model.arena.data_ptr() and x.data_ptr() < model.arena.data_ptr() + model.arena.numel() * 2:
            calls.append(1)
        return orig(x, *a, **kw)
    cpu_backend.K.fp8_quantize = counting
    try:
        tr = MantisHipTrainer(model, gradient_accumulation_steps=2)
        n_lin = 4 * model.config.text_config.num_hidden_layers
        tr.training_step(model, Hh.qwen2vl_batch(z))
        assert len(calls) == n_lin
        tr.training_step(model, Hh.qwen2vl_batch(z))            # 2nd micro-batch of the window: reused
        assert len(calls) == n_lin
        with torch.no_grad():
            model.arena.mul_(0.5)                               # "optimizer step"
        l3 = tr.training_step(model, Hh.qwen2vl_batch(z))       # new window: re-quantised
        assert len(calls) == 2 * n_lin and torch.isfinite(l3)
        model.engine.step_from_batch(Hh.qwen2vl_batch(z), compute_grads=False)
        assert len(calls) == 3 * n_lin
    finally:
        cpu_backend.K.fp8_quantize = orig


def test_set_precision_validates():
    model = Hh.build_qwen2vl_product("cpu")
    with pytest.raises(ValueError):
        model.set_precision("int4")
    assert model.set_precision("fp8").engine.w8 is not None and model.set_precision("bf16").engine.w8 is None
    assert model.set_precision("fp8_rowwise").engine.w8.rowwise and not model.set_precision("fp8").engine.w8.rowwise
