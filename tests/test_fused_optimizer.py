"""`optim.FusedAdamW` as the optimizer the REFERENCE's own loop can drive (VERDICT r03 missing 3):
a `torch.optim.Optimizer` whose single param_group's lr is read every step (cosine + warm-up of
/root/reference/mantis/train/scripts/train_mllava.sh:162-165), with state_dict / load_state_dict for the auto-resume of
/root/reference/mantis/train/train_mllava.py:281-294, and the HF loop's clip_grad_norm_ routed to the fused norm.

CPU: the product's host logic with the oracle operators in place of the HIP backend (pinned to torch.optim.AdamW +
clip_grad_norm_ by tests/test_advice_regressions.py::test_adamw_and_clip_oracle_match_torch); the `-m gpu` twins are
`fused_optimizer_*` in tests/gpu_checks.py."""
import math

import numpy as np
import pytest
import torch

from tests import helpers as Hh


@pytest.fixture()
def cpu_backend(monkeypatch):
    import mantis_amd.engine as eng
    import mantis_amd.optim as opt
    from oracle import ops_ref
    monkeypatch.setattr(eng, "K", ops_ref)
    monkeypatch.setattr(opt, "K", ops_ref)
    return eng


def test_schedule_drives_lr_and_matches_torch_adamw(cpu_backend):
    Hh.check_fused_optimizer_vs_torch("cpu", steps=5)


def test_state_dict_round_trip_resumes_the_same_trajectory(cpu_backend, tmp_path):
    Hh.check_fused_optimizer_resume("cpu", str(tmp_path))


def test_is_a_torch_optimizer_with_one_param_group(cpu_backend):
    from mantis_amd.optim import FusedAdamW
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    opt = FusedAdamW(model, lr=3e-4, betas=(0.8, 0.95), eps=1e-7, weight_decay=0.1)
    assert isinstance(opt, torch.optim.Optimizer) and len(opt.param_groups) == 1
    g = opt.param_groups[0]
    assert g["lr"] == 3e-4 and tuple(g["betas"]) == (0.8, 0.95) and g["eps"] == 1e-7 and g["weight_decay"] == 0.1
    assert {id(p) for p in g["params"]} == {id(p) for p in model.parameters() if p.requires_grad}
    with pytest.raises(NotImplementedError):
        opt.add_param_group({"params": [torch.nn.Parameter(torch.zeros(3))]})
    with pytest.raises(ValueError):
        opt.load_state_dict(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(3))]).state_dict())


def test_no_decay_exempts_the_named_parameters(cpu_backend):
    """weight decay with HF's exemption rule (biases / norm weights): a zero gradient leaves exempt parameters untouched and shrinks
    the others by (1 - lr * wd) per step"""
    from mantis_amd.optim import FusedAdamW
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    opt = FusedAdamW(model, lr=0.5, weight_decay=0.5, max_grad_norm=None, no_decay=lambda n: model._param(n).dim() <= 1)
    assert len(opt._segments) > 3 and {s[3] for s in opt._segments} == {True, False}
    before = {n: model._param(n).detach().float().clone() for n in opt._names}
    model.grad_arena.zero_()
    opt.step()
    for n in opt._names:
        got = model._param(n).detach().float()
        if before[n].dim() <= 1:
            assert torch.equal(got, before[n]), n
        else:
            assert torch.allclose(got, (before[n] * 0.75).to(torch.bfloat16).float(), atol=0, rtol=1e-2), n


def test_hf_trainer_builds_the_fused_optimizer_and_routes_clipping(cpu_backend, tmp_path):
    """The stock loop's pieces around training_step (HF trainer.py:1785-1796): create_optimizer_and_scheduler -> FusedAdamW +
    cosine / warm-up scheduler; _clip_grad_norm -> the fused norm; optimizer.step(); lr_scheduler.step(); _save / _load of the
    optimizer state."""
    transformers = pytest.importorskip("transformers")
    from mantis_amd.trainer import as_hf_trainer
    from mantis_amd.optim import FusedAdamW
    z = Hh.load_case("siglip_training_step_ga4")
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    args = transformers.TrainingArguments(output_dir=str(tmp_path), use_cpu=True, report_to=[], remove_unused_columns=False,
                                          gradient_accumulation_steps=1, learning_rate=1e-3, weight_decay=0.0, max_grad_norm=1.0,
                                          lr_scheduler_type="cosine", warmup_steps=2)
    tr = as_hf_trainer()(model=model, args=args)
    tr.create_optimizer_and_scheduler(num_training_steps=10)
    assert isinstance(tr.optimizer, FusedAdamW)
    lrs, norms = [], []
    for i in range(3):
        b = dict(input_ids=torch.from_numpy(z[f"mb{i}.input_ids"]), attention_mask=torch.from_numpy(z[f"mb{i}.attention_mask"]),
                 labels=torch.from_numpy(z[f"mb{i}.labels"]), pixel_values=Hh.pixels_list(z, f"mb{i}."))
        tr.accelerator.gradient_state._set_sync_gradients(True)
        tr.training_step(model, b)
        want = float(torch.cat([p.grad.float().reshape(-1) for p in model.parameters() if p.grad is not None]).norm())
        norm = tr._clip_grad_norm(model)
        assert abs(float(norm) - want) <= 1e-4 * want
        norms.append(float(norm))
        lrs.append(tr.optimizer.param_groups[0]["lr"])
        tr.optimizer.step()
        tr.lr_scheduler.step()
        model.zero_grad()
    assert np.allclose(lrs, [0.0, 0.5e-3, 1e-3]) and tr.optimizer.step_count == 3      # warm-up 0 -> 1e-3 over two steps
    # the accelerator's own entry (older transformers clip inline through it) goes to the fused norm as well
    tr.accelerator.gradient_state._set_sync_gradients(True)
    tr.training_step(model, b)
    n2 = tr.accelerator.clip_grad_norm_(model.parameters(), 1.0)
    assert n2.dim() == 0 and float(n2) > 0 and tr.optimizer._pending_scale is not None
    tr._save_optimizer_and_scheduler(str(tmp_path))
    sd = torch.load(str(tmp_path / "optimizer.pt"), weights_only=True)
    assert sd["format"] == "mantis_fused_adamw/3" and sd["step"] == 3


def test_folded_norm_range_arithmetic_and_fallback(cpu_backend):
    """FusedAdamW.begin_fold / end_fold: the global sum of squares = tile partials of the GEMM-covered gradient ranges + a plain
    sum-of-squares pass over everything else.  (1) CPU backend: no GEMM folds anything -> the whole arena goes through the plain pass
    and the norm equals the unfolded one; (2) coverage simulated for the largest weight gradients: same norm."""
    from mantis_amd.optim import FusedAdamW
    from mantis_amd.trainer import MantisHipTrainer
    z = Hh.load_case("siglip_training_step_ga1")
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    opt = FusedAdamW(model, lr=1e-3, max_grad_norm=1.0)
    tr = MantisHipTrainer(model, 1, fold_norm_into=opt)
    b = dict(input_ids=torch.from_numpy(z["mb0.input_ids"]), attention_mask=torch.from_numpy(z["mb0.attention_mask"]),
             labels=torch.from_numpy(z["mb0.labels"]), pixel_values=Hh.pixels_list(z, "mb0."))
    tr.training_step(model, b)
    assert opt._norm_ready and opt.folded_tiles == 0
    want = float(model.grad_arena.float().pow(2).sum().sqrt())
    n1 = float(opt.clip_grad_norm(1.0))
    assert abs(n1 - want) <= 1e-5 * want
    # simulated coverage: pretend the fused GEMM produced the partials of every 2-D gradient (one "tile" each)
    f = opt.begin_fold()
    for name in opt._names:
        g = model._param(name).grad
        if g.dim() == 2:
            ptr = f.take(g, 1)
            assert ptr is not None
            idx = (ptr - opt._fold_ws.data_ptr()) // 4
            opt._fold_ws[idx] = g.float().pow(2).sum()
    one_d = next(model._param(n).grad for n in opt._names if model._param(n).dim() == 1)
    assert f.take(one_d, 1) is not None
    f.give_back(one_d, 1)                                     # a declined shape leaves no trace
    opt.end_fold()
    assert opt.folded_tiles == sum(1 for n in opt._names if model._param(n).dim() == 2)
    n2 = float(opt.clip_grad_norm(1.0))
    assert abs(n2 - want) <= 1e-5 * want
    # a gradient written TWICE in one backward (a tied / shared weight, a chunked dW loop): the second fused launch is declined, the first
    # launch's partials -- now stale -- do not count, and the range goes through the separate pass (round-4 advisor finding)
    f = opt.begin_fold()
    twice = None
    for name in opt._names:
        g = model._param(name).grad
        if g.dim() == 2:
            ptr = f.take(g, 1)
            idx = (ptr - opt._fold_ws.data_ptr()) // 4
            opt._fold_ws[idx] = g.float().pow(2).sum()
            if twice is None:
                twice = g
                opt._fold_ws[idx] = 12345.0                   # what a first, partial write would have left: must not reach the norm
    assert f.take(twice, 1) is None and f.take(twice, 1) is None
    opt.end_fold()
    n3 = float(opt.clip_grad_norm(1.0))
    assert abs(n3 - want) <= 1e-5 * want
    # ... and a norm taken for a step that never happens dies with the gradients
    opt.begin_fold()
    opt.end_fold()
    assert opt._norm_ready
    opt.zero_grad(set_to_none=False)
    assert not opt._norm_ready


def test_flat_state_is_defined_in_the_alignment_pads(cpu_backend, monkeypatch):
    """The arenas align every parameter to 256 bytes; the pads between two optimizer segments (where the weight-decay exemption changes)
    are written by no kernel, so the flat fp32 state must be born defined there: with `torch.empty` the master copy held whatever the
    allocator handed back -- invisible in a fresh process, a 5 % mismatch of `state_dict()["master"]` after other work (round 4)."""
    from mantis_amd.optim import FusedAdamW
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    real_empty = torch.empty

    def poisoned_empty(*a, **kw):
        t = real_empty(*a, **kw)
        return t.fill_(float("nan")) if t.is_floating_point() else t
    monkeypatch.setattr(torch, "empty", poisoned_empty)
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.1, no_decay=lambda n: model._param(n).dim() <= 1)
    monkeypatch.setattr(torch, "empty", real_empty)
    covered = torch.zeros(model.grad_arena.numel(), dtype=torch.bool)
    for _, g_off, cnt, _ in opt._segments:
        covered[g_off:g_off + cnt] = True
    assert not bool(covered.all()), "this model is expected to have pads between segments"
    for name in ("master", "master_lo", "exp_avg", "exp_avg_sq"):
        t = getattr(opt, name).float()
        assert bool(torch.isfinite(t).all()) and float(t[~covered].abs().sum()) == 0.0, name
    assert float(opt.state_dict()["master_hi"].float()[~covered].abs().sum()) == 0.0


def test_split_master_restatement_round_trips_every_low_half():
    """The storage of the round-5 optimizer (include/mantis_hip.h: mantis_adamw_split): fp32 master = (bf16 parameter, low 16 bits, tie
    bit in the sign of exp_avg_sq).  The oracle's integer restatement of join / split, over every low-half pattern, ties of both
    parities, signed zeros, denormals and the largest finite value: split gives bf16(master) exactly, join gives the master back bit for
    bit, and WITHOUT the tie bit exactly the ties that rounded up come back wrong (why 32 stored bits are not enough)."""
    from oracle import ops_ref as R
    n = 1 << 17
    master = torch.randn(n, generator=torch.Generator().manual_seed(0)) * 0.05
    b = master.view(torch.int32)
    b[:65536] = (b[:65536] & ~0xFFFF) | torch.arange(65536, dtype=torch.int32)
    b[65536:69632] = (b[65536:69632] & ~0xFFFF) | 0x8000
    master[70000:70006] = torch.tensor([0.0, -0.0, 1e-42, -1e-42, 3.3895e38, -3.3895e38])
    p, lo, v = torch.zeros(n, dtype=torch.bfloat16), torch.zeros(n, dtype=torch.int16), torch.rand(n)
    v0 = v.clone()
    R.master_split(master, p, lo, v)
    assert torch.equal(p.view(torch.int16), master.to(torch.bfloat16).view(torch.int16))
    assert torch.equal(v.abs(), v0)
    ties = v.view(torch.int32) < 0
    up_ties = ((b & 0xFFFF) == 0x8000) & (((b >> 16) & 1) == 1)
    assert torch.equal(ties, up_ties) and int(ties.sum()) > 1000
    assert torch.equal(R.master_join(p, lo, v).view(torch.int32), b)
    wrong = R.master_join(p, lo, v.abs()).view(torch.int32) != b
    assert torch.equal(wrong, up_ties)


def test_split_master_step_equals_the_fp32_master_step():
    """adamw_split_flat == adamw_flat on the joined master, bit for bit, over 5 steps with weight decay (oracle operators; the HIP twin is
    the GPU check `adamw_split_bitwise`)."""
    from oracle import ops_ref as R
    n = 1 << 14
    gen = torch.Generator().manual_seed(3)
    master = torch.randn(n, generator=gen) * 0.05
    m1, v1, mast, p1 = torch.zeros(n), torch.zeros(n), master.clone(), master.to(torch.bfloat16)
    m2, v2, p2, lo2 = torch.zeros(n), torch.zeros(n), torch.zeros(n, dtype=torch.bfloat16), torch.zeros(n, dtype=torch.int16)
    R.master_split(master, p2, lo2, v2)
    for step in range(1, 6):
        g = (torch.randn(n, generator=gen) * 0.01).to(torch.bfloat16)
        R.adamw_flat(p1, g, mast, m1, v1, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
        R.adamw_split_flat(p2, g, lo2, m2, v2, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
        assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2.abs())
        assert torch.equal(R.master_join(p2, lo2, v2).view(torch.int32), mast.view(torch.int32))


def test_round4_checkpoint_with_fp32_masters_still_loads(cpu_backend):
    """A `mantis_fused_adamw/2` state (one flat fp32 `master` array) resumes on the split-master optimizer: same parameters, same joined
    masters, same moments as the optimizer it was exported from, and the same next step."""
    from mantis_amd.optim import FusedAdamW, FP32_MASTER_FORMAT

    def fresh():
        model, _, _ = Hh.build_product_model("siglip", "cpu")
        return model, FusedAdamW(model, lr=1e-2, weight_decay=0.05, max_grad_norm=None)
    gen = torch.Generator().manual_seed(5)
    m_a, o_a = fresh()
    grads = [(torch.randn(m_a.grad_arena.numel(), generator=gen) * 0.02).to(torch.bfloat16) for _ in range(3)]
    for g in grads[:2]:
        m_a.grad_arena.copy_(g)
        o_a.step()
    old = o_a.export_fp32_state()            # plain fp32 master / exp_avg / exp_avg_sq arrays, the round-4 layout
    assert old["format"] == FP32_MASTER_FORMAT and old["master"].dtype == torch.float32 and float(old["exp_avg_sq"].min()) >= 0.0
    old = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in old.items()}
    m_b, o_b = fresh()
    o_b.load_state_dict(old)
    assert torch.equal(m_b.arena, m_a.arena)
    for k in ("master_lo", "exp_avg", "exp_avg_sq", "master"):
        assert torch.equal(getattr(o_b, k), getattr(o_a, k)), k
    for m, o in ((m_a, o_a), (m_b, o_b)):
        m.grad_arena.copy_(grads[2])
        o.step()
    assert torch.equal(m_b.arena, m_a.arena) and torch.equal(o_b.master, o_a.master) and o_b.step_count == 3
