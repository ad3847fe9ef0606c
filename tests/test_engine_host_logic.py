"""CPU test of the PRODUCT's host logic (mantis_amd/engine.py, modeling_llava.py, trainer.py): the HIP operator backend
is replaced by the oracle's operator restatement (oracle/ops_ref.py) via monkeypatch, so what is under test is the kernel
sequencing, the save/recompute policy, the gradient-arena plumbing and the training_step contract -- checked against the
oracle model (pinned to the reference) and against the golden vectors recorded from the reference."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh


@pytest.fixture()
def cpu_backend(monkeypatch):
    import mantis_amd.engine as eng
    from oracle import ops_ref
    monkeypatch.setattr(eng, "K", ops_ref)
    return eng


CASES = ["siglip_b1_img1", "siglip_b1_img2_adjacent", "siglip_b1_img4", "siglip_b1_img_first_last",
         "siglip_b2_equal_rightpad", "siglip_b2_equal_nopad", "siglip_b2_unequal_quirk", "siglip_b1_text_only",
         "clip_b1_img2_adjacent", "clip_b2_equal_rightpad"]


@pytest.mark.parametrize("case", CASES)
def test_step_matches_oracle(cpu_backend, case):
    flavour = case.split("_")[0]
    z = Hh.load_case(case)
    model, meta, _ = Hh.build_product_model(flavour, "cpu")
    oracle = Hh.build_oracle_bf16_weights(flavour)
    overwrite = model._ensure_grad_arena()
    assert overwrite
    rec = {}
    out = model.engine.step(torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]),
                            torch.from_numpy(z["labels"]), Hh.pixels_list(z), compute_grads=True, overwrite_grads=True,
                            need_logits=True, record=rec)
    Hh.check_step_against_oracle(model, oracle, z, out, rec)
    # golden logits (fp32 reference on fp32 weights): loose check that nothing structural is off
    am = rec["merged_attention_mask"].bool().numpy()
    lg = out["logits"].float().numpy()
    assert Hh.rel_l2(lg[am], z["logits"][am]) < 0.08
    assert abs(float(out["loss"]) - float(z["loss"])) < 0.03 * float(z["loss"])
    for n, p in model.named_parameters():
        if n.startswith("vision_tower."):
            assert p.grad is None


@pytest.mark.parametrize("side", ["right", "left"])
def test_fix_unequal_counts_equals_reference_per_sample(cpu_backend, side):
    """LlavaConfig(fix_unequal_counts=True) on a batch with unequal image counts -> every sample gets what the
    reference computes for it alone at B = 1 (fixture of tests/golden/make_golden_fixcounts.py)."""
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    Hh.check_fixed_counts_step(model, side, "cpu")


def test_unequal_counts_without_the_flag_warns_once(cpu_backend):
    import warnings
    import mantis_amd.engine as eng
    z, _ = Hh.fixed_counts_case("right")
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    eng._WARNED_UNEQUAL = False
    args = (torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]), torch.from_numpy(z["labels"]), Hh.pixels_list(z))
    with pytest.warns(UserWarning, match="fix_unequal_counts"):
        model.engine.step(*args, compute_grads=False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model.engine.step(*args, compute_grads=False)          # second time: silent
    # left padding with unequal counts is placed correctly by the reference itself: no warning
    eng._WARNED_UNEQUAL = False
    zl, _ = Hh.fixed_counts_case("left")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model.engine.step(torch.from_numpy(zl["input_ids"]), torch.from_numpy(zl["attention_mask"]), torch.from_numpy(zl["labels"]),
                          Hh.pixels_list(zl), compute_grads=False)


@pytest.mark.parametrize("case", ["siglip_b2_equal_rightpad", "clip_b2_equal_rightpad"])
def test_projector_only_stage(cpu_backend, case):
    """The reference's pre-training stage tunes only multi_modal_projector (train_mllava.py:177-181): same training_step, every
    other parameter frozen.  Gradient arena = the projector alone; its gradients still match the oracle (the dX chain runs through
    the frozen decoder); nothing else receives a .grad."""
    flavour = case.split("_")[0]
    z = Hh.load_case(case)
    model, meta, _ = Hh.build_product_model(flavour, "cpu")
    oracle = Hh.build_oracle_bf16_weights(flavour)
    for n, p in model.named_parameters():
        if "multi_modal_projector" not in n:
            p.requires_grad = False
    assert model._ensure_grad_arena()
    # the arena's placement rule: sizes padded to 8 elements, starts aligned to 256 bytes (mantis_amd/arena.py ARENA_ALIGN)
    _, n_proj = model._place([(n, p.numel()) for n, p in model.named_parameters() if p.requires_grad])
    assert model.grad_arena.numel() == n_proj and all(o % 128 == 0 for o in model._grad_offs.values())
    rec = {}
    out = model.engine.step(torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]),
                            torch.from_numpy(z["labels"]), Hh.pixels_list(z), compute_grads=True, overwrite_grads=True,
                            need_logits=True, record=rec)
    rep = Hh.check_step_against_oracle(model, oracle, z, out, rec)
    assert set(rep) == {n for n, p in model.named_parameters() if p.requires_grad} and len(rep) == 4
    for n, p in model.named_parameters():
        if "multi_modal_projector" not in n:
            assert p.grad is None, n
    assert list(model.grad_buckets()) and sum(b.numel() for b in model.grad_buckets().values()) == n_proj


def test_count_mismatch_raises_value_error(cpu_backend):
    z = Hh.load_case("siglip_b1_img2_adjacent")
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    with pytest.raises(ValueError):
        model.engine.step(torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]),
                          torch.from_numpy(z["labels"]), [torch.from_numpy(z["pixel_values"])[:1]], compute_grads=False)


@pytest.mark.parametrize("ga", [1, 4])
def test_training_step_contract(cpu_backend, ga):
    """Returned value = loss/GA, detached 0-dim; gradients accumulate across micro-batches (HF:trainer.py:1892-1963)."""
    from mantis_amd.trainer import MantisHipTrainer
    z = Hh.load_case(f"siglip_training_step_ga{ga}")
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    tr = MantisHipTrainer(model, gradient_accumulation_steps=ga)
    losses = []
    for i in range(ga):
        batch = dict(input_ids=torch.from_numpy(z[f"mb{i}.input_ids"]), attention_mask=torch.from_numpy(z[f"mb{i}.attention_mask"]),
                     labels=torch.from_numpy(z[f"mb{i}.labels"]), pixel_values=Hh.pixels_list(z, f"mb{i}."))
        out = tr.training_step(model, batch)
        assert out.dim() == 0 and not out.requires_grad
        losses.append(float(out))
    assert np.allclose(losses, z["returned_losses"], rtol=2e-2)
    worst = 1.0
    for k in z.files:
        if k.startswith("grad."):
            g = model._param(k[5:]).grad.float().numpy()
            worst = min(worst, Hh.cosine(g, z[k]))
    assert worst > 0.98, worst


def test_autograd_bridge_matches_direct(cpu_backend):
    """model(**inputs).loss.backward() (stock-Trainer style) publishes the same gradients as the fused training_step."""
    z = Hh.load_case("siglip_b1_img2_adjacent")
    from mantis_amd.trainer import MantisHipTrainer
    m1, _, _ = Hh.build_product_model("siglip", "cpu")
    m2, _, _ = Hh.build_product_model("siglip", "cpu")
    batch = dict(input_ids=torch.from_numpy(z["input_ids"]), attention_mask=torch.from_numpy(z["attention_mask"]),
                 labels=torch.from_numpy(z["labels"]), pixel_values=Hh.pixels_list(z))
    l1 = MantisHipTrainer(m1, 2).training_step(m1, batch)
    out = m2(**batch)
    assert out.logits is None and out["loss"] is out.loss
    (out.loss / 2).backward()
    assert abs(float(l1) - float(out.loss) / 2) < 1e-6
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        if a.requires_grad:
            assert torch.allclose(a.grad.float(), b.grad.float(), atol=1e-3, rtol=2e-2), n


def test_eval_forward_returns_logits(cpu_backend):
    z = Hh.load_case("siglip_b1_img1")
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    model.eval()
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(z["input_ids"]), attention_mask=torch.from_numpy(z["attention_mask"]),
                    pixel_values=Hh.pixels_list(z), labels=torch.from_numpy(z["labels"]))
    assert out.logits.shape == z["logits"].shape
    assert abs(float(out.loss) - float(z["loss"])) < 0.03 * float(z["loss"])


def test_state_dict_names_match_reference():
    model, _, sd = Hh.build_product_model("clip", "cpu")
    own = {n for n, _ in model.named_parameters()}
    ref = {k for k in sd if ".head." not in k}
    assert own == ref


def test_stock_hf_trainer_subclass(cpu_backend, tmp_path):
    """`as_hf_trainer()` is a transformers.Trainer whose training_step is the fused one: constructed like the reference does
    (train_mllava.py:312-319), it returns the value the reference's Trainer returned for the same micro-batch (golden)."""
    transformers = pytest.importorskip("transformers")
    from mantis_amd.trainer import as_hf_trainer
    z = Hh.load_case("siglip_training_step_ga1")
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    args = transformers.TrainingArguments(output_dir=str(tmp_path), use_cpu=True, report_to=[], remove_unused_columns=False,
                                          gradient_accumulation_steps=1)
    trainer = as_hf_trainer()(model=model, args=args)
    trainer.current_gradient_accumulation_steps = 1
    batch = dict(input_ids=torch.from_numpy(z["mb0.input_ids"]), attention_mask=torch.from_numpy(z["mb0.attention_mask"]),
                 labels=torch.from_numpy(z["mb0.labels"]), pixel_values=Hh.pixels_list(z, "mb0."))
    out = trainer.training_step(model, batch)
    assert out.dim() == 0 and not out.requires_grad
    assert abs(float(out) - float(z["returned_losses"][0])) < 2e-2 * float(z["returned_losses"][0])
    assert model._param("multi_modal_projector.linear_1.weight").grad is not None


def _packed_from_case(z, rows):
    """Pack the given batch rows of a golden case (unpadded rows) into one row with pack_samples."""
    from mantis_amd.data import pack_samples
    pv = Hh.pixels_list(z)
    samples = [dict(input_ids=torch.from_numpy(z["input_ids"][[r]]), attention_mask=torch.from_numpy(z["attention_mask"][[r]]),
                    labels=torch.from_numpy(z["labels"][[r]]), pixel_values=pv[r]) for r in rows]
    return pack_samples(samples)


def test_pack_segments_oracle_matches_the_reference_4d_mask():
    """oracle/pack_ref.pack_segments: with no images the merged row is the packed row itself, so kstart / qend must reproduce the
    REFERENCE's block-diagonal 4-D mask (pack_batch_ref.npz, recorded from PackingDataset.pack_batch) and its position ids."""
    from oracle import pack_ref
    z = Hh.load_case("pack_batch_ref")
    for case in ("eq", "ragged"):
        n = int(z[f"{case}.n"])
        lens = [z[f"{case}.s{i}.input_ids"].shape[-1] for i in range(n)]
        ids = z[f"{case}.out.input_ids"]
        seg = np.repeat(np.arange(n), lens)[None]
        key = np.concatenate([z[f"{case}.s{i}.attention_mask"].reshape(-1) for i in range(n)])[None]
        plan = pack_ref.pack_plan(ids, key, None, 0, 1, -(2 ** 62), -1)
        ks, qe, pos, first = pack_ref.pack_segments(plan, ids, seg, 1, -(2 ** 62))
        S = ids.shape[1]
        q = np.arange(S)[:, None]
        k = np.arange(S)[None, :]
        dense = ((k >= ks[0][:, None]) & (k < qe[0][:, None]) & (key[0][None, :] != 0)).astype(np.int32)
        assert np.array_equal(dense, z[f"{case}.out.attention_mask"][0, 0])
        ref_pos = z[f"{case}.out.position_ids"]
        assert np.array_equal(pos[0][key[0] != 0], ref_pos[key[0] != 0])      # (masked keys sit at the end of their sample here)
        assert first[0].sum() == n


@pytest.mark.parametrize("case,rows", [("siglip_b2_equal_nopad", [0, 1]), ("siglip_b2_equal_nopad", [1, 0, 1])])
def test_packed_step_equals_the_separate_samples(cpu_backend, case, rows):
    """Sample packing through the product's host logic (plan segments, segment-bounded attention, per-sample positions, CE rows):
    loss and gradients equal those of running the samples separately (oracle `forward_packed` = unpack and run one by one)."""
    z = Hh.load_case(case)
    packed = _packed_from_case(z, rows)
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    oracle = Hh.build_oracle_bf16_weights("siglip")
    assert model._ensure_grad_arena()
    out = model.engine.step(packed["input_ids"], packed["key_mask"], packed["labels"], packed["pixel_values"], compute_grads=True,
                            overwrite_grads=True, segment_ids=packed["segment_ids"])
    oracle.zero_grad()
    oloss = oracle.forward_packed(packed["input_ids"], packed["pixel_values"], packed["segment_ids"], packed["labels"], packed["key_mask"])
    oloss.backward()
    assert abs(float(out["loss"]) - float(oloss)) <= 5e-3 * float(oloss), (float(out["loss"]), float(oloss))
    for name, p in model.named_parameters():
        if p.requires_grad:
            g, og = p.grad.float().numpy(), oracle.w[name].grad.numpy()
            assert Hh.cosine(g, og) > 0.995 and Hh.rel_l2(g, og) < 6e-2, (name, Hh.cosine(g, og), Hh.rel_l2(g, og))
    # the trainer accepts the reference's packed batch format (4-D mask + position ids) as well
    from mantis_amd.trainer import MantisHipTrainer
    m2, _, _ = Hh.build_product_model("siglip", "cpu")
    ref_fmt = dict(input_ids=packed["input_ids"], attention_mask=packed["attention_mask"], position_ids=packed["position_ids"],
                   labels=packed["labels"], pixel_values=packed["pixel_values"])
    l2 = MantisHipTrainer(m2, 1).training_step(m2, ref_fmt)
    assert abs(float(l2) - float(out["loss"])) < 1e-6


@pytest.mark.parametrize("case", ["siglip_b2_equal_rightpad", "siglip_b1_img4"])
def test_activation_checkpointing_changes_memory_not_results(cpu_backend, case, monkeypatch):
    """`--gradient_checkpointing True` of the reference's launch script (/root/reference/mantis/train/scripts/train_mllava.sh:168; HF
    re-runs every LlamaDecoderLayer in the backward).  Here: `model.gradient_checkpointing_enable()` keeps ONE tensor per decoder layer (its
    input) instead of eleven and runs `decoder.layer_forward` again in the backward -- loss and every gradient must come out identical,
    bit for bit, and the forward must indeed have kept less."""
    import mantis_amd.decoder as D
    z = Hh.load_case(case)
    args = (torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]), torch.from_numpy(z["labels"]), Hh.pixels_list(z))
    kept = []
    real_forward = D.decoder_forward

    def spy(*a, **kw):
        x, ctx = real_forward(*a, **kw)
        kept.append([len(e) for e in ctx["saved"]])
        return x, ctx
    monkeypatch.setattr(D, "decoder_forward", spy)
    res = []
    for on in (False, True):
        model, _, _ = Hh.build_product_model("siglip", "cpu")
        assert type(model).supports_gradient_checkpointing and not model.is_gradient_checkpointing
        if on:
            model.gradient_checkpointing_enable(gradient_checkpointing_kwargs={"use_reentrant": False})      # HF's call (trainer.py)
            assert model.is_gradient_checkpointing
        model._ensure_grad_arena()
        out = model.engine.step(*args, compute_grads=True, overwrite_grads=True)
        res.append((float(out["loss"]), model.grad_arena.clone()))
        model.gradient_checkpointing_disable()
        assert not model.is_gradient_checkpointing
    n_layers = len(kept[0])
    assert kept[0] == [11] * n_layers and kept[1] == [1] * n_layers, kept
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1]) and float(res[0][1].float().abs().sum()) > 0


def test_activation_checkpointing_is_skipped_without_a_backward(cpu_backend):
    """An evaluation forward (compute_grads=False) keeps nothing either way."""
    z = Hh.load_case("siglip_b1_img1")
    model, _, _ = Hh.build_product_model("siglip", "cpu")
    model.gradient_checkpointing_enable()
    out = model.engine.step(torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"]), torch.from_numpy(z["labels"]),
                            Hh.pixels_list(z), compute_grads=False)
    assert out["loss"] is not None
