"""Pin the oracle (oracle/llava_ref.py, oracle/pack_ref.py) against outputs recorded from the reference itself
(tests/golden/*.npz, written by tests/golden/make_golden.py which imports /root/reference in the build container).

Tolerances (fp32 restatement vs fp32 reference): integers exact; logits atol 1e-5;
loss rtol 1e-6 (+1e-6 abs); grads rtol 1e-4 relative-L2."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import pack_ref
from oracle.llava_ref import LlavaRef

G = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(G, "*.npz"))
               if not os.path.basename(p).startswith(("weights_", "label_rule", "siglip_training_step", "cfg1_", "collate_ref", "collate_qwen_ref", "pack_batch_ref", "idefics2_", "qwen2vl_",
                                                        "siglip_b2_unequal_fixed")))


def _pixels(z):
    if "pixel_values" not in z.files:
        return None
    counts = z["pixel_counts"]
    pv = torch.from_numpy(z["pixel_values"])
    return list(torch.split(pv, counts.tolist()))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("case", CASES)
def test_forward_backward_matches_reference(case):
    flavour = case.split("_")[0]
    z = np.load(os.path.join(G, case + ".npz"))
    m = LlavaRef.from_npz(os.path.join(G, f"weights_{flavour}.npz"))
    rec = {}
    loss, logits = m.forward(z["input_ids"], _pixels(z), z["attention_mask"], z["labels"], record=rec)
    loss.backward()
    # integers: exact
    for k in ("merged_attention_mask", "merged_labels", "merged_position_ids"):
        if k in z.files:
            assert np.array_equal(rec[k].numpy(), z[k]), k
    if "merged_embeds" in z.files:
        assert np.array_equal(rec["merged_embeds"].detach().numpy(), z["merged_embeds"]) or \
            np.allclose(rec["merged_embeds"].detach().numpy(), z["merged_embeds"], atol=1e-6)
    for k in ("projector_in", "projector_out"):
        if k in z.files:
            assert np.allclose(rec[k].detach().numpy(), z[k], atol=2e-5), k
    # decoder activations are compared where the merged mask is 1 (fully-masked query rows are don't-care)
    if "merged_attention_mask" in z.files:
        am = z["merged_attention_mask"].astype(bool)
    else:
        am = z["attention_mask"].astype(bool)
    for k in [f for f in z.files if f.startswith("llm_")]:
        assert np.allclose(rec[k].detach().numpy()[am], z[k][am], atol=3e-5), k
    assert np.allclose(logits.detach().numpy()[am], z["logits"][am], atol=1e-5 * max(1.0, np.abs(z["logits"]).max()))
    assert abs(float(loss) - float(z["loss"])) <= 1e-6 * abs(float(z["loss"])) + 1e-6
    ngrads = 0
    for k in z.files:
        if not k.startswith("grad."):
            continue
        g = m.w[k[5:]].grad
        assert g is not None, k
        assert rel_l2(g.numpy(), z[k]) < 1e-4, (k, rel_l2(g.numpy(), z[k]))
        ngrads += 1
    assert ngrads >= 20
    for k, v in m.w.items():
        if k.startswith("vision_tower."):
            assert v.grad is None        # frozen tower, train_mllava.py:240-242


def test_count_mismatch_raises_value_error():
    z = np.load(os.path.join(G, "siglip_b1_img2_adjacent.npz"))
    m = LlavaRef.from_npz(os.path.join(G, "weights_siglip.npz"))
    pv = torch.from_numpy(z["pixel_values"])[:1]
    with pytest.raises(ValueError):
        m.forward(z["input_ids"], [pv], z["attention_mask"], z["labels"])


@pytest.mark.parametrize("ga", [1, 4])
def test_training_step_matches_reference(ga):
    z = np.load(os.path.join(G, f"siglip_training_step_ga{ga}.npz"))
    m = LlavaRef.from_npz(os.path.join(G, "weights_siglip.npz"))
    m.zero_grad()
    losses = []
    for i in range(ga):
        counts = z[f"mb{i}.pixel_counts"].tolist()
        batch = dict(input_ids=z[f"mb{i}.input_ids"], attention_mask=z[f"mb{i}.attention_mask"], labels=z[f"mb{i}.labels"],
                     pixel_values=list(torch.split(torch.from_numpy(z[f"mb{i}.pixel_values"]), counts)))
        out = m.training_step(batch, gradient_accumulation_steps=ga)
        assert out.dim() == 0 and not out.requires_grad
        losses.append(float(out))
    assert np.allclose(losses, z["returned_losses"], rtol=1e-6)
    for k in z.files:
        if k.startswith("grad."):
            assert rel_l2(m.w[k[5:]].grad.numpy(), z[k]) < 1e-4, k


def test_label_rule_matches_reference_fixture():
    z = np.load(os.path.join(G, "label_rule.npz"))
    sep, img = int(z["sep_id"]), int(z["image_id"])
    n = len([k for k in z.files if k.endswith(".ids")])
    assert n >= 5
    for c in range(n):
        ids = z[f"c{c}.ids"]
        assert np.array_equal(pack_ref.llama3_label_mask(ids, sep), z[f"c{c}.llama3"])
        assert np.array_equal(pack_ref.plain_label_mask(ids, img), z[f"c{c}.plain"])


def test_pack_rows_copy_is_bit_exact():
    z = np.load(os.path.join(G, "siglip_b2_unequal_quirk.npz"))
    m = LlavaRef.from_npz(os.path.join(G, "weights_siglip.npz"))
    emb = m.w["language_model.model.embed_tokens.weight"].detach().numpy()[z["input_ids"]]
    plan = pack_ref.pack_plan(z["input_ids"], z["attention_mask"], z["labels"], 3, 16, 298, 299)
    out = pack_ref.pack_rows(plan, emb, z["projector_out"])
    assert np.array_equal(out, z["merged_embeds"])
    assert np.array_equal(plan["attention_mask"], z["merged_attention_mask"])
    assert np.array_equal(plan["position_ids"], z["merged_position_ids"])
    assert np.array_equal(plan["labels"], z["merged_labels"])


def _valid_span(side, b, L, Lb):
    """columns of merged row b that hold the sample's own B = 1 sequence (length Lb) in an L-column batch row"""
    return slice(0, Lb) if side == "right" else slice(L - Lb, L)


@pytest.mark.parametrize("side", ["right", "left"])
def test_fixed_placement_equals_reference_at_batch_size_one(side):
    """with `fix_unequal_counts` a batch with UNEQUAL image counts gives, sample by sample, what the reference
    returns for that sample alone at B = 1 (tests/golden/make_golden_fixcounts.py) -- integers exact, logits / loss / gradients
    at the fp32 bars of this file; the loss of the batch is the label-count-weighted mean of the two B = 1 losses."""
    z = np.load(os.path.join(G, "siglip_b2_unequal_fixed.npz"))
    m = LlavaRef.from_npz(os.path.join(G, "weights_siglip.npz"))
    m.cfg = dict(m.cfg, fix_unequal_counts=True)
    ids, am, lab = z[f"{side}.input_ids"], z[f"{side}.attention_mask"], z[f"{side}.labels"]
    px = list(torch.split(torch.from_numpy(z["pixel_values"]), z["pixel_counts"].tolist()))
    rec = {}
    loss, logits = m.forward(ids, px, am, lab, record=rec)
    loss.backward()
    L = rec["merged_attention_mask"].shape[1]
    n_lab, want_loss = [], 0.0
    for b in range(2):
        Lb = z[f"s{b}.merged_attention_mask"].shape[1]
        sp = _valid_span(side, b, L, Lb)
        assert np.array_equal(rec["merged_attention_mask"].numpy()[b, sp], z[f"s{b}.merged_attention_mask"][0])
        assert np.array_equal(rec["merged_labels"].numpy()[b, sp], z[f"s{b}.merged_labels"][0])
        assert np.array_equal(rec["merged_position_ids"].numpy()[b, sp], z[f"s{b}.merged_position_ids"][0])
        rest = np.ones(L, bool)
        rest[sp] = False                                 # everything outside the sample's own span is padding
        assert not rec["merged_attention_mask"].numpy()[b, rest].any()
        assert (rec["merged_labels"].numpy()[b, rest] == -100).all()
        ref_logits = z[f"s{b}.logits"][0]
        assert np.allclose(logits.detach().numpy()[b, sp], ref_logits, atol=1e-5 * max(1.0, np.abs(ref_logits).max())), b
        n = int((z[f"s{b}.merged_labels"][0, 1:] != -100).sum())
        n_lab.append(n)
        want_loss += n * float(z[f"s{b}.loss"])
    want_loss /= sum(n_lab)
    assert abs(float(loss) - want_loss) <= 1e-6 * abs(want_loss) + 1e-6
    grads = {k: v.grad for k, v in m.trainable().items() if v.grad is not None}
    checked = 0
    for k in [f[len("s0.grad."):] for f in z.files if f.startswith("s0.grad.")]:
        want = (n_lab[0] * z["s0.grad." + k] + n_lab[1] * z["s1.grad." + k]) / sum(n_lab)
        if k in grads:
            assert rel_l2(grads[k].numpy(), want) < 1e-4, k
            checked += 1
    assert checked >= 20


def test_reference_placement_differs_on_the_right_padded_unequal_batch():
    """the default (reference-exact) plan mis-places sample 1's image rows on the same batch: the flag is not a no-op"""
    z = np.load(os.path.join(G, "siglip_b2_unequal_fixed.npz"))
    a = (z["right.input_ids"], z["right.attention_mask"], z["right.labels"], 3, 16, 298, 299)
    ref, fixed = pack_ref.pack_plan(*a), pack_ref.pack_plan(*a, fix_unequal_counts=True)
    assert not np.array_equal(ref["src_kind"], fixed["src_kind"])
    assert np.array_equal(ref["src_kind"][0], fixed["src_kind"][0])          # the sample with the most images is placed alike
    # equal counts: the two formulations agree
    e = np.load(os.path.join(G, "siglip_b2_equal_rightpad.npz"))
    b = (e["input_ids"], e["attention_mask"], e["labels"], 4, 16, 298, 299)
    p0, p1 = pack_ref.pack_plan(*b), pack_ref.pack_plan(*b, fix_unequal_counts=True)
    for k in ("src_kind", "src_idx", "attention_mask", "labels", "position_ids"):
        assert np.array_equal(p0[k], p1[k]), k


def test_oracle_matches_cfg1_reference_fixture():
    """cfg1 = BASELINE.json configs[0] at full size (SigLIP-base/16-224 + Llama-68M, 1 image, 128 tokens): the oracle reproduces the
    loss, logits statistics and every per-parameter gradient norm the REFERENCE's Trainer.training_step produced on the same
    seeded weights and batch (tests/golden/make_golden_cfg1.py)."""
    import json
    from oracle.llava_ref import LlavaRef
    from tests import helpers as Hh
    meta, w, z, f = Hh.load_cfg1()
    model = LlavaRef(w, meta)
    model.zero_grad()
    loss, logits = model.forward(z["input_ids"], [torch.from_numpy(z["pixel_values"])], z["attention_mask"], z["labels"])
    loss.backward()
    assert abs(float(loss) - float(f["loss"])) <= 2e-6 * float(f["loss"]) and float(f["loss"]) == pytest.approx(float(f["returned_loss"]), rel=1e-6)
    assert tuple(logits.shape) == tuple(f["logits_shape"])
    lg = logits.detach().double()
    assert abs(float(lg.abs().mean()) - float(f["logits_abs_mean"])) <= 1e-5 * float(f["logits_abs_mean"])
    assert np.allclose(lg[0, -1, :64].numpy(), f["logits_row0"], atol=2e-5)
    names = json.loads(str(f["grad_names"]))
    for n, want in zip(names, f["grad_norms"]):
        got = float(model.w[n].grad.double().norm())
        assert abs(got - want) <= 2e-4 * want + 1e-9, (n, got, want)
    assert {n for n in model.w if model.w[n].grad is not None} == set(names)
