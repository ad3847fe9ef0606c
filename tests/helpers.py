"""Shared test helpers (test infrastructure)."""
import json
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    return np.load(os.path.join(G, name + ".npz"))


def golden_cfg_and_weights(flavour):
    z = np.load(os.path.join(G, f"weights_{flavour}.npz"))
    meta = json.loads(str(z["__config__"]))
    return meta, {k: z[k] for k in z.files if k != "__config__"}


def build_product_model(flavour, device):
    """The product module (flat bf16 arena) loaded with the golden weights rounded to bf16."""
    from mantis_amd.configuration_llava import LlavaConfig
    from mantis_amd.modeling_llava import LlavaForConditionalGeneration
    meta, sd = golden_cfg_and_weights(flavour)
    model = LlavaForConditionalGeneration(LlavaConfig.from_oracle_meta(meta), device=device, init=None)
    model.load_reference_state_dict(sd)
    return model, meta, sd


def build_oracle_bf16_weights(flavour):
    """Oracle model (fp32 math) on the SAME bf16-rounded weights the product model holds."""
    from oracle.llava_ref import LlavaRef
    meta, sd = golden_cfg_and_weights(flavour)
    sd = {k: torch.from_numpy(v).to(torch.bfloat16).float() for k, v in sd.items()}
    return LlavaRef(sd, meta)


def pixels_list(z, prefix=""):
    if prefix + "pixel_values" not in z.files:
        return None
    pv = torch.from_numpy(z[prefix + "pixel_values"])
    return list(torch.split(pv, z[prefix + "pixel_counts"].tolist()))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


#: (case name, {parameter: (cosine, rel-L2)}) of every whole-step comparison run in this process (tools/gpu_selftest.py sets CURRENT_CASE and
#: writes the per-tensor table: which gradients sit where against SURVEY 8c's proposed 0.999 / 2e-2)
GRAD_REPORTS = []
CURRENT_CASE = None


def _note_report(report):
    GRAD_REPORTS.append((CURRENT_CASE, dict(report)))


def write_grad_parity(path):
    """Every whole-step comparison of this process against SURVEY 8c's proposed bars, one row per case (tools/gpu_selftest.py; the -m gpu
    suite writes it at session end when MANTIS_CHECK_REPORT_DIR is set)."""
    with open(path, "w") as f:
        f.write("# whole-step gradient parity (bf16 HIP path vs fp32 oracle on the same bf16-rounded weights), per case: worst cosine / worst rel-L2 "
                "over the trainable parameters, and every parameter outside SURVEY 8c's proposed bars (cosine >= 0.999, rel-L2 <= 2e-2)\n\n"
                "| case | tensors | worst cosine | worst rel-L2 | outside 0.999 / 2e-2 |\n|---|---|---|---|---|\n")
        for case, rep in GRAD_REPORTS:
            if not rep:
                continue
            wc = min(c for c, _ in rep.values())
            wr = max(r for _, r in rep.values())
            out = [f"{n} ({c:.5f}, {r:.4f})" for n, (c, r) in rep.items() if c < 0.999 or r > 2e-2]
            f.write(f"| {case} | {len(rep)} | {wc:.6f} | {wr:.4f} | {'; '.join(out) if out else '-'} |\n")


# Whole-step gradient bars (bf16 product vs fp32 oracle on the same bf16-rounded weights): SURVEY.md 8c's proposal, cosine >= 0.999 and
# rel-L2 <= 2e-2 per tensor, since round 5 (rounds 1 - 4 ran 0.995 / 6e-2).  Measured on the MI355X over every golden case
# (profiles/r05_grad_parity.md): LLaVA worst 0.99993 / 0.012, Idefics2 0.99986 / 0.017; the exceptions are stated where they apply -- the Qwen2-VL
# key bias (a near-cancelling sum: up to 0.9990 / 0.045) and the q / k projections of the FULL-WIDTH steps (4096-wide contractions over 2812 -
# 4096 positions: 0.9993 / 0.037), which carry their own bars at the call sites.
GRAD_COS, GRAD_REL = 0.999, 2e-2


# Round 6 (verdict item 5): the exceptions above are no longer "measured + margin".  tests/golden/bf16_envelope.json records how far the REFERENCE'S
# OWN bf16 run sits from its fp32 run, per tensor: transformers' Qwen2VLForConditionalGeneration (the class the reference resolves to) on the
# tiny goldens + 24 random inputs, and the pinned oracle classes in bf16 at the full-width geometries of the `*_full_width_vs_oracle` checks
# (tests/golden/make_bf16_envelope.py).  A tensor's bar is then max(SURVEY's 2e-2, 1.5 x the reference's own bf16 deviation) and
# min(0.999, 1 - 1.5 x (1 - its cosine)): the product may be as far from fp32 as 1.5 x what the reference's own training arithmetic is.
ENVELOPE_FACTOR = 1.5
_ENVELOPE = None


def bf16_envelope(case):
    """{tensor: (cosine, rel-L2)} of the reference's bf16 run against its fp32 run for `case` (a key of tests/golden/bf16_envelope.json)"""
    global _ENVELOPE
    if _ENVELOPE is None:
        import json
        with open(os.path.join(G, "bf16_envelope.json")) as f:
            _ENVELOPE = json.load(f)
    return {k: (float(v[0]), float(v[1])) for k, v in _ENVELOPE[case].items() if not k.startswith("__")}


def envelope_bars(envelope, name, grad_cos=GRAD_COS, grad_rel=GRAD_REL):
    """(cosine bar, rel-L2 bar) of tensor `name`: SURVEY 8c's bars, relaxed to ENVELOPE_FACTOR x the reference's own bf16 deviation where that is
    larger; a tensor the envelope does not know keeps SURVEY's bars."""
    if not envelope or name not in envelope:
        return grad_cos, grad_rel
    c, r = envelope[name]
    return min(grad_cos, 1.0 - ENVELOPE_FACTOR * (1.0 - c)), max(grad_rel, ENVELOPE_FACTOR * r)


def check_step_against_oracle(model, oracle, z, out, rec, loss_rtol=5e-3, grad_cos=GRAD_COS, grad_rel=GRAD_REL, envelope=None):
    """bf16 product path vs fp32 oracle on identical (bf16-rounded) weights.  Tolerances: SURVEY.md section 8c."""
    orec = {}
    oracle.zero_grad()
    oloss, ologits = oracle.forward(z["input_ids"], pixels_list(z), z["attention_mask"], z["labels"], record=orec)
    oloss.backward()
    # integers bit-exact
    for k in ("merged_attention_mask", "merged_labels", "merged_position_ids"):
        if k in orec:
            assert np.array_equal(rec[k].cpu().numpy(), orec[k].numpy()), k
    loss = float(out["loss"].float().cpu().reshape(-1)[0])
    assert abs(loss - float(oloss)) <= loss_rtol * abs(float(oloss)), (loss, float(oloss))
    am = (orec["merged_attention_mask"] if "merged_attention_mask" in orec else torch.from_numpy(z["attention_mask"])).bool().numpy()
    for k in orec:
        if k.startswith("llm_layer") or k in ("projector_out", "merged_embeds"):
            a = rec[k].float().cpu().numpy()
            b = orec[k].detach().numpy()
            if k.startswith("llm_layer") or k == "merged_embeds":
                a, b = a[am], b[am]
            assert rel_l2(a, b) < 3e-2, (k, rel_l2(a, b))
    report = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        og = oracle.w[name].grad
        assert p.grad is not None, name
        g = p.grad.float().cpu().numpy()
        if og is None:          # parameter unused by this batch (projector on a text-only batch): exact zeros expected
            assert not g.any(), name
            continue
        c, r = cosine(g, og.numpy()), rel_l2(g, og.numpy())
        report[name] = (c, r)
        if np.linalg.norm(og.numpy()) < 1e-12:
            assert np.linalg.norm(g) < 1e-6, name
            continue
        cbar, rbar = envelope_bars(envelope, name, grad_cos, grad_rel)
        assert c >= cbar and r <= rbar, (name, c, r, cbar, rbar)
    _note_report(report)
    return report


def fixed_counts_case(side):
    """the unequal-image-count batch of tests/golden/siglip_b2_unequal_fixed.npz as a z-like dict (side = "right" | "left" padding)"""
    f = np.load(os.path.join(G, "siglip_b2_unequal_fixed.npz"))

    class Z(dict):
        files = property(lambda self: list(self.keys()))
    z = Z(input_ids=f[f"{side}.input_ids"], attention_mask=f[f"{side}.attention_mask"], labels=f[f"{side}.labels"],
          pixel_values=f["pixel_values"], pixel_counts=f["pixel_counts"])
    return z, f


def check_fixed_counts_step(model, side, device):
    """One step of the PRODUCT on the unequal-count batch with `fix_unequal_counts`: integers equal, sample by sample, what the
    reference recorded for that sample alone at B = 1; loss / activations / gradients against the oracle with the same flag (which
    tests/test_oracle_vs_golden.py pins to those B = 1 reference runs)."""
    z, f = fixed_counts_case(side)
    model.config.fix_unequal_counts = True
    oracle = build_oracle_bf16_weights("siglip")
    oracle.cfg = dict(oracle.cfg, fix_unequal_counts=True)
    assert model._ensure_grad_arena()
    rec = {}
    out = model.engine.step(torch.from_numpy(z["input_ids"]).to(device), torch.from_numpy(z["attention_mask"]).to(device),
                            torch.from_numpy(z["labels"]).to(device), [p.to(device) for p in pixels_list(z)], compute_grads=True,
                            overwrite_grads=True, need_logits=True, record=rec)
    check_step_against_oracle(model, oracle, z, out, rec)
    L = rec["merged_attention_mask"].shape[1]
    for b in range(2):
        Lb = f[f"s{b}.merged_attention_mask"].shape[1]
        sp = slice(0, Lb) if side == "right" else slice(L - Lb, L)
        for k in ("merged_attention_mask", "merged_labels", "merged_position_ids"):
            assert np.array_equal(rec[k].cpu().numpy()[b, sp], f[f"s{b}.{k}"][0]), (k, b)
        lg = out["logits"].float().cpu().numpy()[b, sp]
        assert rel_l2(lg, f[f"s{b}.logits"][0]) < 0.08, b       # bf16 product vs the reference's fp32 B = 1 logits
    return out


def _opt_batches(n):
    z = load_case("siglip_training_step_ga4")
    return [dict(input_ids=torch.from_numpy(z[f"mb{i % 4}.input_ids"]), attention_mask=torch.from_numpy(z[f"mb{i % 4}.attention_mask"]),
                 labels=torch.from_numpy(z[f"mb{i % 4}.labels"]), pixel_values=pixels_list(z, f"mb{i % 4}.")) for i in range(n)]


def check_fused_optimizer_vs_torch(device, steps=5):
    """`FusedAdamW` under a torch LR scheduler (linear warm-up + cosine: train_mllava.sh:162-165's shape) against torch.optim.AdamW +
    clip_grad_norm_ under the SAME scheduler class on fp32 copies, fed the product's bf16 gradients: lr per step equal, fp32 masters
    within 2e-5 relative, bf16 parameters = bf16(torch's fp32 parameters) up to one rounding.  Returns the worst master error."""
    from mantis_amd.trainer import MantisHipTrainer
    from mantis_amd.optim import FusedAdamW
    model, _, _ = build_product_model("siglip", device)
    tr = MantisHipTrainer(model, 1)
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    tp = {n: torch.nn.Parameter(model._param(n).detach().float().cpu().clone()) for n in names}
    topt = torch.optim.AdamW(list(tp.values()), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)

    def lam(s, warm=2, total=8):
        return s / warm if s < warm else 0.5 * (1.0 + np.cos(np.pi * (s - warm) / (total - warm)))
    sch, tsch = torch.optim.lr_scheduler.LambdaLR(opt, lam), torch.optim.lr_scheduler.LambdaLR(topt, lam)
    worst, seen = 0.0, []
    for i, b in enumerate(_opt_batches(steps)):
        tr.training_step(model, b)
        for n in names:
            tp[n].grad = model._param(n).grad.detach().float().cpu().clone()
        total = torch.nn.utils.clip_grad_norm_(list(tp.values()), 1.0)
        norm = opt.clip_grad_norm(1.0)
        assert abs(float(norm) - float(total)) <= 1e-3 * float(total)
        assert opt.param_groups[0]["lr"] == topt.param_groups[0]["lr"]
        seen.append(opt.param_groups[0]["lr"])
        topt.step()
        opt.step()
        sch.step()
        tsch.step()
        opt.zero_grad(set_to_none=True)
        flat_ref = torch.zeros(model.grad_arena.numel())           # the optimizer's flat state follows the gradient arena's layout
        for n in names:
            flat_ref[model._grad_offs[n]: model._grad_offs[n] + tp[n].numel()] = tp[n].detach().reshape(-1)
        err = rel_l2(opt.master.cpu().numpy(), flat_ref.numpy())
        assert err < 2e-5, (i, err)
        worst = max(worst, err)
        for n in names:
            got = model._param(n).detach().float().cpu()
            assert torch.equal(got, tp[n].detach().to(torch.bfloat16).float()) or rel_l2(got.numpy(), tp[n].detach().numpy()) < 4e-3, n
    assert seen[0] == 0.0 and seen[1] == 0.5e-3 and seen[2] == 1e-3 and seen[3] < 1e-3      # the schedule really moved the fused lr
    return worst


def check_fused_optimizer_resume(device, tmpdir):
    """Auto-resume (train_mllava.py:281-294): 2 steps, save model + optimizer + scheduler the way HF does (torch.save of state_dict()),
    build everything anew, load, 2 more steps == 4 uninterrupted steps, bit for bit (parameters, fp32 masters, moments, lr)."""
    import os
    from mantis_amd.trainer import MantisHipTrainer
    from mantis_amd.optim import FusedAdamW
    batches = _opt_batches(4)
    lam = lambda s: 1.0 / (1.0 + s)

    def fresh():
        model, _, _ = build_product_model("siglip", device)
        opt = FusedAdamW(model, lr=1e-3, weight_decay=0.0, max_grad_norm=1.0)
        return model, opt, torch.optim.lr_scheduler.LambdaLR(opt, lam), MantisHipTrainer(model, 1)

    def run(model, opt, sch, tr, bs):
        for b in bs:
            tr.training_step(model, b)
            opt.clip_grad_norm(1.0)
            opt.step()
            sch.step()
            opt.zero_grad(set_to_none=True)
    m_a, o_a, s_a, t_a = fresh()
    run(m_a, o_a, s_a, t_a, batches)                                    # 4 steps straight
    m_b, o_b, s_b, t_b = fresh()
    run(m_b, o_b, s_b, t_b, batches[:2])
    torch.save(o_b.state_dict(), os.path.join(tmpdir, "optimizer.pt"))
    torch.save(s_b.state_dict(), os.path.join(tmpdir, "scheduler.pt"))
    torch.save({n: p.detach().cpu() for n, p in m_b.named_parameters()}, os.path.join(tmpdir, "model.pt"))
    m_c, o_c, s_c, t_c = fresh()
    m_c.load_reference_state_dict({n: v.float().numpy() for n, v in torch.load(os.path.join(tmpdir, "model.pt")).items()})
    o_c.load_state_dict(torch.load(os.path.join(tmpdir, "optimizer.pt"), map_location=device, weights_only=True))
    s_c.load_state_dict(torch.load(os.path.join(tmpdir, "scheduler.pt")))
    assert o_c.step_count == 2 and o_c.param_groups[0]["lr"] == o_b.param_groups[0]["lr"]
    run(m_c, o_c, s_c, t_c, batches[2:])
    assert torch.equal(m_c.arena, m_a.arena), "parameters after resume differ from the uninterrupted run"
    for k in ("master_lo", "exp_avg", "exp_avg_sq", "master"):     # the stored halves AND the joined fp32 masters
        assert torch.equal(getattr(o_c, k), getattr(o_a, k)), k
    assert o_c.step_count == 4 and o_c.param_groups[0]["lr"] == o_a.param_groups[0]["lr"]
    return 0.0


def load_cfg1():
    """cfg1 (Mantis-tiny) fixture recorded from the reference (tests/golden/make_golden_cfg1.py): returns (meta, weights as
    bf16-rounded fp32 tensors regenerated from the stored seed, a z-like dict with the inputs, the fixture itself)."""
    from oracle.llava_ref import random_weights
    f = np.load(os.path.join(G, "cfg1_mantis_tiny_step.npz"))
    meta = json.loads(str(f["meta"]))
    w = {k: v.to(torch.bfloat16).float() for k, v in random_weights(meta, seed=int(f["weight_seed"]), perturb_1d=0.05).items()}
    pix = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(int(f["pixel_seed"])))
    z = dict(input_ids=f["input_ids"], attention_mask=f["attention_mask"], labels=f["labels"], pixel_values=pix.numpy(),
             pixel_counts=np.array([1]))
    return meta, w, z, f


class ZDict(dict):
    """dict with the `.files` attribute of an NpzFile (pixels_list / check_step_against_oracle accept either)."""
    @property
    def files(self):
        return list(self.keys())


# ----------------------------------------------------------------------------------------------------------- Idefics2 path
def build_idefics2_product(device):
    from mantis_amd.configuration_idefics2 import Idefics2Config
    from mantis_amd.modeling_idefics2 import Idefics2ForConditionalGeneration
    meta, sd = golden_cfg_and_weights("idefics2")
    model = Idefics2ForConditionalGeneration(Idefics2Config.from_oracle_meta(meta), device=device, init=None)
    model.load_reference_state_dict(sd)
    return model


def build_idefics2_oracle_bf16():
    from oracle.idefics2_ref import Idefics2Ref
    meta, sd = golden_cfg_and_weights("idefics2")
    return Idefics2Ref({k: torch.from_numpy(v).to(torch.bfloat16).float() for k, v in sd.items()}, meta)


def idefics2_batch(z):
    b = dict(input_ids=torch.from_numpy(z["input_ids"]), attention_mask=torch.from_numpy(z["attention_mask"]),
             labels=torch.from_numpy(z["labels"]))
    b["pixel_values"] = torch.from_numpy(z["pixel_values"]) if "pixel_values" in z.files else None
    b["pixel_attention_mask"] = torch.from_numpy(z["pixel_attention_mask"]) if "pixel_attention_mask" in z.files else None
    return b


def check_idefics2_step_against_oracle(model, oracle, z, out, rec, loss_rtol=5e-3, grad_cos=GRAD_COS, grad_rel=GRAD_REL, envelope=None):
    """bf16 product path vs the fp32 Idefics2 oracle on identical (bf16-rounded) weights."""
    orec = {}
    oracle.zero_grad()
    pv = z["pixel_values"] if "pixel_values" in z.files else None
    pm = z["pixel_attention_mask"] if "pixel_attention_mask" in z.files else None
    oloss, ologits = oracle.forward(z["input_ids"], pv, pm, z["attention_mask"], z["labels"], record=orec)
    oloss.backward()
    loss = float(out["loss"].float().cpu().reshape(-1)[0])
    assert abs(loss - float(oloss)) <= loss_rtol * abs(float(oloss)), (loss, float(oloss))
    am = z["attention_mask"].astype(bool)
    for k in orec:
        a, b = rec[k].float().cpu().numpy(), orec[k].detach().numpy()
        if k.startswith("llm_layer") or k == "merged_embeds":
            a, b = a[am], b[am]
        assert rel_l2(a, b) < 3e-2, (k, rel_l2(a, b))
    lg = out["logits"].float().cpu().numpy()
    assert rel_l2(lg[am], ologits.detach().numpy()[am]) < 3e-2
    report = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        og = oracle.w[name].grad
        g = p.grad.float().cpu().numpy()
        if og is None:
            assert not g.any(), name
            continue
        c, r = cosine(g, og.numpy()), rel_l2(g, og.numpy())
        report[name] = (c, r)
        if np.linalg.norm(og.numpy()) < 1e-12:
            assert np.linalg.norm(g) < 1e-6, name
            continue
        cbar, rbar = envelope_bars(envelope, name, grad_cos, grad_rel)
        assert c >= cbar and r <= rbar, (name, c, r, cbar, rbar)
    _note_report(report)
    return report


# ----------------------------------------------------------------------------------------------------------- Qwen2-VL path
def build_qwen2vl_product(device):
    from mantis_amd.configuration_qwen2_vl import Qwen2VLConfig
    from mantis_amd.modeling_qwen2_vl import Qwen2VLForConditionalGeneration
    meta, sd = golden_cfg_and_weights("qwen2vl")
    model = Qwen2VLForConditionalGeneration(Qwen2VLConfig.from_oracle_meta(meta), device=device, init=None)
    model.load_reference_state_dict(sd)
    return model


def build_qwen2vl_oracle_bf16():
    from oracle.qwen2vl_ref import Qwen2VLRef
    meta, sd = golden_cfg_and_weights("qwen2vl")
    return Qwen2VLRef({k: torch.from_numpy(v).to(torch.bfloat16).float() for k, v in sd.items()}, meta)


def qwen2vl_batch(z):
    b = dict(input_ids=torch.from_numpy(z["input_ids"]), attention_mask=torch.from_numpy(z["attention_mask"]),
             labels=torch.from_numpy(z["labels"]))
    b["pixel_values"] = torch.from_numpy(z["pixel_values"]) if "pixel_values" in z.files else None
    b["image_grid_thw"] = torch.from_numpy(z["image_grid_thw"]) if "image_grid_thw" in z.files else None
    return b


def check_qwen2vl_step_against_oracle(model, oracle, z, out, rec, loss_rtol=5e-3, grad_cos=GRAD_COS, grad_rel=GRAD_REL, act_rel=3e-2,
                                      grad_cos_1d=None, grad_rel_1d=None, kbias_cos=0.998, kbias_rel=6e-2, envelope=None):
    """bf16 product path vs the fp32 Qwen2-VL oracle on identical (bf16-rounded) weights.  grad_cos_1d / grad_rel_1d: separate bar for
    the 1-D parameters (biases, norm weights; default: the matrices' bar); kbias_cos / kbias_rel: the KEY bias, whose gradient is a
    near-cancelling sum (a constant added to every key only shifts the scores of a query uniformly, up to RoPE) and therefore mostly rounding
    noise of dS -- bf16 measured 0.99898 / 0.045 at worst (profiles/r05_grad_parity.md); the REFERENCE'S OWN bf16 run (transformers' class, 4
    goldens + 24 random inputs of the golden geometry) deviates from its fp32 run by up to 0.99843 / 0.063 on this tensor
    (tests/golden/bf16_envelope.json, `reference_bf16:qwen2vl_tiny_random24_worst`): the default bars 0.998 / 6e-2 lie inside 1.5 x that.
    With an fp8 1-D bar given, FP8_COS_KBIAS applies.  envelope: per-tensor bars from the reference's bf16 deviation (envelope_bars)."""
    orec = {}
    oracle.zero_grad()
    pv = z["pixel_values"] if "pixel_values" in z.files else None
    grid = z["image_grid_thw"] if "image_grid_thw" in z.files else None
    oloss, ologits = oracle.forward(z["input_ids"], pv, grid, z["attention_mask"], z["labels"], record=orec)
    oloss.backward()
    loss = float(out["loss"].float().cpu().reshape(-1)[0])
    assert abs(loss - float(oloss)) <= loss_rtol * abs(float(oloss)), (loss, float(oloss))
    am = z["attention_mask"].astype(bool)
    assert np.array_equal(rec["position_ids"].cpu().numpy(), orec["position_ids"].numpy())          # integer work: bit-exact
    for k in orec:
        if k == "position_ids":
            continue
        a, b = rec[k].float().cpu().numpy(), orec[k].detach().numpy()
        if k.startswith("llm_layer") or k == "merged_embeds":
            a, b = a[am], b[am]
        assert rel_l2(a, b) < act_rel, (k, rel_l2(a, b))
    lg = out["logits"].float().cpu().numpy()
    assert rel_l2(lg[am], ologits.detach().numpy()[am]) < act_rel
    report = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        og = oracle.w[name].grad
        g = p.grad.float().cpu().numpy()
        if og is None:
            assert not g.any(), name
            continue
        c, r = cosine(g, og.numpy()), rel_l2(g, og.numpy())
        report[name] = (c, r)
        if np.linalg.norm(og.numpy()) < 1e-12:
            assert np.linalg.norm(g) < 1e-6, name
            continue
        one_d = p.dim() == 1
        cbar, rbar = (grad_cos if grad_cos_1d is None else grad_cos_1d, grad_rel if grad_rel_1d is None else grad_rel_1d) if one_d else (grad_cos, grad_rel)
        if one_d and name.endswith("k_proj.bias"):                 # the near-cancelling key-bias gradient
            cbar = min(kbias_cos, FP8_COS_KBIAS) if grad_cos_1d is not None else kbias_cos
            rbar = max(rbar, kbias_rel)
        if envelope is not None:                                   # bars from the reference's own bf16 deviation (see envelope_bars)
            cbar, rbar = envelope_bars(envelope, name, cbar if not (one_d and name.endswith("k_proj.bias")) else GRAD_COS,
                                       rbar if not (one_d and name.endswith("k_proj.bias")) else GRAD_REL)
        assert c >= cbar and r <= rbar, (name, c, r, cbar, rbar)
    _note_report(report)
    return report


# ----------------------------------------------------------------------------------------------------------- fp8 variant, any path
FP8_COS_2D, FP8_COS_1D, FP8_COS_KBIAS, FP8_GRAD_REL_2D, FP8_ACT_REL = 0.97, 0.93, 0.85, 0.25, 0.12
# The stated tolerance of the fp8-linear variant, set FROM MEASUREMENT (round 3; CPU emulation of the exact fp8 arithmetic on the four
# Qwen2-VL goldens: weight-matrix gradient cosine 0.9814-0.9846 min / rel-L2 <= 0.193, 1-D parameters 0.959-0.987 min; GPU kernels: the
# same to within fp8 rounding flips, worst 1-D value 0.908): matrices >= 0.97 and rel-L2 <= 0.25, biases / norm weights >= 0.93 -- except
# the KEY bias, >= 0.85: its gradient is a near-cancelling sum (a constant added to every key shifts all scores of a query alike, up to
# RoPE), i.e. mostly quantisation noise of e5m2 dS, and it is the one that measured 0.908 / 0.959.  Activations rel-L2 <= 0.12, loss 1e-2.


def check_fp8_grads_against_oracle(model, oracle, loss, oloss, loss_rtol=1e-2, cos_2d=FP8_COS_2D, cos_1d=FP8_COS_1D):
    """The stated tolerance of the fp8-linear variant (per-tensor e4m3 activations / weights, e5m2 output gradients) against the fp32
    oracle of the reference: loss 1e-2 relative, weight-matrix gradient cosine >= 0.97, bias / norm-weight gradient cosine >= 0.93
    (key bias 0.85, see FP8_COS_KBIAS)."""
    assert abs(float(loss) - float(oloss)) <= loss_rtol * abs(float(oloss)), (float(loss), float(oloss))
    worst = 1.0
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        og = oracle.w[name].grad
        g = p.grad.float().cpu().numpy()
        if og is None or np.linalg.norm(og.numpy()) < 1e-12:
            continue
        c = cosine(g, og.numpy())
        bar = cos_2d if p.dim() > 1 else (min(cos_1d, FP8_COS_KBIAS) if name.endswith("k_proj.bias") else cos_1d)
        assert c >= bar, (name, c)
        worst = min(worst, c)
    return worst
