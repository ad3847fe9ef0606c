#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING the reference (TIGER-AI-Lab/Mantis @ /root/reference).

Runs only in the build container (needs /root/reference + transformers). Nothing in the
test-suite, smoke() or bench.py imports this file or reads /root/reference: they only read
the ``*.npz`` fixtures this script writes next to itself.  A fixture is data: inputs,
weights (random-init, seeded) and the reference's outputs.  No reference source text is
stored.

The hot path being recorded:
  * ``LlavaForConditionalGeneration.forward``        mantis/models/mllava/modeling_llava.py:364-549
  * ``_merge_input_ids_with_image_features``         mantis/models/mllava/modeling_llava.py:293-360
  * ``LlavaMultiModalProjector``                     mantis/models/mllava/modeling_llava.py:106-118
  * ``transformers.Trainer.training_step``           (third-party; called from mantis/train/train_mllava.py:312-329)
  * label-mask rule / collation / sample packing     recorded by make_golden_harness.py (runs mantis/train/data.py itself)

Oracle-side shim: transformers 5.x removed two helper names the
reference's processing module imports, and calls ``tie_weights`` with a kwarg; both are
absorbed here without touching the reference files.

Usage:  python tests/golden/make_golden.py          (writes tests/golden/*.npz)
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    import transformers.utils.hub as hub
    for n in ("is_remote_url", "download_url"):
        if not hasattr(hub, n):
            setattr(hub, n, lambda *a, **k: False)
    sys.path.insert(0, REF)
    from mantis.models.mllava.modeling_llava import LlavaForConditionalGeneration
    from mantis.models.mllava.configuration_llava import LlavaConfig

    class Oracle(LlavaForConditionalGeneration):
        def tie_weights(self, *a, **k):
            return self.language_model.tie_weights(*a, **k)

    return Oracle, LlavaConfig


V, IMG, PAD = 300, 298, 299
D, DV = 64, 64


def vision_cfg(flavour):
    if flavour == "siglip":
        return dict(model_type="siglip_vision_model", hidden_size=DV, intermediate_size=112, num_hidden_layers=3,
                    num_attention_heads=4, image_size=56, patch_size=14, num_channels=3,
                    hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)
    return dict(model_type="clip_vision_model", hidden_size=DV, intermediate_size=112, num_hidden_layers=3,
                num_attention_heads=4, image_size=56, patch_size=14, num_channels=3, hidden_act="quick_gelu",
                layer_norm_eps=1e-5, projection_dim=32)


def text_cfg():
    return dict(model_type="llama", hidden_size=D, intermediate_size=176, num_hidden_layers=2,
                num_attention_heads=4, num_key_value_heads=2, vocab_size=V, rope_theta=500000.0,
                rms_norm_eps=1e-5, max_position_embeddings=512, tie_word_embeddings=False,
                attention_bias=False, mlp_bias=False, hidden_act="silu")


def build(flavour, seed):
    Oracle, LlavaConfig = _import_reference()
    torch.manual_seed(seed)
    cfg = LlavaConfig(vision_config=vision_cfg(flavour), text_config=text_cfg(), image_token_index=IMG,
                      pad_token_id=PAD, vocab_size=V,
                      vision_feature_select_strategy="full" if flavour == "siglip" else "default")
    cfg._attn_implementation = "eager"
    model = Oracle(cfg)
    # HF init leaves several tensors at trivial values (LN weight 1, bias 0); perturb EVERYTHING so a
    # forgotten bias / norm weight in the build shows up in parity.
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                if "norm" in n and n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(p + 0.02 * torch.randn(p.shape, generator=g))
        # make the token embedding larger so no row is accidentally ~0 (packing step 5 looks for all-zero rows)
        model.get_input_embeddings().weight.mul_(5.0)
    for p in model.vision_tower.parameters():      # mantis/train/train_mllava.py:240-242
        p.requires_grad_(False)
    model.train()
    return model, cfg


def make_ids(rng, T, img_positions, n_pad=0):
    ids = rng.integers(0, IMG - 1, size=T, dtype=np.int64)
    for p in img_positions:
        ids[p] = IMG
    mask = np.ones(T, dtype=np.int64)
    if n_pad:
        ids[T - n_pad:] = PAD
        mask[T - n_pad:] = 0
    return ids, mask


def make_labels(ids, mask, lead_ignore):
    lab = ids.copy()
    lab[:lead_ignore] = -100
    lab[ids == IMG] = -100
    lab[mask == 0] = -100
    return lab


def run_case(model, cfg, name, ids, mask, labels, pixels, record_acts=True):
    """ids/mask/labels: np [B,T]; pixels: list of np [n_i,3,56,56] or None."""
    out = {}
    out["input_ids"], out["attention_mask"], out["labels"] = ids, mask, labels
    t_ids, t_mask, t_lab = (torch.from_numpy(x) for x in (ids, mask, labels))
    pv = None
    if pixels is not None:
        pv = [torch.from_numpy(p) for p in pixels]
        out["pixel_counts"] = np.array([p.shape[0] for p in pixels], dtype=np.int64)
        out["pixel_values"] = np.concatenate(pixels, 0)
    acts = {}
    hooks = []
    lm = model.language_model
    if record_acts:
        for i, layer in enumerate(lm.model.layers):
            def layer_hook(m, a, o, i=i):
                acts[f"llm_layer{i}_out"] = (o[0] if isinstance(o, tuple) else o).detach()
            hooks.append(layer.register_forward_hook(layer_hook))

        def norm_hook(m, a, o):
            acts["llm_final_norm"] = o.detach()
        hooks.append(lm.model.norm.register_forward_hook(norm_hook))
        def proj_hook(m, a, o):
            acts["projector_in"] = a[0].detach()
            acts["projector_out"] = o.detach()
        hooks.append(model.multi_modal_projector.register_forward_hook(proj_hook))
        # packing outputs: capture by wrapping the bound method
        orig_merge = model._merge_input_ids_with_image_features

        def merge(*a, **k):
            r = orig_merge(*a, **k)
            acts["merged_embeds"], acts["merged_attention_mask"], acts["merged_labels"], acts["merged_position_ids"] = (
                x.detach().clone() for x in r)
            return r
        model._merge_input_ids_with_image_features = merge
    model.zero_grad(set_to_none=True)
    res = model(input_ids=t_ids, pixel_values=pv, attention_mask=t_mask, labels=t_lab)
    res.loss.backward()
    if record_acts:
        for h in hooks:
            h.remove()
        del model._merge_input_ids_with_image_features
    out["loss"] = res.loss.detach().numpy()
    out["logits"] = res.logits.detach().numpy()
    for k, v in acts.items():
        out[k] = v.numpy()
    for n, p in model.named_parameters():
        if p.grad is not None:
            out["grad." + n] = p.grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss={float(res.loss):.6f} logits{tuple(res.logits.shape)}")
    return out


def run_training_step(model, cfg, name, batches, ga):
    """transformers.Trainer.training_step over `ga` micro-batches; records returned losses + accumulated grads."""
    from transformers import Trainer, TrainingArguments
    import tempfile
    args = TrainingArguments(output_dir=tempfile.mkdtemp(), use_cpu=True, report_to=[], remove_unused_columns=False,
                             gradient_accumulation_steps=ga, per_device_train_batch_size=1)
    trainer = Trainer(model=model, args=args)
    trainer.current_gradient_accumulation_steps = ga
    model.zero_grad(set_to_none=True)
    losses = []
    out = {"ga": np.array(ga)}
    for i, (ids, mask, labels, pixels) in enumerate(batches):
        batch = dict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                     labels=torch.from_numpy(labels), pixel_values=[torch.from_numpy(p) for p in pixels])
        out[f"mb{i}.input_ids"], out[f"mb{i}.attention_mask"], out[f"mb{i}.labels"] = ids, mask, labels
        out[f"mb{i}.pixel_values"] = np.concatenate(pixels, 0)
        out[f"mb{i}.pixel_counts"] = np.array([p.shape[0] for p in pixels], dtype=np.int64)
        loss = trainer.training_step(model, batch)
        assert loss.requires_grad is False and loss.dim() == 0
        losses.append(float(loss))
    out["returned_losses"] = np.array(losses, dtype=np.float64)
    for n, p in model.named_parameters():
        if p.grad is not None:
            out["grad." + n] = p.grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: returned losses {losses}")


def main():
    rng = np.random.default_rng(1234)

    def px(n):
        return rng.standard_normal((n, 3, 56, 56)).astype(np.float32)

    for flavour, seed in (("siglip", 11), ("clip", 23)):
        model, cfg = build(flavour, seed)
        sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
        meta = dict(vision=vision_cfg(flavour), text=text_cfg(), image_token_index=IMG, pad_token_id=PAD, vocab_size=V,
                    vision_feature_select_strategy=cfg.vision_feature_select_strategy, vision_feature_layer=-2,
                    projector_hidden_act="gelu", ignore_index=-100)
        import json
        np.savez_compressed(os.path.join(HERE, f"weights_{flavour}.npz"), __config__=np.array(json.dumps(meta)), **sd)

        T = 24
        specs = {
            "b1_img1": [([5], 0)],
            "b1_img2_adjacent": [([7, 8], 0)],
            "b1_img4": [([1, 6, 11, 17], 0)],
            "b1_img_first_last": [([0, T - 1], 0)],
            "b2_equal_rightpad": [([3, 10], 0), ([2, 9], 3)],
            "b2_equal_nopad": [([4, 12], 0), ([0, 20], 0)],
        }
        if flavour == "clip":
            specs = {k: specs[k] for k in ("b1_img2_adjacent", "b2_equal_rightpad")}
        for cname, rows in specs.items():
            ids, mask, labs, pixels = [], [], [], []
            for pos, npad in rows:
                i, m = make_ids(rng, T, pos, npad)
                ids.append(i); mask.append(m); labs.append(make_labels(i, m, lead_ignore=6)); pixels.append(px(len(pos)))
            run_case(model, cfg, f"{flavour}_{cname}", np.stack(ids), np.stack(mask), np.stack(labs), pixels)

        if flavour == "siglip":
            # text-only (pixel_values=None): forward skips the merge, loss shift uses the raw attention mask
            i, m = make_ids(rng, T, [], 4)
            run_case(model, cfg, "siglip_b1_text_only", i[None], m[None], make_labels(i, m, 3)[None], None,
                     record_acts=False)
            # documented quirk: unequal image counts + right padding
            ia, ma = make_ids(rng, T, [3, 10], 0)
            ib, mb = make_ids(rng, T, [2], 5)
            run_case(model, cfg, "siglip_b2_unequal_quirk", np.stack([ia, ib]), np.stack([ma, mb]),
                     np.stack([make_labels(ia, ma, 6), make_labels(ib, mb, 6)]), [px(2), px(1)])
            # count mismatch -> ValueError (modeling_llava.py:347-351)
            try:
                i, m = make_ids(rng, T, [3, 10], 0)
                model(input_ids=torch.from_numpy(i[None]), pixel_values=[torch.from_numpy(px(1))],
                      attention_mask=torch.from_numpy(m[None]), labels=torch.from_numpy(make_labels(i, m, 2)[None]))
                raised = False
            except ValueError:
                raised = True
            assert raised
            # Trainer.training_step with GA in {1, 4}
            for ga in (1, 4):
                batches = []
                for j in range(ga):
                    i, m = make_ids(rng, T, [2 + j, 12], 0)
                    batches.append((i[None], m[None], make_labels(i, m, 5)[None], [px(2)]))
                run_training_step(model, cfg, f"siglip_training_step_ga{ga}", batches, ga)
    # label_rule.npz / collate_ref.npz / pack_batch_ref.npz: tests/golden/make_golden_harness.py (runs the reference's data.py)


if __name__ == "__main__":
    main()
