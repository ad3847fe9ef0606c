#!/usr/bin/env python3
"""cfg1 fixture (BASELINE.json configs[0]: Mantis-tiny = SigLIP-base/16-224 + Llama-68M, 1 image 224^2, 128 tokens, bs 1, CPU):
the REFERENCE's `Trainer.training_step` on seeded weights and a seeded batch -> loss, logits statistics and per-parameter
gradient norms ("the cfg1 full-size run's loss + grad-norms (seeded) as a smoke value").

The 180 M weights are not stored: they are `oracle.llava_ref.random_weights(meta, seed, perturb_1d=0.05)` rounded to bf16
(a seeded torch-CPU procedure that reproduces bit-identically on the GPU box, same image / torch build).  Stored: the
integer inputs, the pixel seed, and the reference's outputs.  Runs only in the build container.

Usage: python tests/golden/make_golden_cfg1.py
"""
import json
import os
import sys
import tempfile

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

WEIGHT_SEED, PIXEL_SEED, IDS_SEED = 7, 5, 9
T, V, IMG, PAD = 128, 32002, 32000, 32001


def meta_cfg1():
    return dict(
        vision=dict(model_type="siglip_vision_model", hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                    num_attention_heads=12, image_size=224, patch_size=16, num_channels=3, hidden_act="gelu_pytorch_tanh",
                    layer_norm_eps=1e-6),
        text=dict(model_type="llama", hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=12,
                  num_key_value_heads=12, vocab_size=V, rope_theta=10000.0, rms_norm_eps=1e-6, max_position_embeddings=2048,
                  tie_word_embeddings=False, attention_bias=False, mlp_bias=False, hidden_act="silu"),
        image_token_index=IMG, pad_token_id=PAD, vocab_size=V, vision_feature_select_strategy="full", vision_feature_layer=-2,
        projector_hidden_act="gelu", ignore_index=-100)


def inputs():
    g = torch.Generator().manual_seed(IDS_SEED)
    ids = torch.randint(0, 32000, (1, T), generator=g)
    ids[0, 17] = IMG
    labels = ids.clone()
    labels[:, : T // 2] = -100
    labels[ids == IMG] = -100
    pix = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(PIXEL_SEED))
    return ids, torch.ones_like(ids), labels, pix


def main():
    from make_golden import _import_reference
    from oracle.llava_ref import random_weights
    from transformers import Trainer, TrainingArguments
    Oracle, LlavaConfig = _import_reference()
    meta = meta_cfg1()
    w = {k: v.to(torch.bfloat16).float() for k, v in random_weights(meta, seed=WEIGHT_SEED, perturb_1d=0.05).items()}
    cfg = LlavaConfig(vision_config=meta["vision"], text_config=meta["text"], image_token_index=IMG, pad_token_id=PAD,
                      vocab_size=V, vision_feature_select_strategy="full")
    cfg._attn_implementation = "eager"
    model = Oracle(cfg)
    missing, unexpected = model.load_state_dict(w, strict=False)
    assert not unexpected, unexpected
    assert all(".head." in k or "post_layernorm" in k for k in missing), missing      # SigLIP pooling head / post-LN: dead on this path
    for p in model.vision_tower.parameters():          # train_mllava.py:240-242
        p.requires_grad_(False)
    model.train()
    ids, am, labels, pix = inputs()
    args = TrainingArguments(output_dir=tempfile.mkdtemp(), use_cpu=True, report_to=[], remove_unused_columns=False,
                             gradient_accumulation_steps=1, per_device_train_batch_size=1)
    trainer = Trainer(model=model, args=args)
    trainer.current_gradient_accumulation_steps = 1
    model.zero_grad(set_to_none=True)
    ret = trainer.training_step(model, dict(input_ids=ids, attention_mask=am, labels=labels, pixel_values=[pix]))
    with torch.no_grad():
        res = model(input_ids=ids, pixel_values=[pix], attention_mask=am, labels=labels)
    names, norms = [], []
    for n, p in model.named_parameters():
        if p.grad is not None:
            names.append(n)
            norms.append(float(p.grad.double().norm()))
    lg = res.logits.double()
    out = dict(meta=np.array(json.dumps(meta)), weight_seed=np.array(WEIGHT_SEED), pixel_seed=np.array(PIXEL_SEED),
               input_ids=ids.numpy(), attention_mask=am.numpy(), labels=labels.numpy(),
               returned_loss=np.array(float(ret)), loss=np.array(float(res.loss)), logits_shape=np.array(lg.shape),
               logits_mean=np.array(float(lg.mean())), logits_abs_mean=np.array(float(lg.abs().mean())),
               logits_row0=lg[0, -1, :64].numpy(), grad_names=np.array(json.dumps(names)), grad_norms=np.array(norms),
               total_grad_norm=np.array(float(np.sqrt(np.sum(np.square(norms))))))
    np.savez_compressed(os.path.join(HERE, "cfg1_mantis_tiny_step.npz"), **out)
    print(f"cfg1: returned loss {float(ret):.6f}, {len(names)} gradients, total grad norm {float(out['total_grad_norm']):.6f}, "
          f"logits {tuple(lg.shape)}")


if __name__ == "__main__":
    main()
