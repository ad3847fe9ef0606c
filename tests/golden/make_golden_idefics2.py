#!/usr/bin/env python3
"""Golden fixtures for the Idefics2 path (BASELINE.json configs[3]) by RUNNING the reference fork
/root/reference/mantis/models/idefics2/modeling_idefics2.py (Idefics2ForConditionalGeneration.forward :1797-1912 over
Idefics2Model.forward :1580-1722) in the build container.  Recorded per case: inputs, the vision tower's last_hidden_state, the
connector (perceiver) output, merged embeddings, every text layer's output, fp32 logits, loss, and the gradient of every
trainable parameter (the vision tower is frozen, as under the reference's LoRA target list, train_idefics2.py:156).

Oracle-side shim: `tie_weights` override for transformers 5.x, `use_cache=False` (the fork's DynamicCache call is gone
in HF 5).  Runs only here; tests read the .npz files.   Usage: python tests/golden/make_golden_idefics2.py
"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
V, IMG, PAD = 300, 298, 0
NL = 4                     # latents per image = <image> tokens per image


def meta():
    return dict(
        vision=dict(hidden_size=64, intermediate_size=112, num_hidden_layers=3, num_attention_heads=4, image_size=56, patch_size=14,
                    num_channels=3, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6),
        perceiver=dict(hidden_act="silu", resampler_n_latents=NL, resampler_depth=2, resampler_n_heads=4, resampler_head_dim=16,
                       num_key_value_heads=2, attention_dropout=0.0),
        text=dict(model_type="mistral", hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4,
                  num_key_value_heads=2, vocab_size=V, rope_theta=10000.0, rms_norm_eps=1e-5, max_position_embeddings=512,
                  sliding_window=None, hidden_act="silu", pad_token_id=PAD),
        image_token_id=IMG, vocab_size=V)


def build(seed):
    import transformers.utils.hub as hub
    for n in ("is_remote_url", "download_url"):
        if not hasattr(hub, n):
            setattr(hub, n, lambda *a, **k: False)
    sys.path.insert(0, REF)
    from mantis.models.idefics2.modeling_idefics2 import Idefics2ForConditionalGeneration
    from transformers import Idefics2Config

    class Oracle(Idefics2ForConditionalGeneration):
        def tie_weights(self, *a, **k):
            return None

    m = meta()
    cfg = Idefics2Config(vision_config=m["vision"], perceiver_config=m["perceiver"], text_config=m["text"], image_token_id=IMG,
                         tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    torch.manual_seed(seed)
    model = Oracle(cfg)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                if "norm" in n and n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif n.endswith("perceiver_resampler.latents"):
                p.copy_(1.0 + 0.5 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(p + 0.05 * torch.randn(p.shape, generator=g))
    for p in model.model.vision_model.parameters():
        p.requires_grad_(False)
    model.train()
    return model


def run_case(model, name, ids, mask, labels, pixels, pixel_mask):
    out = dict(input_ids=ids, attention_mask=mask, labels=labels)
    acts, hooks = {}, []
    mm = model.model
    if pixels is not None:
        out["pixel_values"] = pixels
        if pixel_mask is not None:
            out["pixel_attention_mask"] = pixel_mask
        hooks.append(mm.vision_model.register_forward_hook(lambda m, a, o: acts.__setitem__("vision_last_hidden_state", o.last_hidden_state.detach())))
        hooks.append(mm.connector.register_forward_hook(lambda m, a, o: acts.__setitem__("connector_out", o.detach())))
        hooks.append(mm.connector.modality_projection.register_forward_hook(lambda m, a, o: acts.__setitem__("modality_projection_out", o.detach())))
        orig = mm.inputs_merger

        def merger(*a, **k):
            r = orig(*a, **k)
            acts["merged_embeds"] = r.detach().clone()
            return r
        mm.inputs_merger = merger
    for i, layer in enumerate(mm.text_model.layers):
        hooks.append(layer.register_forward_hook(lambda m, a, o, i=i: acts.__setitem__(f"llm_layer{i}_out", (o[0] if isinstance(o, tuple) else o).detach())))
    model.zero_grad(set_to_none=True)
    res = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), labels=torch.from_numpy(labels),
                pixel_values=None if pixels is None else torch.from_numpy(pixels),
                pixel_attention_mask=None if pixel_mask is None else torch.from_numpy(pixel_mask), use_cache=False)
    res.loss.backward()
    for h in hooks:
        h.remove()
    if pixels is not None:
        del mm.inputs_merger
    out["loss"] = res.loss.detach().numpy()
    out["logits"] = res.logits.detach().numpy()
    for k, v in acts.items():
        out[k] = v.numpy()
    for n, p in model.named_parameters():
        if p.grad is not None:
            out["grad." + n] = p.grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss={float(res.loss):.6f} logits{tuple(res.logits.shape)} grads={sum(1 for k in out if k.startswith('grad.'))}")


def main():
    model = build(31)
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "weights_idefics2.npz"), __config__=np.array(json.dumps(meta())), **sd)
    rng = np.random.default_rng(77)

    def text(T, img_starts, n_pad=0):
        ids = rng.integers(1, 290, size=T, dtype=np.int64)
        for s in img_starts:
            ids[s: s + NL] = IMG
        mask = np.ones(T, np.int64)
        if n_pad:
            ids[T - n_pad:] = PAD
            mask[T - n_pad:] = 0
        lab = ids.copy()
        lab[:5] = IMG                       # the reference's ignore index on this path is image_token_id (train_idefics2.py:164)
        lab[mask == 0] = IMG
        return ids, mask, lab

    def px(n):
        return rng.standard_normal((n, 3, 56, 56)).astype(np.float32)

    # B=1, two full-resolution images
    i, m, l = text(26, [3, 14])
    run_case(model, "idefics2_b1_img2", i[None], m[None], l[None], px(2)[None], None)
    # B=1, variable resolution: image 0 is 42 x 28 px inside the 56 x 56 canvas (3 x 2 patches), image 1 full
    pm = np.ones((1, 2, 56, 56), bool)
    pm[0, 0] = False
    pm[0, 0, :42, :28] = True
    pv = px(2)[None]
    pv[0, 0][:, ~pm[0, 0]] = 0.0
    i, m, l = text(26, [2, 12])
    run_case(model, "idefics2_b1_navit", i[None], m[None], l[None], pv, pm)
    # B=2: sample 0 has two images, sample 1 one image (+ an all-zero padding image) and right-padded text
    pv = np.zeros((2, 2, 3, 56, 56), np.float32)
    pv[0] = px(2)
    pv[1, 0] = px(1)[0]
    pm = np.ones((2, 2, 56, 56), bool)
    pm[1, 0] = False
    pm[1, 0, :28, :56] = True
    pv[1, 0][:, ~pm[1, 0]] = 0.0
    a = text(28, [1, 16])
    b = text(28, [6], n_pad=4)
    run_case(model, "idefics2_b2_padimg_rightpad", np.stack([a[0], b[0]]), np.stack([a[1], b[1]]), np.stack([a[2], b[2]]), pv, pm)
    # text only
    i, m, l = text(20, [], n_pad=3)
    run_case(model, "idefics2_b1_text_only", i[None], m[None], l[None], None, None)


if __name__ == "__main__":
    main()
