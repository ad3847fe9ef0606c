#!/usr/bin/env python3
"""Golden fixtures for the Qwen2-VL path (BASELINE.json configs[4]).

The reference's generation model for this path is HuggingFace's own class: /root/reference/mantis/models/qwen2_vl/
modeling_qwen2_vl.py:1 is `from transformers.models.qwen2_vl.modeling_qwen2_vl import *` (its star import no longer resolves `torch`
under transformers 5.x, SURVEY 8c, so the class is taken from transformers directly, which is what the reference's
`Qwen2VLForConditionalGeneration` name resolves to), with the vision tower frozen as /root/reference/mantis/train/
train_qwen2_vl.py:209-212 does.  Recorded per case: inputs, 3-D rope index, vision tower output before / after the patch merger,
merged embeddings, every text layer's output, fp32 logits, loss, gradients of every trainable parameter.

Runs only in the build container (transformers 5.15 present); tests read the .npz files.
Usage: python tests/golden/make_golden_qwen2vl.py
"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
V, IMG, VID, VSTART, VEND, PAD = 320, 311, 312, 309, 310, 0


def meta():
    return dict(
        vision=dict(depth=2, embed_dim=64, hidden_size=112, hidden_act="quick_gelu", mlp_ratio=4, num_heads=4, in_channels=3,
                    patch_size=14, spatial_merge_size=2, temporal_patch_size=2),
        text=dict(hidden_size=112, intermediate_size=256, num_hidden_layers=2, num_attention_heads=7, num_key_value_heads=1,
                  vocab_size=V, rms_norm_eps=1e-6, max_position_embeddings=512, hidden_act="silu",
                  rope_parameters=dict(rope_type="default", rope_theta=1000000.0, mrope_section=[2, 3, 3]),
                  pad_token_id=PAD, tie_word_embeddings=False),
        image_token_id=IMG, video_token_id=VID, vision_start_token_id=VSTART, vision_end_token_id=VEND, vocab_size=V)


def build(seed):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    m = meta()
    cfg = Qwen2VLConfig(vision_config=m["vision"], text_config=m["text"], image_token_id=IMG, video_token_id=VID,
                        vision_start_token_id=VSTART, vision_end_token_id=VEND, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    torch.manual_seed(seed)
    model = Qwen2VLForConditionalGeneration(cfg)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                if ("norm" in n or "ln_q" in n) and n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(p + 0.05 * torch.randn(p.shape, generator=g))
    for n, p in model.named_parameters():
        if "visual" in n:                       # train_qwen2_vl.py:209-212
            p.requires_grad_(False)
    model.train()
    return model


def run_case(model, name, ids, mask, labels, pixels, grid):
    out = dict(input_ids=ids, attention_mask=mask, labels=labels)
    acts, hooks = {}, []
    mm = model.model
    mmtt = (ids == IMG).astype(np.int32)
    out["mm_token_type_ids"] = mmtt
    kw = {}
    if pixels is not None:
        out["pixel_values"] = pixels
        out["image_grid_thw"] = grid
        kw = dict(pixel_values=torch.from_numpy(pixels), image_grid_thw=torch.from_numpy(grid))
        hooks.append(mm.visual.register_forward_hook(lambda m, a, o: (acts.__setitem__("vision_last_hidden_state", o.last_hidden_state.detach()),
                                                                        acts.__setitem__("vision_merged", o.pooler_output.detach()))[0]))
        hooks.append(mm.visual.blocks[0].register_forward_hook(lambda m, a, o: acts.__setitem__("vision_block0_out", o.detach())))
        hooks.append(mm.visual.patch_embed.register_forward_hook(lambda m, a, o: acts.__setitem__("vision_patch_embed", o.detach())))
    lmod = mm.language_model

    def pre(m, a, k):
        acts["merged_embeds"] = k["inputs_embeds"].detach().clone()
        if k.get("position_ids") is not None:          # None for text-only input: the text model then counts 0..T-1 itself
            acts["position_ids"] = k["position_ids"].detach().clone()
    hooks.append(lmod.register_forward_pre_hook(pre, with_kwargs=True))
    for i, layer in enumerate(lmod.layers):
        hooks.append(layer.register_forward_hook(lambda m, a, o, i=i: acts.__setitem__(f"llm_layer{i}_out", (o[0] if isinstance(o, tuple) else o).detach())))
    model.zero_grad(set_to_none=True)
    mm.rope_deltas = None
    res = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), labels=torch.from_numpy(labels),
                mm_token_type_ids=torch.from_numpy(mmtt), use_cache=False, **kw)
    res.loss.backward()
    for h in hooks:
        h.remove()
    out["loss"] = res.loss.detach().numpy()
    out["logits"] = res.logits.detach().numpy()
    for k, v in acts.items():
        out[k] = v.numpy()
    for n, p in model.named_parameters():
        if p.grad is not None:
            out["grad." + n] = p.grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss={float(res.loss):.6f} logits{tuple(res.logits.shape)} grads={sum(1 for k in out if k.startswith('grad.'))} "
          f"acts={sorted(acts)}")


def main():
    model = build(53)
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "weights_qwen2vl.npz"), __config__=np.array(json.dumps(meta())), **sd)
    rng = np.random.default_rng(99)

    def text(T, images, n_pad=0):
        """images: list of (start, n_tokens): <|vision_start|> <|image_pad|> * n <|vision_end|> placed at start."""
        ids = rng.integers(1, 300, size=T, dtype=np.int64)
        for s, n in images:
            ids[s] = VSTART
            ids[s + 1: s + 1 + n] = IMG
            ids[s + 1 + n] = VEND
        mask = np.ones(T, np.int64)
        if n_pad:
            ids[T - n_pad:] = PAD
            mask[T - n_pad:] = 0
        lab = ids.copy()
        lab[:6] = -100
        lab[ids == IMG] = -100
        lab[mask == 0] = -100
        return ids, mask, lab

    def px(grids):
        n = int(sum(t * h * w for t, h, w in grids))
        return rng.standard_normal((n, 3 * 2 * 14 * 14)).astype(np.float32)

    # B=1, two images of different aspect (4x6 and 6x4 patches -> 6 merged tokens each)
    g = np.array([[1, 4, 6], [1, 6, 4]], np.int64)
    i, m, l = text(34, [(2, 6), (17, 6)])
    run_case(model, "qwen2vl_b1_img2", i[None], m[None], l[None], px(g), g)
    # B=1, one larger image (8x4 patches -> 8 tokens), dynamic resolution
    g = np.array([[1, 8, 4]], np.int64)
    i, m, l = text(24, [(3, 8)])
    run_case(model, "qwen2vl_b1_img1_tall", i[None], m[None], l[None], px(g), g)
    # B=2: sample 0 two images, sample 1 one image and right-padded text
    g = np.array([[1, 4, 4], [1, 2, 6], [1, 4, 6]], np.int64)
    a = text(30, [(1, 4), (12, 3)])
    b = text(30, [(5, 6)], n_pad=5)
    run_case(model, "qwen2vl_b2_rightpad", np.stack([a[0], b[0]]), np.stack([a[1], b[1]]), np.stack([a[2], b[2]]), px(g), g)
    # text only
    i, m, l = text(20, [], n_pad=3)
    run_case(model, "qwen2vl_b1_text_only", i[None], m[None], l[None], None, None)


if __name__ == "__main__":
    main()
