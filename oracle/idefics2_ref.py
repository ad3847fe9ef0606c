"""ORACLE (test infrastructure, never imported by the product path): the Idefics2 training forward of the Mantis fork.

Pure-torch fp32, transformers-free CPU restatement of SURVEY.md section 8 row f1 (BASELINE.json configs[3]):
  NaViT patch embedding + bucketised position ids     /root/reference/mantis/models/idefics2/modeling_idefics2.py:155-210
  SigLIP encoder with a patch (key) attention mask    :214-291 (eager attention), :555-603 (layer), :710-767 (post_layernorm!)
  padding-image removal, pixel mask -> patch mask     :1636-1658
  connector: modality projection (SwiGLU MLP)         :506-522, :1320-1334
  perceiver resampler                                 :812-912 (attention over concat[context, latents], GQA), :1187-1257, :1259-1317
  inputs_merger                                       :1545-1565
  text model = HF MistralModel (third-party: RMSNorm, GQA + RoPE, SwiGLU; default position ids = arange)  invoked at :1700-1708
  lm_head, logits.float(), shifted + attention-mask-filtered CE with ignore_index = image_token_id        :1884-1899

PARITY PIN: tests/golden/make_golden_idefics2.py imports the reference fork in the build container and records inputs, weights,
activations, loss and gradients; tests/test_idefics2_oracle.py checks this restatement against them.
"""
import json

import numpy as np
import torch
import torch.nn.functional as F

from .llava_ref import ACT, apply_rope, attention, layernorm, rmsnorm, rope_cos_sin


def bucketized_position_ids(patch_mask, num_patches_per_side):
    """modeling_idefics2.py:190-210.  patch_mask bool [I, ph, pw] -> int64 [I, ph*pw] (0 where the patch is padding)."""
    I, ph, pw = patch_mask.shape
    boundaries = torch.arange(1 / num_patches_per_side, 1.0, 1 / num_patches_per_side)
    out = torch.zeros((I, ph * pw), dtype=torch.int64)
    for i in range(I):
        m = patch_mask[i]
        nh, nw = m[:, 0].sum(), m[0].sum()
        fh = torch.arange(0, 1 - 1e-6, 1 / nh)
        fw = torch.arange(0, 1 - 1e-6, 1 / nw)
        bh = torch.bucketize(fh, boundaries, right=True)
        bw = torch.bucketize(fw, boundaries, right=True)
        out[i][m.reshape(-1)] = (bh[:, None] * num_patches_per_side + bw).flatten()
    return out


def patch_mask_from_pixel_mask(pixel_mask, patch):
    """:1653-1658: a patch is valid if any of its pixels is.  pixel_mask bool [I, H, W] -> bool [I, H/p, W/p]."""
    sub = pixel_mask.unfold(1, patch, patch).unfold(2, patch, patch)
    return sub.sum(dim=(-1, -2)) > 0


class Idefics2Ref:
    def __init__(self, weights, cfg, dtype=torch.float32, train_vision=False):
        self.cfg = cfg
        self.vc, self.pc, self.tc = cfg["vision"], cfg["perceiver"], cfg["text"]
        self.w = {}
        for k, v in weights.items():
            t = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(dtype).clone()
            frozen = k.startswith("model.vision_model.") and not train_vision
            t.requires_grad_(not frozen)
            self.w[k] = t

    @classmethod
    def from_npz(cls, path, **kw):
        z = np.load(path)
        meta = json.loads(str(z["__config__"]))
        return cls({k: z[k] for k in z.files if k != "__config__"}, meta, **kw)

    def zero_grad(self):
        for t in self.w.values():
            t.grad = None

    # ---- vision tower (frozen): NaViT embeddings + encoder + post_layernorm
    def vision(self, pixels, patch_mask):
        w, vc = self.w, self.vc
        P, dv, nh = vc["patch_size"], vc["hidden_size"], vc["num_attention_heads"]
        pre = "model.vision_model."
        x = F.conv2d(pixels, w[pre + "embeddings.patch_embedding.weight"], w[pre + "embeddings.patch_embedding.bias"], stride=P)
        x = x.flatten(2).transpose(1, 2)
        pos = bucketized_position_ids(patch_mask, vc["image_size"] // P)
        x = x + F.embedding(pos, w[pre + "embeddings.position_embedding.weight"])
        km = patch_mask.reshape(patch_mask.shape[0], -1)
        key_mask = None if bool(km.all()) else km.to(torch.int64)          # :749-752
        I, N, _ = x.shape
        hd = dv // nh
        act = ACT[vc["hidden_act"]]
        eps = vc["layer_norm_eps"]
        for i in range(vc["num_hidden_layers"]):
            p = f"{pre}encoder.layers.{i}."
            r = x
            y = layernorm(x, w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], eps)
            q = F.linear(y, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"]).view(I, N, nh, hd).transpose(1, 2)
            k = F.linear(y, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"]).view(I, N, nh, hd).transpose(1, 2)
            v = F.linear(y, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"]).view(I, N, nh, hd).transpose(1, 2)
            a = attention(q, k, v, hd ** -0.5, causal=False, key_mask=key_mask).reshape(I, N, dv)
            x = r + F.linear(a, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
            r = x
            y = layernorm(x, w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], eps)
            y = F.linear(act(F.linear(y, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])), w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
            x = r + y
        return layernorm(x, w[pre + "post_layernorm.weight"], w[pre + "post_layernorm.bias"], eps)

    def _mlp(self, x, p, act):
        w = self.w
        return F.linear(act(F.linear(x, w[p + "gate_proj.weight"])) * F.linear(x, w[p + "up_proj.weight"]), w[p + "down_proj.weight"])

    # ---- connector: modality projection + perceiver resampler
    def connector(self, feats, patch_key_mask, record=None):
        w, pc, tc = self.w, self.pc, self.tc
        pre = "model.connector."
        ctx = self._mlp(feats, pre + "modality_projection.", ACT[tc["hidden_act"]])
        if record is not None:
            record["modality_projection_out"] = ctx
        I, N, d = ctx.shape
        nl, nh, nkv, hd = pc["resampler_n_latents"], pc["resampler_n_heads"], pc["num_key_value_heads"], pc["resampler_head_dim"]
        eps = tc["rms_norm_eps"]
        lat = w[pre + "perceiver_resampler.latents"][None].expand(I, nl, d)
        key_mask = torch.cat([patch_key_mask.to(torch.int64), torch.ones(I, nl, dtype=torch.int64)], dim=1)      # :1293-1296
        act = ACT[pc["hidden_act"]]
        for i in range(pc["resampler_depth"]):
            p = f"{pre}perceiver_resampler.layers.{i}."
            r = lat
            ln = rmsnorm(lat, w[p + "input_latents_norm.weight"], eps)
            cn = rmsnorm(ctx, w[p + "input_context_norm.weight"], eps)
            hs = torch.cat([cn, ln], dim=1)                                                                      # :856
            q = F.linear(ln, w[p + "self_attn.q_proj.weight"]).view(I, nl, nh, hd).transpose(1, 2)
            k = F.linear(hs, w[p + "self_attn.k_proj.weight"]).view(I, N + nl, nkv, hd).transpose(1, 2)
            v = F.linear(hs, w[p + "self_attn.v_proj.weight"]).view(I, N + nl, nkv, hd).transpose(1, 2)
            a = attention(q, k, v, hd ** -0.5, causal=False, key_mask=key_mask).reshape(I, nl, nh * hd)
            lat = r + F.linear(a, w[p + "self_attn.o_proj.weight"])
            r = lat
            lat = r + self._mlp(rmsnorm(lat, w[p + "post_attention_layernorm.weight"], eps), p + "mlp.", act)
        return rmsnorm(lat, w[pre + "perceiver_resampler.norm.weight"], eps)

    # ---- text model (Mistral = the Llama block; default positions arange) + head + loss
    def text(self, x, attention_mask, record=None):
        w, tc = self.w, self.tc
        d, nh, nkv = tc["hidden_size"], tc["num_attention_heads"], tc["num_key_value_heads"]
        hd = tc.get("head_dim") or d // nh
        eps = tc["rms_norm_eps"]
        B, L, _ = x.shape
        pos = torch.arange(L)[None].expand(B, L)
        cos, sin = rope_cos_sin(pos, hd, tc["rope_theta"])
        pre = "model.text_model."
        for i in range(tc["num_hidden_layers"]):
            p = f"{pre}layers.{i}."
            r = x
            y = rmsnorm(x, w[p + "input_layernorm.weight"], eps)
            q = F.linear(y, w[p + "self_attn.q_proj.weight"]).view(B, L, nh, hd).transpose(1, 2)
            k = F.linear(y, w[p + "self_attn.k_proj.weight"]).view(B, L, nkv, hd).transpose(1, 2)
            v = F.linear(y, w[p + "self_attn.v_proj.weight"]).view(B, L, nkv, hd).transpose(1, 2)
            q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
            a = attention(q, k, v, hd ** -0.5, causal=True, key_mask=attention_mask).reshape(B, L, nh * hd)
            x = r + F.linear(a, w[p + "self_attn.o_proj.weight"])
            r = x
            x = r + self._mlp(rmsnorm(x, w[p + "post_attention_layernorm.weight"], eps), p + "mlp.", ACT[tc["hidden_act"]])
            if record is not None:
                record[f"llm_layer{i}_out"] = x
        return rmsnorm(x, w[pre + "norm.weight"], eps)

    def forward(self, input_ids, pixel_values, pixel_attention_mask, attention_mask, labels, record=None):
        """pixel_values [B, max_images, 3, H, W] (all-zero images are padding, :1636-1639) or None; pixel_attention_mask bool
        [B, max_images, H, W] or None.  Returns (loss, logits)."""
        cfg, w = self.cfg, self.w
        ids = torch.as_tensor(input_ids)
        am = torch.as_tensor(attention_mask)
        emb = F.embedding(ids, w["model.text_model.embed_tokens.weight"])
        if pixel_values is not None:
            pv = torch.as_tensor(pixel_values).float()       # (cast to the weights' dtype below, after the exact == 0 test on the fp32 pixels)
            B, M = pv.shape[:2]
            pv = pv.reshape(B * M, *pv.shape[2:])
            real = (pv == 0.0).sum(dim=(-1, -2, -3)) != pv[0].numel()
            pv = pv[real]
            if pixel_attention_mask is None:
                pm = torch.ones(pv.shape[0], pv.shape[2], pv.shape[3], dtype=torch.bool)
            else:
                pm = torch.as_tensor(pixel_attention_mask).bool().reshape(B * M, *pixel_attention_mask.shape[2:])[real]
            patch_mask = patch_mask_from_pixel_mask(pm, self.vc["patch_size"])
            feats = self.vision(pv.to(self.w["lm_head.weight"].dtype), patch_mask)
            if record is not None:
                record["vision_last_hidden_state"] = feats
            img = self.connector(feats, patch_mask.reshape(patch_mask.shape[0], -1), record)
            if record is not None:
                record["connector_out"] = img
            sel = ids == cfg["image_token_id"]
            rows = img.reshape(-1, img.shape[-1])
            if int(sel.sum()) != rows.shape[0]:
                raise ValueError(f"{int(sel.sum())} <image> tokens for {rows.shape[0]} image hidden states")   # torch's shape-mismatch error
            bi, ti = torch.nonzero(sel, as_tuple=True)
            emb = emb.index_put((bi, ti), rows)                                                                # :1561-1565
            if record is not None:
                record["merged_embeds"] = emb
        h = self.text(emb, am, record)
        logits = F.linear(h, w["lm_head.weight"]).float()
        loss = None
        if labels is not None:
            lab = torch.as_tensor(labels)
            keep = am[..., 1:] != 0
            loss = F.cross_entropy(logits[..., :-1, :][keep], lab[..., 1:][keep], ignore_index=cfg["image_token_id"])   # :1884-1899
        return loss, logits

    def forward_packed(self, input_ids, pixel_values, pixel_attention_mask, segment_ids, attention_mask, labels):
        """Long-sequence packing defined through its meaning (cf. LlavaRef.forward_packed): the packed row is unpacked into its samples,
        each runs through `forward` alone, the loss is the mean over the label positions of all of them.  pixel_values
        [1, n_images, 3, H, W]: images in order of appearance; a sample owns as many as its <image> tokens / n_latents."""
        cfg = self.cfg
        ids, seg, lab, am = (torch.as_tensor(x) for x in (input_ids, segment_ids, labels, attention_mask))
        pv = None if pixel_values is None else torch.as_tensor(pixel_values)
        pm = None if pixel_attention_mask is None else torch.as_tensor(pixel_attention_mask)
        nl = self.pc["resampler_n_latents"]
        total, count, img0 = 0.0, 0, 0
        for sid in torch.unique_consecutive(seg[0]).tolist():
            sel = seg[0] == sid
            si, sl, sa = ids[0][sel][None], lab[0][sel][None], am[0][sel][None]
            n_img = int((si == cfg["image_token_id"]).sum()) // nl
            spv = None if n_img == 0 else pv[:, img0: img0 + n_img]
            spm = None if (n_img == 0 or pm is None) else pm[:, img0: img0 + n_img]
            img0 += n_img
            _, logits = self.forward(si, spv, spm, sa, None)
            keep = sa[:, 1:] != 0
            lg, tg = logits[:, :-1][keep], sl[:, 1:][keep]
            valid = tg != cfg["image_token_id"]
            if int(valid.sum()):
                total = total + F.cross_entropy(lg[valid], tg[valid], reduction="sum")
                count += int(valid.sum())
        return total / max(count, 1)
