"""ORACLE (test infrastructure, never imported by the product path): the Qwen2-VL training forward.

Pure-torch fp32, transformers-free CPU restatement of SURVEY.md section 8 row f3 (BASELINE.json configs[4]).  The reference's model for
this path is HuggingFace's class (/root/reference/mantis/models/qwen2_vl/modeling_qwen2_vl.py:1 star-imports it; third-party
dependency `transformers`, absent from /root/reference, version in this image 5.15.0), driven by
/root/reference/mantis/train/train_qwen2_vl.py (pixel budget :126-128, frozen `visual` :209-212).  Restated here, by the file
transformers/models/qwen2_vl/modeling_qwen2_vl.py of that version:
  PatchEmbed (Conv3d with kernel == stride == a GEMM over flattened patches)              :251-274
  2-D vision rotary embedding (h | w halves of the rotary dim), per-image attention       :239-248, :225-236, :342-422, vision_utils
  vision block: LayerNorm, fused qkv with bias, QuickGELU MLP                             :425-450, :293-302
  PatchMerger: LayerNorm, 2x2 merge = 4 consecutive rows, Linear-GELU(erf)-Linear         :277-290
  get_rope_index: 3-D (t, h, w) position ids of the merged sequence                       :862-1018
  image rows masked_scatter'ed over the <|image_pad|> tokens                              :1160-1166
  multimodal RoPE (sections of the rotary dim take t / h / w positions)                   :117-170, :180-222
  Qwen2 decoder layer: RMSNorm, GQA with q/k/v BIAS, SwiGLU                               :453-466, :469-556, :559-625
  lm_head, shifted CE with ignore_index -100 (HF ForCausalLMLoss: mean over labelled positions; NOT filtered by attention_mask)

PARITY PIN: tests/golden/make_golden_qwen2vl.py runs the HF class in the build container and records inputs, weights, activations,
3-D position ids, loss and gradients; tests/test_qwen2vl_oracle.py checks this restatement against them.
"""
import json

import numpy as np
import torch
import torch.nn.functional as F

from .llava_ref import ACT, attention, layernorm, rmsnorm, rotate_half


# ----------------------------------------------------------------------------- integer bookkeeping (bit-exact)
def vision_position_ids(grid_thw, merge):
    """(h, w) patch coordinates in the tower's row order: patches are stored merge-window by merge-window (row-major windows, row-major
    inside a window), so 4 consecutive rows are one 2x2 merge group.  grid_thw int [n, 3] -> int64 [sum t*h*w, 2]."""
    out = []
    for t, h, w in np.asarray(grid_thw).tolist():
        hp = torch.arange(h)[:, None].expand(h, w)
        wp = torch.arange(w)[None, :].expand(h, w)

        def win(p):
            return p.reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
        out.append(torch.stack([win(hp), win(wp)], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def rope_index(input_ids, attention_mask, image_grid_thw, image_token_id, merge):
    """get_rope_index (:914-1018) for images only: int64 [3, B, T].  Text runs count 0,1,2,...; an image of (t, h, w) patches occupies
    t * (h/merge) * (w/merge) tokens with (t, row, col) indices offset by the running position, and advances it by
    max(h, w) / merge.  Positions where attention_mask == 0 stay 0."""
    ids = np.asarray(input_ids)
    B, T = ids.shape
    am = np.ones_like(ids) if attention_mask is None else np.asarray(attention_mask)
    grids = [] if image_grid_thw is None else np.asarray(image_grid_thw).tolist()
    pos = np.zeros((3, B, T), np.int64)
    gi = 0
    for b in range(B):
        keep = am[b] != 0
        row = ids[b][keep]
        is_img = row == image_token_id
        cur, cols, i = 0, [], 0
        n = len(row)
        while i < n:
            j = i
            while j < n and is_img[j] == is_img[i]:
                j += 1
            if not is_img[i]:
                r = np.arange(j - i, dtype=np.int64) + cur
                cols.append(np.stack([r, r, r]))
                cur += j - i
            else:
                # ONE grid per run of image tokens (itertools.groupby over mm_token_type_ids, :986-1008)
                t, h, w = grids[gi]
                gi += 1
                gh, gw = h // merge, w // merge
                tt = np.repeat(np.arange(t, dtype=np.int64), gh * gw)
                hh = np.tile(np.repeat(np.arange(gh, dtype=np.int64), gw), t)
                ww = np.tile(np.arange(gw, dtype=np.int64), t * gh)
                v = np.stack([tt, hh, ww]) + cur
                if v.shape[1] != j - i:
                    raise ValueError(f"image run of {j - i} tokens but grid {t}x{h}x{w} gives {v.shape[1]}")
                cols.append(v)
                cur += max(h, w) // merge
            i = j
        if cols:
            pos[:, b, keep] = np.concatenate(cols, axis=1)
    return torch.from_numpy(pos)


def mrope_cos_sin(position_ids, head_dim, theta, sections):
    """:156-170 + :207-213: cos/sin [B, T, head_dim]; rotary-frequency index f takes the t / h / w position by section."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = position_ids[..., None].float() * inv_freq                 # [3, B, T, hd/2]
    which = torch.repeat_interleave(torch.arange(3), torch.tensor(sections))      # [hd/2]
    sel = torch.stack([freqs[int(which[f]), :, :, f] for f in range(head_dim // 2)], dim=-1)
    emb = torch.cat((sel, sel), dim=-1)
    return emb.cos(), emb.sin()


class Qwen2VLRef:
    def __init__(self, weights, cfg, dtype=torch.float32):
        self.cfg = cfg
        self.vc, self.tc = cfg["vision"], cfg["text"]
        self.w = {}
        for k, v in weights.items():
            t = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(dtype).clone()
            t.requires_grad_(".visual." not in k)                      # train_qwen2_vl.py:209-212
            self.w[k] = t

    @classmethod
    def from_npz(cls, path, **kw):
        z = np.load(path)
        cfg = json.loads(str(z["__config__"]))
        return cls({k: z[k] for k in z.files if k != "__config__"}, cfg, **kw)

    def zero_grad(self):
        for t in self.w.values():
            t.grad = None

    # ---- vision tower + merger (frozen)
    def vision(self, pixel_values, grid_thw, record=None, n_layers=None):
        w, vc = self.w, self.vc
        pre = "model.visual."
        dv, nh = vc["embed_dim"], vc["num_heads"]
        hd = dv // nh
        merge = vc["spatial_merge_size"]
        x = F.linear(pixel_values, w[pre + "patch_embed.proj.weight"].reshape(dv, -1))
        if record is not None:
            record["vision_patch_embed"] = x
        hw = vision_position_ids(grid_thw, merge)                                           # [N, 2]
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))
        rot = (hw[:, :, None].float() * inv).flatten(1)                                     # [N, hd/2]: h freqs | w freqs
        emb = torch.cat((rot, rot), dim=-1)
        cos, sin = emb.cos()[:, None], emb.sin()[:, None]
        lens = [t * h * ww for t, h, ww in np.asarray(grid_thw).tolist()]
        act = ACT[vc["hidden_act"]]
        for i in range(vc["depth"] if n_layers is None else n_layers):
            p = f"{pre}blocks.{i}."
            y = layernorm(x, w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-6)
            qkv = F.linear(y, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"]).view(-1, 3, nh, hd)
            q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
            q = (q * cos + rotate_half(q) * sin).to(v.dtype)           # (HF rotates in fp32 and casts back: apply_rotary_pos_emb_vision)
            k = (k * cos + rotate_half(k) * sin).to(v.dtype)
            outs, s = [], 0
            for n in lens:                                             # per-image attention (cu_seqlens)
                sl = slice(s, s + n)
                o = attention(q[sl].transpose(0, 1)[None], k[sl].transpose(0, 1)[None], v[sl].transpose(0, 1)[None], hd ** -0.5, False)
                outs.append(o[0].reshape(n, dv))
                s += n
            x = x + F.linear(torch.cat(outs), w[p + "attn.proj.weight"], w[p + "attn.proj.bias"])
            y = layernorm(x, w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-6)
            x = x + F.linear(act(F.linear(y, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])), w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
            if record is not None:
                record[f"vision_block{i}_out"] = x
        if record is not None:
            record["vision_last_hidden_state"] = x
        m = pre + "merger."
        y = layernorm(x, w[m + "ln_q.weight"], w[m + "ln_q.bias"], 1e-6).view(-1, dv * merge * merge)
        y = F.gelu(F.linear(y, w[m + "mlp.0.weight"], w[m + "mlp.0.bias"]))
        return F.linear(y, w[m + "mlp.2.weight"], w[m + "mlp.2.bias"])

    # ---- Qwen2 decoder
    def text(self, x, attention_mask, position_ids, record=None, n_layers=None):
        w, tc = self.w, self.tc
        d, nh, nkv = tc["hidden_size"], tc["num_attention_heads"], tc["num_key_value_heads"]
        hd = d // nh
        eps = tc["rms_norm_eps"]
        B, L, _ = x.shape
        rp = tc["rope_parameters"]
        cos, sin = mrope_cos_sin(position_ids, hd, rp["rope_theta"], rp["mrope_section"])
        cos, sin = cos[:, None].to(x.dtype), sin[:, None].to(x.dtype)      # (HF's rotary module returns the tables in the activations' dtype)
        pre = "model.language_model."
        for i in range(tc["num_hidden_layers"] if n_layers is None else n_layers):
            p = f"{pre}layers.{i}."
            y = rmsnorm(x, w[p + "input_layernorm.weight"], eps)
            q = F.linear(y, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"]).view(B, L, nh, hd).transpose(1, 2)
            k = F.linear(y, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"]).view(B, L, nkv, hd).transpose(1, 2)
            v = F.linear(y, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"]).view(B, L, nkv, hd).transpose(1, 2)
            q = q * cos + rotate_half(q) * sin
            k = k * cos + rotate_half(k) * sin
            a = attention(q, k, v, hd ** -0.5, causal=True, key_mask=attention_mask).reshape(B, L, nh * hd)
            x = x + F.linear(a, w[p + "self_attn.o_proj.weight"])
            y = rmsnorm(x, w[p + "post_attention_layernorm.weight"], eps)
            g = F.linear(y, w[p + "mlp.gate_proj.weight"])
            u = F.linear(y, w[p + "mlp.up_proj.weight"])
            x = x + F.linear(ACT[tc["hidden_act"]](g) * u, w[p + "mlp.down_proj.weight"])
            if record is not None:
                record[f"llm_layer{i}_out"] = x
        return rmsnorm(x, w[pre + "norm.weight"], eps)

    def forward(self, input_ids, pixel_values, image_grid_thw, attention_mask, labels, record=None, n_vit_layers=None, n_llm_layers=None):
        """pixel_values fp32 [sum t*h*w, C*tp*p*p] (the processor's flattened patches) or None.  Returns (loss, logits).
        n_vit_layers / n_llm_layers: run only the first n blocks (bench.py's bounded CPU-baseline sample)."""
        cfg, w = self.cfg, self.w
        ids = torch.as_tensor(input_ids)
        am = torch.ones_like(ids) if attention_mask is None else torch.as_tensor(attention_mask)
        emb = F.embedding(ids, w["model.language_model.embed_tokens.weight"])
        B, T = ids.shape
        if pixel_values is not None:
            # (pixels in the weights' dtype: fp32 for the parity oracle, bf16 for tests/golden/make_bf16_envelope.py)
            img = self.vision(torch.as_tensor(pixel_values).to(self.w["lm_head.weight"].dtype), image_grid_thw, record, n_vit_layers)
            if record is not None:
                record["vision_merged"] = img
            sel = ids == cfg["image_token_id"]
            if int(sel.sum()) != img.shape[0]:
                raise ValueError(f"Image features and image tokens do not match, tokens: {int(sel.sum())}, features: {img.shape[0]}")
            bi, ti = torch.nonzero(sel, as_tuple=True)
            emb = emb.index_put((bi, ti), img)
            pos = rope_index(ids, am, image_grid_thw, cfg["image_token_id"], self.vc["spatial_merge_size"])
        else:
            pos = torch.arange(T)[None, None].expand(3, B, T)              # text only: the text model counts 0..T-1 itself
        if record is not None:
            record["merged_embeds"] = emb
            record["position_ids"] = pos
        h = self.text(emb, am, pos, record, n_llm_layers)
        logits = F.linear(h, w["lm_head.weight"]).float()
        loss = None
        if labels is not None:
            lab = torch.as_tensor(labels)
            loss = F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), lab[:, 1:].reshape(-1), ignore_index=-100)
        return loss, logits

    def forward_packed(self, input_ids, pixel_values, image_grid_thw, segment_ids, labels):
        """Sample packing defined through its meaning (cf. LlavaRef.forward_packed): the packed row is unpacked into its samples, each
        runs through `forward` alone (its own rope index, its own attention), the loss is the mean over the label positions of all of
        them.  Images are consumed in order of appearance; a sample owns the grids whose merged sizes fill its <|image_pad|> runs."""
        cfg = self.cfg
        ids, seg, lab = (torch.as_tensor(x) for x in (input_ids, segment_ids, labels))
        grids = [] if image_grid_thw is None else np.asarray(image_grid_thw).tolist()
        pv = None if pixel_values is None else torch.as_tensor(pixel_values)
        m2 = self.vc["spatial_merge_size"] ** 2
        total, count, gi, p0 = 0.0, 0, 0, 0
        for sid in torch.unique_consecutive(seg[0]).tolist():
            sel = seg[0] == sid
            si, sl = ids[0][sel][None], lab[0][sel][None]
            need = int((si == cfg["image_token_id"]).sum())
            g0, got, npatch = gi, 0, 0
            while got < need:
                t, h, w = grids[gi]
                got += t * h * w // m2
                npatch += t * h * w
                gi += 1
            spv = None if need == 0 else pv[p0: p0 + npatch]
            p0 += npatch
            _, logits = self.forward(si, spv, None if need == 0 else np.asarray(grids[g0:gi]), None, None)
            lg, tg = logits[0, :-1], sl[0, 1:]
            valid = tg != -100
            if int(valid.sum()):
                total = total + F.cross_entropy(lg[valid], tg[valid], reduction="sum")
                count += int(valid.sum())
        return total / max(count, 1)
