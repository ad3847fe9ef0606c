"""ORACLE (test infrastructure, never imported by the product path).

Pure-torch, transformers-free CPU restatement of the reference hot path, used
  * as the checker for the HIP path (tests/, __graft_entry__.smoke()), and
  * as the "port" CPU baseline timed by bench.py (`cpu_baseline.kind == "port"`).

What it restates (`HF:` = the transformers copy the reference runs on):
  row C  vision tower      HF:models/siglip/modeling_siglip.py:116-186,250-358 / HF:models/clip/modeling_clip.py
                           invoked at /root/reference/mantis/models/mllava/modeling_llava.py:456-458
  row D  feature select    modeling_llava.py:460-467
  row E  projector         modeling_llava.py:106-118
  row F  token embedding   modeling_llava.py:427
  row G  packing           modeling_llava.py:293-360   (integer plan: oracle/pack_ref.py)
  row H  Llama decoder     HF:models/llama/modeling_llama.py:53-67,113-176,191-325,367-418
                           invoked at modeling_llava.py:510-519
  row I  lm_head + loss    modeling_llava.py:521-537
  row A/J/L training_step  HF:trainer.py:1892-1963 (forward-loss, /GA, backward, detached loss)

PARITY PIN: the reference ships no tests or golden vectors for this path, so this
restatement is pinned against outputs of the reference itself, run in the build container by
tests/golden/make_golden.py: logits, loss, per-layer activations, packing integers and every
trainable gradient (tests/test_oracle_vs_golden.py; fp32 tolerances stated there).
"""
import json
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import pack_ref


# ----------------------------------------------------------------------------- per-op references
def layernorm(x, w, b, eps):
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).to(x.dtype)


def rmsnorm(x, w, eps):
    # HF:models/llama/modeling_llama.py:60-65 (stats in fp32, cast back, then scale)
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return w * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def gelu_erf(x):
    return F.gelu(x)


def gelu_tanh(x):
    return F.gelu(x, approximate="tanh")


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


ACT = {"gelu": gelu_erf, "gelu_pytorch_tanh": gelu_tanh, "quick_gelu": quick_gelu, "silu": F.silu}


def rope_cos_sin(position_ids, head_dim, theta):
    # HF:models/llama/modeling_llama.py:95-127 (fp32 trig)
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = position_ids[..., None].float() * inv_freq          # [B,L,hd/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x, cos, sin):
    """x [B,H,L,hd]; cos/sin [B,L,hd] (cast to x.dtype first, as HF does)."""
    cos = cos.to(x.dtype)[:, None]
    sin = sin.to(x.dtype)[:, None]
    return x * cos + rotate_half(x) * sin


def attention(q, k, v, scale, causal, key_mask=None):
    """Eager attention, softmax in fp32 (HF:models/llama/modeling_llama.py:191-214,
    HF:models/siglip/modeling_siglip.py:227-247).  q [B,H,Lq,hd], k/v [B,Hkv,Lk,hd];
    key_mask [B,Lk] of {0,1}.  Returns [B,Lq,H,hd]."""
    B, H, Lq, hd = q.shape
    rep = H // k.shape[1]
    if rep > 1:
        k = k.repeat_interleave(rep, dim=1)
        v = v.repeat_interleave(rep, dim=1)
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    neg = torch.finfo(s.dtype).min
    if causal:
        cm = torch.ones(Lq, k.shape[2], dtype=torch.bool).tril()
        s = s.masked_fill(~cm, neg)
    if key_mask is not None:
        s = s.masked_fill(key_mask[:, None, None, :] == 0, neg)
    p = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(p, v).transpose(1, 2).contiguous()


def masked_shift_ce(logits, labels, attention_mask, ignore_index=-100):
    """/root/reference/mantis/models/mllava/modeling_llava.py:521-537."""
    sm = attention_mask[..., 1:] != 0
    sl = logits[..., :-1, :][sm]
    tg = labels[..., 1:][sm]
    return F.cross_entropy(sl.reshape(-1, sl.shape[-1]).float() if sl.dtype != torch.float32 else sl.reshape(-1, sl.shape[-1]),
                           tg.reshape(-1), ignore_index=ignore_index)


# ----------------------------------------------------------------------------- model
class LlavaRef:
    """Functional model over a dict of tensors named like the reference's state_dict (HF-5 flat vision names;
    a 4.x ``vision_tower.vision_model.*`` prefix is accepted)."""

    def __init__(self, weights, cfg, dtype=torch.float32, train_vision=False):
        self.cfg = cfg
        self.dtype = dtype
        self.w = {}
        for k, v in weights.items():
            k = k.replace("vision_tower.vision_model.", "vision_tower.")
            t = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
            t = t.detach().clone().to(dtype)
            trainable = (not k.startswith("vision_tower.")) or train_vision
            t.requires_grad_(trainable)
            self.w[k] = t
        self.vc, self.tc = cfg["vision"], cfg["text"]
        self.is_clip = self.vc["model_type"] == "clip_vision_model"

    @classmethod
    def from_npz(cls, path, **kw):
        z = np.load(path)
        cfg = json.loads(str(z["__config__"]))
        return cls({k: z[k] for k in z.files if k != "__config__"}, cfg, **kw)

    def trainable(self):
        return {k: v for k, v in self.w.items() if v.requires_grad}

    def zero_grad(self):
        for v in self.w.values():
            v.grad = None

    # ---- row C
    def vision_tower(self, pixel_values, n_layers=None):
        w, vc = self.w, self.vc
        P, dv = vc["patch_size"], vc["hidden_size"]
        x = pixel_values.to(self.dtype)
        pe = F.conv2d(x, w["vision_tower.embeddings.patch_embedding.weight"],
                      w.get("vision_tower.embeddings.patch_embedding.bias"), stride=P)
        h = pe.flatten(2).transpose(1, 2)
        if self.is_clip:
            cls = w["vision_tower.embeddings.class_embedding"].expand(h.shape[0], 1, -1)
            h = torch.cat([cls, h], dim=1)
        h = h + w["vision_tower.embeddings.position_embedding.weight"][None]
        eps = vc["layer_norm_eps"]
        if self.is_clip:
            h = layernorm(h, w["vision_tower.pre_layrnorm.weight"], w["vision_tower.pre_layrnorm.bias"], eps)
        nh = vc["num_attention_heads"]
        hd = dv // nh
        act = ACT[vc["hidden_act"]]
        # hidden_states[-2] == output of layer (num_layers-1)  -> run num_layers + vision_feature_layer + 1 layers
        if n_layers is None:
            n_layers = vc["num_hidden_layers"] + self.cfg.get("vision_feature_layer", -2) + 1
        for i in range(n_layers):
            p = f"vision_tower.encoder.layers.{i}."
            r = h
            y = layernorm(h, w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], eps)
            B, N, _ = y.shape
            q = F.linear(y, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"]).view(B, N, nh, hd).transpose(1, 2)
            k = F.linear(y, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"]).view(B, N, nh, hd).transpose(1, 2)
            v = F.linear(y, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"]).view(B, N, nh, hd).transpose(1, 2)
            a = attention(q, k, v, hd ** -0.5, causal=False).reshape(B, N, dv)
            h = r + F.linear(a, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
            r = h
            y = layernorm(h, w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], eps)
            y = F.linear(act(F.linear(y, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])), w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
            h = r + y
        return h

    # ---- row E
    def projector(self, x):
        w = self.w
        act = ACT[self.cfg.get("projector_hidden_act", "gelu")]
        h = F.linear(x, w["multi_modal_projector.linear_1.weight"], w["multi_modal_projector.linear_1.bias"])
        return F.linear(act(h), w["multi_modal_projector.linear_2.weight"], w["multi_modal_projector.linear_2.bias"])

    # ---- row H
    def decoder(self, x, attention_mask, position_ids, n_layers=None, record=None):
        w, tc = self.w, self.tc
        d, nh, nkv = tc["hidden_size"], tc["num_attention_heads"], tc["num_key_value_heads"]
        hd = tc.get("head_dim") or d // nh
        eps = tc["rms_norm_eps"]
        cos, sin = rope_cos_sin(position_ids, hd, tc["rope_theta"])
        B, L, _ = x.shape
        n_layers = tc["num_hidden_layers"] if n_layers is None else n_layers
        for i in range(n_layers):
            p = f"language_model.model.layers.{i}."
            r = x
            y = rmsnorm(x, w[p + "input_layernorm.weight"], eps)
            q = F.linear(y, w[p + "self_attn.q_proj.weight"]).view(B, L, nh, hd).transpose(1, 2)
            k = F.linear(y, w[p + "self_attn.k_proj.weight"]).view(B, L, nkv, hd).transpose(1, 2)
            v = F.linear(y, w[p + "self_attn.v_proj.weight"]).view(B, L, nkv, hd).transpose(1, 2)
            q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
            a = attention(q, k, v, hd ** -0.5, causal=True, key_mask=attention_mask).reshape(B, L, nh * hd)
            x = r + F.linear(a, w[p + "self_attn.o_proj.weight"])
            r = x
            y = rmsnorm(x, w[p + "post_attention_layernorm.weight"], eps)
            y = F.linear(F.silu(F.linear(y, w[p + "mlp.gate_proj.weight"])) * F.linear(y, w[p + "mlp.up_proj.weight"]),
                         w[p + "mlp.down_proj.weight"])
            x = r + y
            if record is not None:
                record[f"llm_layer{i}_out"] = x
        x = rmsnorm(x, w["language_model.model.norm.weight"], eps)
        if record is not None:
            record["llm_final_norm"] = x
        return x

    # ---- rows B..I
    def forward(self, input_ids, pixel_values, attention_mask, labels, record=None, n_vit_layers=None, n_llm_layers=None):
        """pixel_values: list of [n_i,3,H,W] tensors, one tensor, or None.  Returns (loss, logits)."""
        cfg, w = self.cfg, self.w
        ids = torch.as_tensor(input_ids)
        attn = torch.as_tensor(attention_mask)
        lab = None if labels is None else torch.as_tensor(labels)
        emb = F.embedding(ids, w["language_model.model.embed_tokens.weight"])
        pos = None
        if pixel_values is not None:
            if isinstance(pixel_values, (list, tuple)):
                pixel_values = torch.cat([torch.as_tensor(p) for p in pixel_values if p is not None], 0)
            feats = self.vision_tower(torch.as_tensor(pixel_values), n_vit_layers)
            strat = cfg["vision_feature_select_strategy"]
            if strat == "default":
                feats = feats[:, 1:]
            elif strat != "full":
                raise ValueError(f"Unexpected select feature strategy: {strat}")
            if record is not None:
                record["projector_in"] = feats
            img = self.projector(feats)
            if record is not None:
                record["projector_out"] = img
            I, N, d = img.shape
            plan = pack_ref.pack_plan(ids.numpy(), attn.numpy(), None if lab is None else lab.numpy(), I, N,
                                      cfg["image_token_index"], cfg["pad_token_id"] if cfg["pad_token_id"] is not None else -1,
                                      cfg.get("ignore_index", -100),
                                      fix_unequal_counts=bool(cfg.get("fix_unequal_counts", False)))
            B, L = plan["src_kind"].shape
            kind = torch.from_numpy(plan["src_kind"])
            sidx = torch.from_numpy(plan["src_idx"])
            rows = torch.zeros(B, L, d, dtype=emb.dtype)
            bidx = torch.arange(B)[:, None].expand(B, L)
            tk = kind == pack_ref.TEXT
            ik = kind == pack_ref.IMAGE
            # index_put on a fresh tensor: differentiable gather of text rows / image rows
            rows = rows.index_put((bidx[tk], torch.nonzero(tk)[:, 1]), emb[bidx[tk], sidx[tk]])
            rows = rows.index_put((bidx[ik], torch.nonzero(ik)[:, 1]), img.reshape(-1, d)[sidx[ik]])
            emb = rows
            attn = torch.from_numpy(plan["attention_mask"])
            pos = torch.from_numpy(plan["position_ids"])
            lab = torch.from_numpy(plan["labels"]) if lab is not None else torch.full_like(attn, cfg.get("ignore_index", -100))
            self._last_merged_labels, self._last_merged_mask = lab, attn
            if record is not None:
                record.update(merged_embeds=emb, merged_attention_mask=attn, merged_labels=lab, merged_position_ids=pos)
        if pos is None:
            # HF LlamaModel default: arange (no cache)
            pos = torch.arange(emb.shape[1])[None].expand(emb.shape[0], -1)
        h = self.decoder(emb, attn, pos, n_llm_layers, record)
        logits = F.linear(h, w["language_model.lm_head.weight"])
        loss = None
        if lab is not None:
            loss = masked_shift_ce(logits, lab, attn, cfg.get("ignore_index", -100))
        return loss, logits

    def forward_packed(self, input_ids, pixel_values, segment_ids, labels, attention_mask=None):
        """Sample packing (/root/reference/mantis/train/data.py:1546-1671) defined through its meaning: the packed row is UNPACKED
        into its samples, every sample runs through `forward` alone (batch size 1, exactly the reference's mllava collator regime),
        and the loss is the mean over all label positions of all samples -- what the block-diagonal mask + restarted position ids
        compute in one pass, with no prediction across a sample boundary.  Independent of the kstart / qend machinery under test."""
        cfg = self.cfg
        ids = torch.as_tensor(input_ids)
        seg = torch.as_tensor(segment_ids)
        lab = torch.as_tensor(labels)
        am = torch.ones_like(ids) if attention_mask is None else torch.as_tensor(attention_mask)
        if isinstance(pixel_values, (list, tuple)):
            pixel_values = torch.cat([torch.as_tensor(p) for p in pixel_values if p is not None], 0)
        pv = None if pixel_values is None else torch.as_tensor(pixel_values)
        total, count, img0 = 0.0, 0, 0
        ign = cfg.get("ignore_index", -100)
        for b in range(ids.shape[0]):
            for sid in torch.unique_consecutive(seg[b]).tolist():
                sel = seg[b] == sid
                si, sl, sa = ids[b][sel][None], lab[b][sel][None], am[b][sel][None]
                n_img = int((si == cfg["image_token_index"]).sum())
                spv = None if n_img == 0 else pv[img0: img0 + n_img]
                img0 += n_img
                _, logits = self.forward(si, spv, sa, sl)
                # the per-sample numerator / denominator of masked_shift_ce
                plan_lab = self._last_merged_labels if spv is not None else sl
                plan_am = self._last_merged_mask if spv is not None else sa
                keep = plan_am[:, 1:] != 0
                lg = logits[:, :-1][keep]
                tg = plan_lab[:, 1:][keep]
                valid = tg != ign
                if int(valid.sum()):
                    total = total + F.cross_entropy(lg[valid].float(), tg[valid], reduction="sum")
                    count += int(valid.sum())
        return total / max(count, 1)

    # ---- rows A, J, L
    def training_step(self, inputs, gradient_accumulation_steps=1, **fw):
        """HF:trainer.py:1892-1963 restated: loss = model(**inputs).loss; (loss/GA).backward(); return loss.detach()/GA.
        Gradients accumulate in .grad across calls (the caller zeroes them, HF:trainer.py:1796)."""
        loss, _ = self.forward(inputs["input_ids"], inputs.get("pixel_values"), inputs["attention_mask"],
                               inputs["labels"], **fw)
        loss = loss / gradient_accumulation_steps
        loss.backward()
        return loss.detach()


def random_weights(cfg, seed=0, std=0.02, dtype=torch.float32, perturb_1d=0.0):
    """Random-init weights with the reference's shapes/names (normal(0, initializer_range), norm weights 1, biases 0;
    /root/reference/mantis/models/mllava/modeling_llava.py:150-170) for the CPU-baseline leg of bench.py.
    perturb_1d > 0 (cfg1 fixture): afterwards every bias gets N(0, perturb_1d) and every norm weight 1 + N(0, perturb_1d) from a
    second seeded stream, so a dropped bias / norm weight shows up in parity (the main stream is unchanged)."""
    w = _random_weights(cfg, seed, std, dtype)
    if perturb_1d:
        g2 = torch.Generator().manual_seed(seed + 7919)
        for k in sorted(w):
            if w[k].dim() == 1 and "class_embedding" not in k:
                base = 1.0 if ("norm" in k and k.endswith("weight")) else 0.0
                w[k] = (base + perturb_1d * torch.randn(w[k].shape, generator=g2)).to(dtype)
    return w


def _random_weights(cfg, seed, std, dtype):
    g = torch.Generator().manual_seed(seed)
    vc, tc = cfg["vision"], cfg["text"]
    dv, iv, P, C = vc["hidden_size"], vc["intermediate_size"], vc["patch_size"], vc.get("num_channels", 3)
    d, it, V = tc["hidden_size"], tc["intermediate_size"], tc["vocab_size"]
    nh, nkv = tc["num_attention_heads"], tc["num_key_value_heads"]
    hd = tc.get("head_dim") or d // nh
    is_clip = vc["model_type"] == "clip_vision_model"
    npos = (vc["image_size"] // P) ** 2 + (1 if is_clip else 0)
    w = {}

    def rn(*shape):
        return torch.randn(*shape, generator=g, dtype=torch.float32).mul_(std).to(dtype)

    w["vision_tower.embeddings.patch_embedding.weight"] = rn(dv, C, P, P)
    if is_clip:
        w["vision_tower.embeddings.class_embedding"] = rn(dv)
        w["vision_tower.pre_layrnorm.weight"] = torch.ones(dv, dtype=dtype)
        w["vision_tower.pre_layrnorm.bias"] = torch.zeros(dv, dtype=dtype)
    else:
        w["vision_tower.embeddings.patch_embedding.bias"] = torch.zeros(dv, dtype=dtype)
    w["vision_tower.embeddings.position_embedding.weight"] = rn(npos, dv)
    for i in range(vc["num_hidden_layers"]):
        p = f"vision_tower.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[p + f"self_attn.{n}.weight"] = rn(dv, dv)
            w[p + f"self_attn.{n}.bias"] = torch.zeros(dv, dtype=dtype)
        for n in ("layer_norm1", "layer_norm2"):
            w[p + n + ".weight"] = torch.ones(dv, dtype=dtype)
            w[p + n + ".bias"] = torch.zeros(dv, dtype=dtype)
        w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"] = rn(iv, dv), torch.zeros(iv, dtype=dtype)
        w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"] = rn(dv, iv), torch.zeros(dv, dtype=dtype)
    w["multi_modal_projector.linear_1.weight"], w["multi_modal_projector.linear_1.bias"] = rn(d, dv), torch.zeros(d, dtype=dtype)
    w["multi_modal_projector.linear_2.weight"], w["multi_modal_projector.linear_2.bias"] = rn(d, d), torch.zeros(d, dtype=dtype)
    w["language_model.model.embed_tokens.weight"] = rn(V, d)
    for i in range(tc["num_hidden_layers"]):
        p = f"language_model.model.layers.{i}."
        w[p + "self_attn.q_proj.weight"] = rn(nh * hd, d)
        w[p + "self_attn.k_proj.weight"] = rn(nkv * hd, d)
        w[p + "self_attn.v_proj.weight"] = rn(nkv * hd, d)
        w[p + "self_attn.o_proj.weight"] = rn(d, nh * hd)
        w[p + "mlp.gate_proj.weight"], w[p + "mlp.up_proj.weight"] = rn(it, d), rn(it, d)
        w[p + "mlp.down_proj.weight"] = rn(d, it)
        w[p + "input_layernorm.weight"] = torch.ones(d, dtype=dtype)
        w[p + "post_attention_layernorm.weight"] = torch.ones(d, dtype=dtype)
    w["language_model.model.norm.weight"] = torch.ones(d, dtype=dtype)
    w["language_model.lm_head.weight"] = rn(V, d)
    return w
