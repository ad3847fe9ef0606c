"""Drop-in module for the reference's `LlavaForConditionalGeneration`
(/root/reference/mantis/models/mllava/modeling_llava.py:251-262 constructor, :364-549 forward): same forward keyword
arguments, same output fields, same parameter names (state_dict-compatible with the reference / HF checkpoints), but

  * every parameter is a VIEW into one flat bf16 arena in HBM (and every gradient a view into one flat gradient arena):
    q|k|v and gate|up are stored adjacently so the fused projections are zero-copy views, data-parallel buckets are
    contiguous slices, and the optimizer is a single launch over the arena;
  * forward+backward run through `LlavaEngine` (hand-written gfx950 kernels via the C-ABI), not autograd.

`forward(..., labels=...)` in training mode with grad enabled runs the fused forward+backward and returns a loss whose
`.backward()` (called by a stock `transformers.Trainer`) publishes the already-computed gradients scaled by the incoming
gradient; `MantisHipTrainer.training_step` (trainer.py) skips even that and accumulates straight into `.grad`.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import nn

from .arena import ArenaModule, numel as _numel
from .configuration_llava import LlavaConfig
from . import engine as _engine


@dataclass
class LlavaCausalLMOutputWithPast:
    """modeling_llava.py:63-103"""
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[Tuple] = None
    hidden_states: Optional[Tuple] = None
    attentions: Optional[Tuple] = None
    image_hidden_states: Optional[Tuple] = None

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions,
                                 self.image_hidden_states) if v is not None)[k]

    def keys(self):
        return [k for k in ("loss", "logits", "past_key_values", "hidden_states", "attentions", "image_hidden_states")
                if getattr(self, k) is not None]


def _param_specs(cfg: LlavaConfig):
    """(name, shape) in ARENA ORDER: [vision tower | projector | embed | layer 0..n-1 | final norm | lm_head]."""
    vc, tc = cfg.vision_config, cfg.text_config
    dv, iv, P, C = vc.hidden_size, vc.intermediate_size, vc.patch_size, vc.num_channels
    is_clip = vc.model_type == "clip_vision_model"
    npos = (vc.image_size // P) ** 2 + (1 if is_clip else 0)
    s = []
    vtp = "vision_tower."
    if is_clip:
        s.append((vtp + "embeddings.class_embedding", (dv,)))
    s.append((vtp + "embeddings.patch_embedding.weight", (dv, C, P, P)))
    if not is_clip:
        s.append((vtp + "embeddings.patch_embedding.bias", (dv,)))
    s.append((vtp + "embeddings.position_embedding.weight", (npos, dv)))
    if is_clip:
        s += [(vtp + "pre_layrnorm.weight", (dv,)), (vtp + "pre_layrnorm.bias", (dv,))]
    for i in range(vc.num_hidden_layers):
        p = f"{vtp}encoder.layers.{i}."
        s += [(p + "self_attn.q_proj.weight", (dv, dv)), (p + "self_attn.k_proj.weight", (dv, dv)),
              (p + "self_attn.v_proj.weight", (dv, dv)),
              (p + "self_attn.q_proj.bias", (dv,)), (p + "self_attn.k_proj.bias", (dv,)), (p + "self_attn.v_proj.bias", (dv,)),
              (p + "self_attn.out_proj.weight", (dv, dv)), (p + "self_attn.out_proj.bias", (dv,)),
              (p + "layer_norm1.weight", (dv,)), (p + "layer_norm1.bias", (dv,)),
              (p + "layer_norm2.weight", (dv,)), (p + "layer_norm2.bias", (dv,)),
              (p + "mlp.fc1.weight", (iv, dv)), (p + "mlp.fc1.bias", (iv,)),
              (p + "mlp.fc2.weight", (dv, iv)), (p + "mlp.fc2.bias", (dv,))]
    s += [(vtp + "post_layernorm.weight", (dv,)), (vtp + "post_layernorm.bias", (dv,))]
    d, it, V = tc.hidden_size, tc.intermediate_size, tc.vocab_size
    H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    s += [("multi_modal_projector.linear_1.weight", (d, dv)), ("multi_modal_projector.linear_1.bias", (d,)),
          ("multi_modal_projector.linear_2.weight", (d, d)), ("multi_modal_projector.linear_2.bias", (d,))]
    s.append(("language_model.model.embed_tokens.weight", (V, d)))
    for i in range(tc.num_hidden_layers):
        p = f"language_model.model.layers.{i}."
        # order inside a layer = REVERSE of backward completion, so that the three DP sub-buckets of a layer are contiguous
        # slices: [norms | q k v o] (done last), [gate up], [down] (done first) -- see grad_buckets()
        s += [(p + "input_layernorm.weight", (d,)), (p + "post_attention_layernorm.weight", (d,)),
              (p + "self_attn.q_proj.weight", (H * hd, d)), (p + "self_attn.k_proj.weight", (Hkv * hd, d)),
              (p + "self_attn.v_proj.weight", (Hkv * hd, d)), (p + "self_attn.o_proj.weight", (d, H * hd)),
              (p + "mlp.gate_proj.weight", (it, d)), (p + "mlp.up_proj.weight", (it, d)),
              (p + "mlp.down_proj.weight", (d, it))]
    s += [("language_model.model.norm.weight", (d,)), ("language_model.lm_head.weight", (V, d))]
    return s


class LlavaForConditionalGeneration(ArenaModule):
    config_class = LlavaConfig
    supports_gradient_checkpointing = True       # per decoder layer (decoder.decoder_forward(checkpoint=True)); the frozen tower saves nothing anyway
    frozen_prefixes = ("vision_tower.",)      # train_mllava.py:240-242

    def __init__(self, config: LlavaConfig, device=None, dtype=torch.bfloat16, init="normal", seed=0):
        super().__init__()
        self.config = config
        self.vocab_size = config.vocab_size
        self.pad_token_id = config.pad_token_id if config.pad_token_id is not None else -1
        self._init_arena(_param_specs(config), device, dtype)
        self._trainable_start = self._offs["multi_modal_projector.linear_1.weight"]
        self.engine = _engine.LlavaEngine(self)
        self._build_views()
        if init == "normal":
            self.reset_parameters(seed)
        self.train()

    def _build_views(self):
        cfg, vc, tc = self.config, self.config.vision_config, self.config.text_config
        g = lambda n: self._param(n).data
        dv, P, C = vc.hidden_size, vc.patch_size, vc.num_channels
        is_clip = vc.model_type == "clip_vision_model"
        vtp = "vision_tower."
        kraw = C * P * P
        kp = (kraw + 7) // 8 * 8

        def patch_w_padded():
            w = g(vtp + "embeddings.patch_embedding.weight").view(dv, kraw)
            if kp == kraw:
                return w
            out = torch.zeros((dv, kp), dtype=w.dtype, device=w.device)
            out[:, :kraw] = w
            return out
        layers = []
        for i in range(vc.num_hidden_layers):
            p = f"{vtp}encoder.layers.{i}."
            layers.append(dict(
                qkv_w=self._flat(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", 3 * dv, dv),
                qkv_b=self._flat(p + "self_attn.q_proj.bias", p + "self_attn.v_proj.bias", 1, 3 * dv).view(-1),
                out_w=g(p + "self_attn.out_proj.weight"), out_b=g(p + "self_attn.out_proj.bias"),
                ln1_w=g(p + "layer_norm1.weight"), ln1_b=g(p + "layer_norm1.bias"),
                ln2_w=g(p + "layer_norm2.weight"), ln2_b=g(p + "layer_norm2.bias"),
                fc1_w=g(p + "mlp.fc1.weight"), fc1_b=g(p + "mlp.fc1.bias"),
                fc2_w=g(p + "mlp.fc2.weight"), fc2_b=g(p + "mlp.fc2.bias")))
        self.vt = dict(patch_kp=kp, patch_w_padded=patch_w_padded,
                       patch_b=None if is_clip else g(vtp + "embeddings.patch_embedding.bias"),
                       pos=g(vtp + "embeddings.position_embedding.weight"),
                       cls=g(vtp + "embeddings.class_embedding") if is_clip else None,
                       pre_ln=(g(vtp + "pre_layrnorm.weight"), g(vtp + "pre_layrnorm.bias")) if is_clip else None,
                       layers=layers)
        self.proj = dict(w1=g("multi_modal_projector.linear_1.weight"), b1=g("multi_modal_projector.linear_1.bias"),
                         w2=g("multi_modal_projector.linear_2.weight"), b2=g("multi_modal_projector.linear_2.bias"))
        d, it = tc.hidden_size, tc.intermediate_size
        H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
        ll = []
        for i in range(tc.num_hidden_layers):
            p = f"language_model.model.layers.{i}."
            ll.append(dict(qkv=self._flat(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (H + 2 * Hkv) * hd, d),
                           o=g(p + "self_attn.o_proj.weight"),
                           gu=self._flat(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", 2 * it, d),
                           down=g(p + "mlp.down_proj.weight"),
                           ln1=g(p + "input_layernorm.weight"), ln2=g(p + "post_attention_layernorm.weight")))
        self.lm = dict(embed=g("language_model.model.embed_tokens.weight"), layers=ll,
                       norm=g("language_model.model.norm.weight"), head=g("language_model.lm_head.weight"))

    @torch.no_grad()
    def reset_parameters(self, seed=0):
        """normal(0, initializer_range), norm weights 1, biases 0 (modeling_llava.py:150-170 / HF _init_weights)."""
        std = self.config.text_config.get("initializer_range", 0.02)
        self._param_version += 1
        gen = torch.Generator(device=self.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if p.dim() == 1 and "class_embedding" not in name:
                p.fill_(1.0 if ("norm" in name and name.endswith("weight")) else 0.0)
            else:
                # chunked to bound the fp32 temporary (lm_head is 0.5 G elements)
                flat = p.view(-1)
                step = 1 << 26
                for a in range(0, flat.numel(), step):
                    b = min(flat.numel(), a + step)
                    flat[a:b] = torch.randn(b - a, generator=gen, device=self.device, dtype=torch.float32).mul_(std)

    # ------------------------------------------------------------------ reference-compatible accessors
    def get_input_embeddings(self):
        return self.language_model.model.embed_tokens

    def get_output_embeddings(self):
        return self.language_model.lm_head

    def tie_weights(self, *a, **k):
        return None

    def load_reference_state_dict(self, sd, strict=True):
        """Accepts the reference's state_dict names (HF-5 flat `vision_tower.*` or 4.x `vision_tower.vision_model.*`);
        pooling-head tensors of SigLIP (dead on this path) are ignored."""
        return self.copy_state_dict(sd, rename=lambda k: k.replace("vision_tower.vision_model.", "vision_tower."),
                                    ignorable=lambda k: ".head." in k or k.endswith("position_ids"), strict=strict)

    def _build_grad_views(self):
        tc = self.config.text_config
        gv = self._grad_views.get
        d, it = tc.hidden_size, tc.intermediate_size
        H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
        self.grads = dict(head=gv("language_model.lm_head.weight"), norm=gv("language_model.model.norm.weight"),
                          embed=gv("language_model.model.embed_tokens.weight"),
                          w1=gv("multi_modal_projector.linear_1.weight"), b1=gv("multi_modal_projector.linear_1.bias"),
                          w2=gv("multi_modal_projector.linear_2.weight"), b2=gv("multi_modal_projector.linear_2.bias"))
        self.grads_layers = []
        for i in range(tc.num_hidden_layers):
            p = f"language_model.model.layers.{i}."
            self.grads_layers.append(dict(
                qkv=self._gflat(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (H + 2 * Hkv) * hd, d),
                o=gv(p + "self_attn.o_proj.weight"),
                gu=self._gflat(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", 2 * it, d),
                down=gv(p + "mlp.down_proj.weight"), ln1=gv(p + "input_layernorm.weight"),
                ln2=gv(p + "post_attention_layernorm.weight")))

    def grad_buckets(self):
        """Contiguous slices of the gradient arena in the order backward completes them (for the DP reducer):
        'head' (final norm + lm_head); per decoder layer n-1 .. 0 three sub-buckets fired as their last dW lands --
        ('layer', i, 'down') after dW(down_proj), ('layer', i, 'gu') after dW(gate|up), ('layer', i, 'attn') = norms + q|k|v|o at
        the end of the layer -- so the first byte of a layer moves after its first GEMM, not after its last; 'front'
        (projector + embedding)."""
        self._ensure_grad_arena()
        span = self._bucket_span
        out = {"head": span(lambda n: n.startswith("language_model.model.norm") or n.startswith("language_model.lm_head"))}
        for i in range(self.config.text_config.num_hidden_layers):
            p = f"language_model.model.layers.{i}."
            out[("layer", i, "down")] = span(lambda n: n == p + "mlp.down_proj.weight")
            out[("layer", i, "gu")] = span(lambda n: n in (p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"))
            out[("layer", i, "attn")] = span(lambda n: n.startswith(p) and ".mlp." not in n)
        out["front"] = span(lambda n: n.startswith("multi_modal_projector.") or n.startswith("language_model.model.embed_tokens"))
        return {k: v for k, v in out.items() if v is not None}

    # ------------------------------------------------------------------ forward (contract 2 of SURVEY.md section 8b)
    def forward(self, input_ids=None, pixel_values=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, vision_feature_layer=None, vision_feature_select_strategy=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, return_logits=None,
                _record=None, segment_ids=None):
        if inputs_embeds is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("generation / KV-cache paths are out of scope (SURVEY.md section 2); training forward only")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states are not produced by the fused kernels")
        if vision_feature_layer is not None and vision_feature_layer != self.config.vision_feature_layer:
            raise NotImplementedError("per-call vision_feature_layer override")
        if vision_feature_select_strategy is not None and vision_feature_select_strategy != self.config.vision_feature_select_strategy:
            if vision_feature_select_strategy not in ("default", "full"):
                raise ValueError(f"Unexpected select feature strategy: {vision_feature_select_strategy}")
            raise NotImplementedError("per-call vision_feature_select_strategy override")
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        if attention_mask is not None and attention_mask.dim() == 4:
            # the reference's packed batch (mantis/train/data.py:1609-1671): block-diagonal 4-D mask + restarting position ids
            from .data import segments_from_packed
            segment_ids, attention_mask = segments_from_packed(dict(attention_mask=attention_mask, position_ids=position_ids,
                                                                    segment_ids=segment_ids))
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        want_grads = self.training and labels is not None and torch.is_grad_enabled() and \
            any(p.requires_grad for p in self.parameters())
        if return_logits is None:
            return_logits = not want_grads        # training: the [B,L,V] logits are never materialised unless asked for
        if want_grads:
            loss = self._autograd_step(lambda: self.engine.step(
                input_ids, attention_mask, labels, pixel_values, grad_scale=1.0, loss_scale=1.0, compute_grads=True, overwrite_grads=True,
                need_logits=return_logits, record=_record, segment_ids=segment_ids))       # arena.FusedStep
            logits = self._last_logits
        else:
            out = self.engine.step(input_ids, attention_mask, labels, pixel_values, compute_grads=False,
                                   need_logits=return_logits, record=_record, segment_ids=segment_ids)
            loss = None if labels is None else out["loss"].reshape(())
            logits = out["logits"]
        if not return_dict:
            return ((loss,) if loss is not None else ()) + ((logits,) if logits is not None else ())
        return LlavaCausalLMOutputWithPast(loss=loss, logits=logits)
