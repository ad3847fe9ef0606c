"""Drop-in module for the reference fork's `Idefics2ForConditionalGeneration`
(/root/reference/mantis/models/idefics2/modeling_idefics2.py:1729-1912 over Idefics2Model :1487-1722): same forward keyword arguments,
same output fields, same parameter names (state_dict-compatible), on the flat-arena layout of `ArenaModule` and the hand-written gfx950
kernels (no autograd graph): NaViT SigLIP tower (frozen, as under the reference's LoRA target list train_idefics2.py:156), modality
projection + perceiver resampler, `inputs_merger`, Mistral-7B decoder, fp32 cross-entropy with ignore_index = image_token_id.

Documented divergences: full-parameter training of connector + text model instead of LoRA adapters; the vision tower is frozen;
`do_image_splitting` / generation / KV-cache paths are out of scope."""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import decoder as D
from . import decoder_fp8 as D8
from . import hip_ops as K   # tests may monkeypatch `modeling_idefics2.K` with the oracle's operators to test the host logic
from .arena import ArenaModule
from .launch import launch_context
from .configuration_idefics2 import Idefics2Config


@dataclass
class Idefics2CausalLMOutputWithPast:
    """modeling_idefics2.py:119-152"""
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[Tuple] = None
    hidden_states: Optional[Tuple] = None
    attentions: Optional[Tuple] = None
    image_hidden_states: Optional[torch.Tensor] = None

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions,
                                 self.image_hidden_states) if v is not None)[k]


def _param_specs(cfg: Idefics2Config):
    """(name, shape) in ARENA ORDER: [vision tower | connector | embed | text layer 0..n-1 | final norm | lm_head]."""
    vc, pc, tc = cfg.vision_config, cfg.perceiver_config, cfg.text_config
    dv, iv, P, C = vc.hidden_size, vc.intermediate_size, vc.patch_size, vc.num_channels
    s = []
    v = "model.vision_model."
    s += [(v + "embeddings.patch_embedding.weight", (dv, C, P, P)), (v + "embeddings.patch_embedding.bias", (dv,)),
          (v + "embeddings.position_embedding.weight", ((vc.image_size // P) ** 2, dv))]
    for i in range(vc.num_hidden_layers):
        p = f"{v}encoder.layers.{i}."
        s += [(p + "self_attn.q_proj.weight", (dv, dv)), (p + "self_attn.k_proj.weight", (dv, dv)), (p + "self_attn.v_proj.weight", (dv, dv)),
              (p + "self_attn.q_proj.bias", (dv,)), (p + "self_attn.k_proj.bias", (dv,)), (p + "self_attn.v_proj.bias", (dv,)),
              (p + "self_attn.out_proj.weight", (dv, dv)), (p + "self_attn.out_proj.bias", (dv,)),
              (p + "layer_norm1.weight", (dv,)), (p + "layer_norm1.bias", (dv,)), (p + "layer_norm2.weight", (dv,)), (p + "layer_norm2.bias", (dv,)),
              (p + "mlp.fc1.weight", (iv, dv)), (p + "mlp.fc1.bias", (iv,)), (p + "mlp.fc2.weight", (dv, iv)), (p + "mlp.fc2.bias", (dv,))]
    s += [(v + "post_layernorm.weight", (dv,)), (v + "post_layernorm.bias", (dv,))]
    d, it, V = tc.hidden_size, tc.intermediate_size, tc.vocab_size
    H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    c = "model.connector."
    s += [(c + "modality_projection.gate_proj.weight", (it, dv)), (c + "modality_projection.up_proj.weight", (it, dv)),
          (c + "modality_projection.down_proj.weight", (d, it)), (c + "perceiver_resampler.latents", (pc.resampler_n_latents, d))]
    ph, pkv, phd = pc.resampler_n_heads, pc.num_key_value_heads, pc.resampler_head_dim
    for i in range(pc.resampler_depth):
        p = f"{c}perceiver_resampler.layers.{i}."
        s += [(p + "input_latents_norm.weight", (d,)), (p + "input_context_norm.weight", (d,)), (p + "post_attention_layernorm.weight", (d,)),
              (p + "self_attn.q_proj.weight", (ph * phd, d)), (p + "self_attn.k_proj.weight", (pkv * phd, d)),
              (p + "self_attn.v_proj.weight", (pkv * phd, d)), (p + "self_attn.o_proj.weight", (d, ph * phd)),
              (p + "mlp.gate_proj.weight", (4 * d, d)), (p + "mlp.up_proj.weight", (4 * d, d)), (p + "mlp.down_proj.weight", (d, 4 * d))]
    s += [(c + "perceiver_resampler.norm.weight", (d,))]
    t = "model.text_model."
    s.append((t + "embed_tokens.weight", (V, d)))
    for i in range(tc.num_hidden_layers):
        p = f"{t}layers.{i}."
        # same in-layer order as the LLaVA path: [norms | q k v o] [gate up] [down] = reverse of backward completion (DP sub-buckets)
        s += [(p + "input_layernorm.weight", (d,)), (p + "post_attention_layernorm.weight", (d,)),
              (p + "self_attn.q_proj.weight", (H * hd, d)), (p + "self_attn.k_proj.weight", (Hkv * hd, d)),
              (p + "self_attn.v_proj.weight", (Hkv * hd, d)), (p + "self_attn.o_proj.weight", (d, H * hd)),
              (p + "mlp.gate_proj.weight", (it, d)), (p + "mlp.up_proj.weight", (it, d)), (p + "mlp.down_proj.weight", (d, it))]
    s += [(t + "norm.weight", (d,)), ("lm_head.weight", (V, d))]
    return s


class Idefics2ForConditionalGeneration(ArenaModule):
    config_class = Idefics2Config
    supports_gradient_checkpointing = True       # per decoder layer (decoder.decoder_forward(checkpoint=True)); the frozen tower saves nothing anyway
    frozen_prefixes = ("model.vision_model.",)

    def __init__(self, config: Idefics2Config, device=None, dtype=torch.bfloat16, init="normal", seed=0):
        super().__init__()
        self.config = config
        self.image_token_id = config.image_token_id
        self.vocab_size = config.vocab_size
        self._init_arena(_param_specs(config), device, dtype)
        self.engine = Idefics2Engine(self)
        self._build_views()
        if init == "normal":
            self.reset_parameters(seed)
        self.train()

    def _build_views(self):
        cfg, vc, pc, tc = self.config, self.config.vision_config, self.config.perceiver_config, self.config.text_config
        g = lambda n: self._param(n).data
        dv, P, C = vc.hidden_size, vc.patch_size, vc.num_channels
        v = "model.vision_model."
        kraw = C * P * P
        kp = (kraw + 7) // 8 * 8

        def patch_w_padded():
            w = g(v + "embeddings.patch_embedding.weight").view(dv, kraw)
            if kp == kraw:
                return w
            out = torch.zeros((dv, kp), dtype=w.dtype, device=w.device)
            out[:, :kraw] = w
            return out
        layers = []
        for i in range(vc.num_hidden_layers):
            p = f"{v}encoder.layers.{i}."
            layers.append(dict(
                qkv_w=self._flat(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", 3 * dv, dv),
                qkv_b=self._flat(p + "self_attn.q_proj.bias", p + "self_attn.v_proj.bias", 1, 3 * dv).view(-1),
                out_w=g(p + "self_attn.out_proj.weight"), out_b=g(p + "self_attn.out_proj.bias"),
                ln1_w=g(p + "layer_norm1.weight"), ln1_b=g(p + "layer_norm1.bias"), ln2_w=g(p + "layer_norm2.weight"), ln2_b=g(p + "layer_norm2.bias"),
                fc1_w=g(p + "mlp.fc1.weight"), fc1_b=g(p + "mlp.fc1.bias"), fc2_w=g(p + "mlp.fc2.weight"), fc2_b=g(p + "mlp.fc2.bias")))
        self.vt = dict(patch_kp=kp, patch_w_padded=patch_w_padded, patch_b=g(v + "embeddings.patch_embedding.bias"),
                       pos=g(v + "embeddings.position_embedding.weight"), layers=layers,
                       post_ln=(g(v + "post_layernorm.weight"), g(v + "post_layernorm.bias")))
        d, it = tc.hidden_size, tc.intermediate_size
        c = "model.connector."
        ph, pkv, phd = pc.resampler_n_heads, pc.num_key_value_heads, pc.resampler_head_dim
        pl = []
        for i in range(pc.resampler_depth):
            p = f"{c}perceiver_resampler.layers.{i}."
            pl.append(dict(lat_norm=g(p + "input_latents_norm.weight"), ctx_norm=g(p + "input_context_norm.weight"),
                           post_norm=g(p + "post_attention_layernorm.weight"),
                           qkv=self._flat(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (ph + 2 * pkv) * phd, d),
                           o=g(p + "self_attn.o_proj.weight"),
                           gu=self._flat(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", 8 * d, d), down=g(p + "mlp.down_proj.weight")))
        self.conn = dict(mp_gu=self._flat(c + "modality_projection.gate_proj.weight", c + "modality_projection.up_proj.weight", 2 * it, dv),
                         mp_down=g(c + "modality_projection.down_proj.weight"), latents=g(c + "perceiver_resampler.latents"),
                         layers=pl, norm=g(c + "perceiver_resampler.norm.weight"))
        H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
        t = "model.text_model."
        ll = []
        for i in range(tc.num_hidden_layers):
            p = f"{t}layers.{i}."
            ll.append(dict(qkv=self._flat(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (H + 2 * Hkv) * hd, d),
                           o=g(p + "self_attn.o_proj.weight"),
                           gu=self._flat(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", 2 * it, d), down=g(p + "mlp.down_proj.weight"),
                           ln1=g(p + "input_layernorm.weight"), ln2=g(p + "post_attention_layernorm.weight")))
        self.lm = dict(embed=g(t + "embed_tokens.weight"), layers=ll, norm=g(t + "norm.weight"), head=g("lm_head.weight"))

    def _build_grad_views(self):
        pc, tc = self.config.perceiver_config, self.config.text_config
        gv = self._grad_views.get
        d, it, dv = tc.hidden_size, tc.intermediate_size, self.config.vision_config.hidden_size
        H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
        t, c = "model.text_model.", "model.connector."
        self.grads = dict(head=gv("lm_head.weight"), norm=gv(t + "norm.weight"), embed=gv(t + "embed_tokens.weight"))
        self.grads_layers = []
        for i in range(tc.num_hidden_layers):
            p = f"{t}layers.{i}."
            self.grads_layers.append(dict(
                qkv=self._gflat(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (H + 2 * Hkv) * hd, d), o=gv(p + "self_attn.o_proj.weight"),
                gu=self._gflat(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", 2 * it, d), down=gv(p + "mlp.down_proj.weight"),
                ln1=gv(p + "input_layernorm.weight"), ln2=gv(p + "post_attention_layernorm.weight")))
        ph, pkv, phd = pc.resampler_n_heads, pc.num_key_value_heads, pc.resampler_head_dim
        pl = []
        for i in range(pc.resampler_depth):
            p = f"{c}perceiver_resampler.layers.{i}."
            pl.append(dict(lat_norm=gv(p + "input_latents_norm.weight"), ctx_norm=gv(p + "input_context_norm.weight"),
                           post_norm=gv(p + "post_attention_layernorm.weight"),
                           qkv=self._gflat(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (ph + 2 * pkv) * phd, d),
                           o=gv(p + "self_attn.o_proj.weight"),
                           gu=self._gflat(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", 8 * d, d), down=gv(p + "mlp.down_proj.weight")))
        self.grads_conn = dict(mp_gu=self._gflat(c + "modality_projection.gate_proj.weight", c + "modality_projection.up_proj.weight", 2 * it, dv),
                               mp_down=gv(c + "modality_projection.down_proj.weight"), latents=gv(c + "perceiver_resampler.latents"),
                               layers=pl, norm=gv(c + "perceiver_resampler.norm.weight"))

    @torch.no_grad()
    def reset_parameters(self, seed=0):
        """normal(0, initializer_range), norm weights 1, biases 0, latents 1 (modeling_idefics2.py:1366-1387, :1275)."""
        std = self.config.text_config.get("initializer_range", 0.02)
        self._param_version += 1
        gen = torch.Generator(device=self.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith("perceiver_resampler.latents"):
                p.fill_(1.0)
            elif p.dim() == 1:
                p.fill_(1.0 if ("norm" in name and name.endswith("weight")) else 0.0)
            else:
                flat = p.view(-1)
                step = 1 << 26
                for a in range(0, flat.numel(), step):
                    b = min(flat.numel(), a + step)
                    flat[a:b] = torch.randn(b - a, generator=gen, device=self.device, dtype=torch.float32).mul_(std)

    def get_input_embeddings(self):
        return self.model.text_model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    def tie_weights(self, *a, **k):
        return None

    def load_reference_state_dict(self, sd, strict=True):
        return self.copy_state_dict(sd, ignorable=lambda k: k.endswith("position_ids") or ".head." in k, strict=strict)

    def grad_buckets(self):
        """Contiguous gradient-arena slices in backward-completion order: 'head', per text layer n-1..0 ('layer', i, 'down' | 'gu' |
        'attn'), then 'front' = connector (modality projection + perceiver) + token embedding."""
        self._ensure_grad_arena()
        span = self._bucket_span
        out = {"head": span(lambda n: n.startswith("model.text_model.norm") or n.startswith("lm_head"))}
        for i in range(self.config.text_config.num_hidden_layers):
            p = f"model.text_model.layers.{i}."
            out[("layer", i, "down")] = span(lambda n: n == p + "mlp.down_proj.weight")
            out[("layer", i, "gu")] = span(lambda n: n in (p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"))
            out[("layer", i, "attn")] = span(lambda n: n.startswith(p) and ".mlp." not in n)
        out["front"] = span(lambda n: n.startswith("model.connector.") or n.startswith("model.text_model.embed_tokens"))
        return {k: v for k, v in out.items() if v is not None}

    # ------------------------------------------------------------------ forward (modeling_idefics2.py:1797-1912)
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, pixel_values=None,
                pixel_attention_mask=None, image_hidden_states=None, labels=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, return_logits=None, _record=None):
        if inputs_embeds is not None or past_key_values is not None or use_cache or image_hidden_states is not None:
            raise NotImplementedError("generation / KV-cache / precomputed image states are out of scope: training forward only")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states are not produced by the fused kernels")
        if position_ids is not None:
            raise NotImplementedError("explicit position_ids (the reference passes None: Mistral's default arange)")
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        want_grads = self.training and labels is not None and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if return_logits is None:
            return_logits = not want_grads        # training: the [B,L,V] logits are never materialised unless asked for
        if want_grads:
            # stock `loss.backward()` callers: the fused forward+backward runs now, .backward() publishes the gradients (arena.FusedStep)
            loss = self._autograd_step(lambda: self.engine.step(
                input_ids, attention_mask, labels, pixel_values, pixel_attention_mask, grad_scale=1.0, loss_scale=1.0, compute_grads=True,
                overwrite_grads=True, need_logits=return_logits, record=_record))
            lg = self._last_logits
        else:
            out = self.engine.step(input_ids, attention_mask, labels, pixel_values, pixel_attention_mask, compute_grads=False,
                                   need_logits=return_logits, record=_record)
            loss = None if labels is None else out["loss"].reshape(())
            lg = out["logits"]
        logits = None if lg is None else lg.float()          # :1882 logits.float()
        if not return_dict:
            return ((loss,) if loss is not None else ()) + ((logits,) if logits is not None else ())
        return Idefics2CausalLMOutputWithPast(loss=loss, logits=logits)


class Idefics2Engine:
    """Host side of the Idefics2 step: sequences the gfx950 kernels for the vision tower, the connector (forward + backward), the merger
    and -- through decoder.py -- the Mistral decoder, head and loss."""

    def __init__(self, model):
        self.m = model
        self.cfg = model.config
        self._verified = False
        self.w8 = None                       # decoder_fp8.Fp8Weights after model.set_precision("fp8")
        self.weights_unchanged = False       # set by MantisHipTrainer on the 2nd.. micro-batch of an accumulation window

    def step_from_batch(self, inputs, launch=None, **kw):
        """launch: the caller's `launch.LaunchContext`, in force for exactly this call (see engine.LlavaEngine.step_from_batch)"""
        if kw.get("segment_ids") is None and inputs.get("segment_ids") is not None:
            kw["segment_ids"] = inputs["segment_ids"]
        with launch_context(launch):
            return self.step(inputs["input_ids"], inputs["attention_mask"], inputs.get("labels"), inputs.get("pixel_values"),
                             inputs.get("pixel_attention_mask"), **kw)

    # ------------------------------------------------------------------ NaViT image preparation (device kernel + one tiny readback)
    def _bucket_table(self, tab_n, dev):
        """bucket[n][j] = the reference's bucketised fractional coordinate j of n attended patches along one side (:190-210), built ONCE per
        (patches per side, table size) with the reference's own float32 torch ops -- torch.arange with the 0-dim tensor step that
        `1 / nb_patches` is there, torch.bucketize(right=True) -- so the device kernel's table lookups give bit-identical ids."""
        key = (tab_n, str(dev))
        cache = self.__dict__.setdefault("_bucket_cache", {})
        t = cache.get(key)
        if t is None:
            side = self.cfg.vision_config.image_size // self.cfg.vision_config.patch_size
            boundaries = torch.arange(1 / side, 1.0, 1 / side)
            tab = torch.zeros((tab_n, tab_n), dtype=torch.int32)
            for n in range(1, tab_n):
                frac = torch.arange(0, 1 - 1e-6, 1 / torch.tensor(n))
                b = torch.bucketize(frac, boundaries, right=True)
                tab[n, : min(n, b.numel())] = b[:n].to(torch.int32)
            t = cache[key] = tab.to(dev)
        return t

    def _prepare_images(self, pixel_values, pixel_attention_mask):
        """Padding-image removal (:1636-1639), pixel mask -> patch mask (:1653-1658), bucketised NaViT position ids (:190-210) by ONE device
        kernel over the uploaded pixels (K.navit_prepare); the host reads back only the per-image `real` flags and status words (the
        shapes of everything downstream depend on the number of real images) -- round 2 did all of it on the host: a count_nonzero over
        ~38 MB of fp32 pixels and a Python loop with bucketize per image, every step.
        -> (pixels fp32 [I, C, H, W] on the device, pos_ids int32 [I, N], tower key mask int32 [I, N] or None, key mask int32 [I, N])"""
        vc = self.cfg.vision_config
        P = vc.patch_size
        dev = self.m.device
        pv = torch.as_tensor(pixel_values)
        Bm = pv.shape[0] * pv.shape[1]
        pv = pv.reshape(Bm, *pv.shape[2:]).to(dev, non_blocking=True).to(torch.float32).contiguous()
        Hh, Ww = pv.shape[2:]
        pm_in = None
        if pixel_attention_mask is not None:
            pm_in = torch.as_tensor(pixel_attention_mask).reshape(Bm, Hh, Ww).to(dev, non_blocking=True)
        side = vc.image_size // P
        tab_n = max(side, Hh // P, Ww // P) + 1
        real, pm, pos, status = K.navit_prepare(pv, pm_in, P, side, self._bucket_table(tab_n, dev))
        flags = torch.stack([real, status, pm.amin(dim=1)]).cpu()     # the step's one host readback (3 x Bm ints)
        real_h = flags[0] != 0
        if bool(((flags[1] != 0) & real_h).any()):
            raise ValueError("pixel_attention_mask: the attended patches of an image do not form an (rows x columns) grid -- the reference's "
                             "position-id assignment (modeling_idefics2.py:207) fails on such a mask")
        if not bool(real_h.all()):
            idx = torch.nonzero(real_h).reshape(-1).to(dev)
            pv, pm, pos = pv.index_select(0, idx), pm.index_select(0, idx), pos.index_select(0, idx)
        all_attended = bool((flags[2][real_h] != 0).all())        # every patch of every real image attended: the tower needs no key mask
        return pv, pos, (None if all_attended else pm), pm

    def prefetch_vision(self, inputs, after_event=None, stream=None, launch=None):
        """Image preparation + the frozen tower of a FUTURE batch.  The TOWER goes to `stream` (meant to be a lowest-priority stream: its
        workgroups take the compute units the current step leaves idle; `MantisHipTrainer.prefetch_early`); the image PREPARATION -- one
        small kernel whose three ints per image the host reads back, because the shapes downstream depend on the number of real images --
        runs on a plain default-priority side stream: on the lowest-priority queue that kernel would be dispatched behind everything
        already queued on the compute stream and the host would sit in its readback before it could enqueue the current step (round-4
        advisor finding).  The tower is frozen and depends on nothing but the pixels, so `after_event` is not needed for correctness.
        Picked up by the step that is handed the SAME pixel_values object."""
        pv = inputs.get("pixel_values")
        if pv is None or not torch.cuda.is_available():
            return
        dev = self.m.device
        prep = getattr(self, "_side", None)
        if prep is None:
            prep = self._side = torch.cuda.Stream(device=dev)
        if stream is None:
            stream = prep
        with torch.cuda.stream(prep):
            pix, pos_ids, vit_kmask, km_all = self._prepare_images(pv, inputs.get("pixel_attention_mask"))
            prepared = torch.cuda.Event()
            prepared.record(prep)
        if stream is not prep:
            stream.wait_event(prepared)
            for t in (pix, pos_ids, vit_kmask, km_all):          # allocated on `prep`, consumed on `stream`
                if t is not None:
                    t.record_stream(stream)
        with torch.cuda.stream(stream), launch_context(launch):
            feats, N = self.vision_forward(pix, pos_ids, vit_kmask)
            done = torch.cuda.Event()
            done.record(stream)
        slots = self.__dict__.setdefault("_prefetched", {})
        slots[id(pv)] = (pv, feats, N, pix.shape[0], km_all, done)
        while len(slots) > 2:                               # a prefetched batch that never arrives must not pile up
            slots.pop(next(iter(slots)))

    # ------------------------------------------------------------------ vision tower (frozen, forward only)
    def vision_forward(self, pix, pos_ids, kmask):
        m, vc = self.m, self.cfg.vision_config
        vt = m.vt
        I = pix.shape[0]
        P, dv = vc.patch_size, vc.hidden_size
        N = (pix.shape[2] // P) * (pix.shape[3] // P)
        patches = K.im2col(pix, P, vt["patch_kp"])
        pe = K.gemm_nt(patches, vt["patch_w_padded"](), bias=vt["patch_b"])
        x = K.add(pe, K.gather_rows(vt["pos"], pos_ids.reshape(-1)))
        eps = vc.layer_norm_eps
        nh = vc.num_attention_heads
        hd = dv // nh
        for lw in vt["layers"]:
            y = K.layernorm_fwd(x, lw["ln1_w"], lw["ln1_b"], eps)
            qkv = K.gemm_nt(y, lw["qkv_w"], bias=lw["qkv_b"])
            o, _ = K.attn_fwd(qkv, I, N, nh, nh, hd, kmask, hd ** -0.5, False, want_lse=False)
            x = K.gemm_nt(o, lw["out_w"], bias=lw["out_b"], residual=x)
            y = K.layernorm_fwd(x, lw["ln2_w"], lw["ln2_b"], eps)
            hmid = K.gemm_nt(y, lw["fc1_w"], bias=lw["fc1_b"], act=vc.hidden_act)
            x = K.gemm_nt(hmid, lw["fc2_w"], bias=lw["fc2_b"], residual=x)
        return K.layernorm_fwd(x, vt["post_ln"][0], vt["post_ln"][1], eps), N

    # ------------------------------------------------------------------ connector forward (saves what its backward needs)
    def connector_forward(self, feats, I, N, km_all, record=None):
        """feats [I*N, d_v] -> image hidden states [I*n_latents, d].  Perceiver attention (:812-912) as the reference computes it, a true
        cross attention: queries are projected from the 64 latent rows only, keys / values from concat[context, latents] (the fused q|k|v
        parameter is used as its two row slices); round 2 ran a self-attention over all 1088 concatenated rows and threw 1024 of the
        outputs away (17x the attention FLOPs of the block, forward and backward)."""
        m, pc, tc = self.m, self.cfg.perceiver_config, self.cfg.text_config
        cw = m.conn
        dev = feats.device
        nl, H, Hkv, hd = pc.resampler_n_latents, pc.resampler_n_heads, pc.num_key_value_heads, pc.resampler_head_dim
        eps = tc.rms_norm_eps
        gu, a = K.linear_gu_swiglu(feats, cw["mp_gu"])
        ctx = K.gemm_nt(a, cw["mp_down"])
        if record is not None:
            record["modality_projection_out"] = ctx.view(I, N, -1)
        Lk = N + nl
        base = torch.arange(I, dtype=torch.int32)[:, None] * Lk
        ctx_idx = (base + torch.arange(N, dtype=torch.int32)[None]).reshape(-1).to(dev)
        lat_idx = (base + N + torch.arange(nl, dtype=torch.int32)[None]).reshape(-1).to(dev)
        lat_src = torch.arange(nl, dtype=torch.int32).repeat(I).to(dev)
        kmask = torch.cat([km_all.to(dev), torch.ones((I, nl), dtype=torch.int32, device=dev)], dim=1).contiguous()     # :1293-1296
        lat = K.gather_rows(cw["latents"], lat_src)
        saved = []
        for lw in cw["layers"]:
            wq, wkv = lw["qkv"][: H * hd], lw["qkv"][H * hd:]        # q_proj | [k_proj ; v_proj] rows of the fused parameter
            ln, rstd_l = K.rmsnorm_fwd(lat, lw["lat_norm"], eps)
            cn, rstd_c = K.rmsnorm_fwd(ctx, lw["ctx_norm"], eps)
            hs = torch.empty((I * Lk, ctx.shape[1]), dtype=ctx.dtype, device=dev)      # concat[context, latents] per image (:861)
            K.scatter_rows(cn, ctx_idx, I * Lk, out=hs)
            K.scatter_rows(ln, lat_idx, I * Lk, out=hs)
            q = K.gemm_nt(ln, wq)                                      # [I*nl, H*hd]
            kv = K.gemm_nt(hs, wkv)                                    # [I*Lk, 2*Hkv*hd]
            o_lat, lse = K.attn_fwd_cross(q, kv[:, : Hkv * hd], kv[:, Hkv * hd:], I, nl, Lk, H, Hkv, hd, kmask, hd ** -0.5)
            lat_mid = K.gemm_nt(o_lat, lw["o"], residual=lat)
            n2, rstd2 = K.rmsnorm_fwd(lat_mid, lw["post_norm"], eps)
            gu2, a2 = K.linear_gu_swiglu(n2, lw["gu"])
            lat_out = K.gemm_nt(a2, lw["down"], residual=lat_mid)
            saved.append((lat, rstd_l, rstd_c, hs, ln, q, kv, lse, o_lat, lat_mid, rstd2, n2, gu2, a2))
            lat = lat_out
        out, rstd_f = K.rmsnorm_fwd(lat, cw["norm"], eps)
        return out, dict(feats=feats, gu=gu, a=a, ctx=ctx, saved=saved, lat_final=lat, rstd_f=rstd_f, ctx_idx=ctx_idx, lat_idx=lat_idx,
                         kmask=kmask, I=I, N=N, Lk=Lk)

    def connector_backward(self, dimg, c, acc):
        m, pc = self.m, self.cfg.perceiver_config
        cw, g = m.conn, m.grads_conn
        nl, H, Hkv, hd = pc.resampler_n_latents, pc.resampler_n_heads, pc.num_key_value_heads, pc.resampler_head_dim
        I, Lk = c["I"], c["Lk"]
        d_lat = K.rmsnorm_bwd(dimg, c["lat_final"], cw["norm"], c["rstd_f"], None, g["norm"], acc)
        d_ctx = None
        for li in reversed(range(len(cw["layers"]))):
            lw, lg = cw["layers"][li], g["layers"][li]
            lat_in, rstd_l, rstd_c, hs, ln, q, kv, lse, o_lat, lat_mid, rstd2, n2, gu2, a2 = c["saved"].pop()
            wq, wkv = lw["qkv"][: H * hd], lw["qkv"][H * hd:]
            if lg["down"] is not None:
                K.linear_dw(d_lat, a2, lg["down"], acc)
            dgu2 = K.swiglu_bwd(K.linear_dx(d_lat, lw["down"]), gu2)
            if lg["gu"] is not None:
                K.linear_dw(dgu2, n2, lg["gu"], acc)
            dn2 = K.linear_dx(dgu2, lw["gu"])
            d_mid = K.rmsnorm_bwd(dn2, lat_mid, lw["post_norm"], rstd2, d_lat, lg["post_norm"], acc)
            if lg["o"] is not None:
                K.linear_dw(d_mid, o_lat, lg["o"], acc)
            do_lat = K.linear_dx(d_mid, lw["o"])
            dq = torch.empty_like(q)
            dkv = torch.empty_like(kv)
            K.attn_bwd_cross(q, kv[:, : Hkv * hd], kv[:, Hkv * hd:], o_lat, do_lat, lse, dq, dkv[:, : Hkv * hd], dkv[:, Hkv * hd:], I, nl, Lk,
                             H, Hkv, hd, c["kmask"], hd ** -0.5)
            if lg["qkv"] is not None:                                  # the fused parameter's gradient, slice by slice
                K.linear_dw(dq, ln, lg["qkv"][: H * hd], acc)
                K.linear_dw(dkv, hs, lg["qkv"][H * hd:], acc)
            d_hs = K.linear_dx(dkv, wkv)
            d_ln = K.add(K.gather_rows(d_hs, c["lat_idx"]), K.linear_dx(dq, wq))      # latents feed the queries AND their own keys / values
            d_cn = K.gather_rows(d_hs, c["ctx_idx"])
            d_lat = K.rmsnorm_bwd(d_ln, lat_in, lw["lat_norm"], rstd_l, d_mid, lg["lat_norm"], acc)
            d_ctx = K.rmsnorm_bwd(d_cn, c["ctx"], lw["ctx_norm"], rstd_c, d_ctx, lg["ctx_norm"], acc)
        if g["latents"] is not None:                                       # the latents are broadcast over the images (:1289)
            K.colsum(d_lat.view(I, -1), g["latents"].view(-1), acc)
        if g["mp_down"] is not None:
            K.linear_dw(d_ctx, c["a"], g["mp_down"], acc)
        dgu = K.swiglu_bwd(K.linear_dx(d_ctx, cw["mp_down"]), c["gu"])
        if g["mp_gu"] is not None:
            K.linear_dw(dgu, c["feats"], g["mp_gu"], acc)

    # ------------------------------------------------------------------ full step
    def step(self, input_ids, attention_mask, labels, pixel_values, pixel_attention_mask=None, grad_scale=1.0, loss_scale=1.0,
             compute_grads=True, overwrite_grads=True, need_logits=False, record=None, on_bucket_ready=None, segment_ids=None):
        m, cfg, tc = self.m, self.cfg, self.cfg.text_config
        dev = m.device
        ids_cpu = input_ids.detach().to("cpu") if input_ids.device.type != "cpu" else input_ids
        B, T = ids_cpu.shape
        IMG = cfg.image_token_id
        am_cpu = attention_mask.detach().to("cpu") if attention_mask.device.type != "cpu" else attention_mask
        ids_d = input_ids.to(dev, non_blocking=True)
        attn_d = attention_mask.to(dev, non_blocking=True).to(torch.int64)
        lab_d = None if labels is None else labels.to(dev, non_blocking=True).to(torch.int64)
        img = cctx = None
        n_rows = 0
        is_img = ids_cpu == IMG
        if pixel_values is not None:
            pre = self.__dict__.get("_prefetched", {}).pop(id(pixel_values), None)
            if pre is not None and pre[0] is pixel_values and record is None:
                feats, N, I, km_all = pre[1], pre[2], pre[3], pre[4]        # computed ahead on another stream (prefetch_vision)
                torch.cuda.current_stream().wait_event(pre[5])
                for t in (feats, km_all):
                    if t is not None:
                        t.record_stream(torch.cuda.current_stream())
            else:
                pix, pos_ids, vit_kmask, km_all = self._prepare_images(pixel_values, pixel_attention_mask)
                I = pix.shape[0]
                feats, N = self.vision_forward(pix, pos_ids, vit_kmask)
            if record is not None:
                record["vision_last_hidden_state"] = feats.view(I, N, -1)
            img, cctx = self.connector_forward(feats, I, N, km_all, record)
            if record is not None:
                record["connector_out"] = img.view(I, cfg.perceiver_config.resampler_n_latents, -1)
            n_rows = img.shape[0]
            n_tok = int(is_img.sum())
            if n_tok != n_rows:      # torch raises a shape-mismatch error in inputs_merger (:1564) for this
                raise ValueError(f"{n_tok} <image> tokens in input_ids but {n_rows} image hidden states "
                                 f"({I} images x {cfg.perceiver_config.resampler_n_latents} latents)")
            if bool((am_cpu[is_img] == 0).any()):
                raise NotImplementedError("attention_mask == 0 on an <image> token")
        # inputs_merger (:1545-1565) = the packing plan with ONE slot per <image> token: slot r takes image-hidden row r in row-major
        # order; labels equal to image_token_id are the ignored ones (:1898); positions are Mistral's default arange
        plan = K.pack_plan(ids_d, attn_d, lab_d, 1, n_rows, IMG if img is not None else -(2 ** 62), -1, IMG, T)
        plan.position_ids = torch.arange(T, device=dev, dtype=torch.int64)[None].expand(B, T).contiguous()
        kstart = qend = None
        if segment_ids is not None:
            # long-sequence packing (BASELINE configs[3]; /root/reference/mantis/train/data.py:1546-1671): several samples in one row,
            # block-diagonal attention through O(L) segment bounds, positions restart per sample (:1641-1648: arange per item),
            # no prediction across a sample boundary
            seg_d = segment_ids.to(dev, non_blocking=True).to(torch.int32).contiguous()
            K.pack_segments(plan, ids_d, seg_d, -(2 ** 62))
            kstart, qend = plan.kstart, plan.qend
            plan.position_ids = (torch.arange(T, device=dev, dtype=torch.int64)[None] - kstart.to(torch.int64)).contiguous()
        D.compact_ce_rows(plan, ids_cpu, am_cpu, labels, IMG, IMG, dev, vocab_size=tc.vocab_size,      # labels equal to image_token_id are
                          refuse_image_targets=True)                                                  # the ignored ones
        x = K.pack_rows_fwd(plan, ids_d, m.lm["embed"], img)
        if record is not None and img is not None:
            record["merged_embeds"] = x.view(B, T, -1)
        kmask = None if D.no_padding(am_cpu, 0, T) else plan.kmask      # no pad position in the batch: no key mask (host-side decision)
        x, dctx = D8.forward(K, self, m.lm, tc, x, B, T, plan.position_ids, kmask, kstart, compute_grads, record,
                             checkpoint=m.is_gradient_checkpointing)
        loss, count, logits_full, hctx = D.head_and_loss(K, m.lm, tc, x, plan, B, T, labels is not None, grad_scale, loss_scale,
                                                         compute_grads, need_logits, record)
        if count is not None and not self._verified:
            if int(count[1]) != 0:
                raise IndexError(f"{int(count[1])} label(s) are >= vocab_size {tc.vocab_size}")
            self._verified = True
        out = dict(loss=loss, logits=logits_full, plan=plan)
        if not compute_grads:
            return out
        acc = not overwrite_grads
        g = m.grads
        dx = D8.backward(K, self, m.lm, g, m.grads_layers, tc, dctx, hctx, plan, B, T, kmask, kstart, qend, acc, on_bucket_ready)
        if g.get("embed") is not None:
            if overwrite_grads:
                g["embed"].zero_()
            K.embed_grad(dx, ids_d, plan, g["embed"], True)
        if img is not None:
            self.connector_backward(K.gather_rows(dx, plan.img_slot), cctx, acc)
        elif overwrite_grads:
            for n, p in m.named_parameters():          # no image in this batch: the connector receives exactly zero gradient
                if n.startswith("model.connector.") and p.grad is not None:
                    p.grad.zero_()
        if on_bucket_ready is not None:
            on_bucket_ready("front")
        return out
