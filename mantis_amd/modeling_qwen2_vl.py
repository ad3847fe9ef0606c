"""Drop-in module for the reference's `Qwen2VLForConditionalGeneration` (/root/reference/mantis/models/qwen2_vl/modeling_qwen2_vl.py:1
star-imports HF's class, transformers/models/qwen2_vl/modeling_qwen2_vl.py:1207-1338 over Qwen2VLModel :848-1204; driven by
/root/reference/mantis/train/train_qwen2_vl.py:126-128,209-212): same forward keyword arguments, same output fields, same parameter
names (state_dict-compatible with transformers 5.x; 4.x checkpoints are renamed on load), on the flat-arena layout of `ArenaModule` and
the hand-written gfx950 kernels (no autograd graph): dynamic-resolution ViT with 2-D rotary embedding and per-image attention + 2x2
patch merger (`visual`, frozen as the reference's training script keeps it), `<|image_pad|>` merge, Qwen2 decoder with q/k/v bias and
multimodal RoPE, fp32 cross-entropy.

Sample packing (several samples in one row, `segment_ids`): block-diagonal attention through segment bounds, the 3-D rope index
restarting per sample.  Documented divergences: video inputs, generation / KV-cache and `rope_deltas` bookkeeping are out of scope
(SURVEY.md section 2);
labels at positions with attention_mask == 0 must be -100 (what the reference's collator produces) -- checked on the first step."""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import decoder as D
from . import decoder_fp8 as D8
from . import hip_ops as K   # tests may monkeypatch `modeling_qwen2_vl.K` with the oracle's operators to test the host logic
from .arena import ArenaModule
from .launch import launch_context
from .configuration_qwen2_vl import Qwen2VLConfig


@dataclass
class Qwen2VLCausalLMOutputWithPast:
    """transformers/models/qwen2_vl/modeling_qwen2_vl.py:84-93"""
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[Tuple] = None
    hidden_states: Optional[Tuple] = None
    attentions: Optional[Tuple] = None
    rope_deltas: Optional[torch.Tensor] = None

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions, self.rope_deltas)
                     if v is not None)[k]


def _param_specs(cfg: Qwen2VLConfig):
    """(name, shape) in ARENA ORDER: [visual (tower + merger) | embed | text layer 0..n-1 | final norm | lm_head]."""
    vc, tc = cfg.vision_config, cfg.text_config
    dv, P, C, tp, mg = vc.embed_dim, vc.patch_size, vc.in_channels, vc.temporal_patch_size, vc.spatial_merge_size
    iv = int(dv * vc.mlp_ratio)
    v = "model.visual."
    s = [(v + "patch_embed.proj.weight", (dv, C, tp, P, P))]
    for i in range(vc.depth):
        p = f"{v}blocks.{i}."
        s += [(p + "norm1.weight", (dv,)), (p + "norm1.bias", (dv,)), (p + "norm2.weight", (dv,)), (p + "norm2.bias", (dv,)),
              (p + "attn.qkv.weight", (3 * dv, dv)), (p + "attn.qkv.bias", (3 * dv,)), (p + "attn.proj.weight", (dv, dv)),
              (p + "attn.proj.bias", (dv,)), (p + "mlp.fc1.weight", (iv, dv)), (p + "mlp.fc1.bias", (iv,)),
              (p + "mlp.fc2.weight", (dv, iv)), (p + "mlp.fc2.bias", (dv,))]
    dm = dv * mg * mg
    d, it, V = tc.hidden_size, tc.intermediate_size, tc.vocab_size
    H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
    s += [(v + "merger.ln_q.weight", (dv,)), (v + "merger.ln_q.bias", (dv,)), (v + "merger.mlp.0.weight", (dm, dm)),
          (v + "merger.mlp.0.bias", (dm,)), (v + "merger.mlp.2.weight", (d, dm)), (v + "merger.mlp.2.bias", (d,))]
    t = "model.language_model."
    s.append((t + "embed_tokens.weight", (V, d)))
    for i in range(tc.num_hidden_layers):
        p = f"{t}layers.{i}."
        # [norms | q k v weights | q k v biases | o] [gate up] [down] = reverse of backward completion (DP sub-buckets)
        s += [(p + "input_layernorm.weight", (d,)), (p + "post_attention_layernorm.weight", (d,)),
              (p + "self_attn.q_proj.weight", (H * hd, d)), (p + "self_attn.k_proj.weight", (Hkv * hd, d)),
              (p + "self_attn.v_proj.weight", (Hkv * hd, d)),
              (p + "self_attn.q_proj.bias", (H * hd,)), (p + "self_attn.k_proj.bias", (Hkv * hd,)), (p + "self_attn.v_proj.bias", (Hkv * hd,)),
              (p + "self_attn.o_proj.weight", (d, H * hd)),
              (p + "mlp.gate_proj.weight", (it, d)), (p + "mlp.up_proj.weight", (it, d)), (p + "mlp.down_proj.weight", (d, it))]
    s += [(t + "norm.weight", (d,)), ("lm_head.weight", (V, d))]
    return s


def _rename_hf4(k):
    """transformers 4.45-4.51 checkpoint names (what the reference was written against) -> 5.x names."""
    if k.startswith("visual."):
        return "model." + k
    if k.startswith("model.") and not k.startswith(("model.visual.", "model.language_model.")):
        return "model.language_model." + k[len("model."):]
    return k


class Qwen2VLForConditionalGeneration(ArenaModule):
    config_class = Qwen2VLConfig
    supports_gradient_checkpointing = True       # per decoder layer (decoder.decoder_forward(checkpoint=True)); the frozen tower saves nothing anyway
    frozen_prefixes = ("model.visual.",)                     # train_qwen2_vl.py:209-212

    def __init__(self, config: Qwen2VLConfig, device=None, dtype=torch.bfloat16, init="normal", seed=0):
        super().__init__()
        self.config = config
        self.image_token_id = config.image_token_id
        self.vocab_size = config.vocab_size
        self.precision = "bf16"
        self._init_arena(_param_specs(config), device, dtype)
        self.engine = Qwen2VLEngine(self)
        self._build_views()
        if init == "normal":
            self.reset_parameters(seed)
        self.train()

    def _build_views(self):
        vc, tc = self.config.vision_config, self.config.text_config
        g = lambda n: self._param(n).data
        dv = vc.embed_dim
        v = "model.visual."
        blocks = []
        for i in range(vc.depth):
            p = f"{v}blocks.{i}."
            blocks.append(dict(ln1=(g(p + "norm1.weight"), g(p + "norm1.bias")), ln2=(g(p + "norm2.weight"), g(p + "norm2.bias")),
                               qkv_w=g(p + "attn.qkv.weight"), qkv_b=g(p + "attn.qkv.bias"), proj_w=g(p + "attn.proj.weight"),
                               proj_b=g(p + "attn.proj.bias"), fc1_w=g(p + "mlp.fc1.weight"), fc1_b=g(p + "mlp.fc1.bias"),
                               fc2_w=g(p + "mlp.fc2.weight"), fc2_b=g(p + "mlp.fc2.bias")))
        self.vt = dict(patch_w=g(v + "patch_embed.proj.weight").view(dv, -1), blocks=blocks,
                       ln_q=(g(v + "merger.ln_q.weight"), g(v + "merger.ln_q.bias")),
                       m0_w=g(v + "merger.mlp.0.weight"), m0_b=g(v + "merger.mlp.0.bias"),
                       m2_w=g(v + "merger.mlp.2.weight"), m2_b=g(v + "merger.mlp.2.bias"))
        d, it = tc.hidden_size, tc.intermediate_size
        H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
        t = "model.language_model."
        ll = []
        for i in range(tc.num_hidden_layers):
            p = f"{t}layers.{i}."
            ll.append(dict(qkv=self._flat(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (H + 2 * Hkv) * hd, d),
                           qkv_b=self._flat(p + "self_attn.q_proj.bias", p + "self_attn.v_proj.bias", 1, (H + 2 * Hkv) * hd).view(-1),
                           o=g(p + "self_attn.o_proj.weight"),
                           gu=self._flat(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", 2 * it, d), down=g(p + "mlp.down_proj.weight"),
                           ln1=g(p + "input_layernorm.weight"), ln2=g(p + "post_attention_layernorm.weight")))
        self.lm = dict(embed=g(t + "embed_tokens.weight"), layers=ll, norm=g(t + "norm.weight"), head=g("lm_head.weight"))

    def _build_grad_views(self):
        tc = self.config.text_config
        gv = self._grad_views.get
        d, it = tc.hidden_size, tc.intermediate_size
        H, Hkv, hd = tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim
        t = "model.language_model."
        self.grads = dict(head=gv("lm_head.weight"), norm=gv(t + "norm.weight"), embed=gv(t + "embed_tokens.weight"))
        self.grads_layers = []
        for i in range(tc.num_hidden_layers):
            p = f"{t}layers.{i}."
            qb = self._gflat(p + "self_attn.q_proj.bias", p + "self_attn.v_proj.bias", 1, (H + 2 * Hkv) * hd)
            self.grads_layers.append(dict(
                qkv=self._gflat(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (H + 2 * Hkv) * hd, d),
                qkv_b=None if qb is None else qb.view(-1), o=gv(p + "self_attn.o_proj.weight"),
                gu=self._gflat(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", 2 * it, d), down=gv(p + "mlp.down_proj.weight"),
                ln1=gv(p + "input_layernorm.weight"), ln2=gv(p + "post_attention_layernorm.weight")))

    @torch.no_grad()
    def reset_parameters(self, seed=0):
        """normal(0, initializer_range), norm weights 1, biases 0 (HF PreTrainedModel._init_weights)."""
        std = self.config.text_config.get("initializer_range", 0.02)
        self._param_version += 1
        gen = torch.Generator(device=self.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if p.dim() == 1:
                p.fill_(1.0 if (("norm" in name or "ln_q" in name) and name.endswith("weight")) else 0.0)
            else:
                flat = p.view(-1)
                step = 1 << 26
                for a in range(0, flat.numel(), step):
                    b = min(flat.numel(), a + step)
                    flat[a:b] = torch.randn(b - a, generator=gen, device=self.device, dtype=torch.float32).mul_(std)

    def get_input_embeddings(self):
        return self.model.language_model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    def tie_weights(self, *a, **k):
        return None

    def load_reference_state_dict(self, sd, strict=True):
        return self.copy_state_dict(sd, rename=_rename_hf4, ignorable=lambda k: k.endswith("inv_freq"), strict=strict)

    def grad_buckets(self):
        """Contiguous gradient-arena slices in backward-completion order: 'head', per text layer n-1..0 ('layer', i, 'down' | 'gu' |
        'attn'), then 'front' = token embedding."""
        self._ensure_grad_arena()
        span = self._bucket_span
        out = {"head": span(lambda n: n.startswith("model.language_model.norm") or n.startswith("lm_head"))}
        for i in range(self.config.text_config.num_hidden_layers):
            p = f"model.language_model.layers.{i}."
            out[("layer", i, "down")] = span(lambda n: n == p + "mlp.down_proj.weight")
            out[("layer", i, "gu")] = span(lambda n: n in (p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"))
            out[("layer", i, "attn")] = span(lambda n: n.startswith(p) and ".mlp." not in n)
        out["front"] = span(lambda n: n.startswith("model.language_model.embed_tokens"))
        return {k: v for k, v in out.items() if v is not None}

    # ------------------------------------------------------------------ forward (HF Qwen2VLForConditionalGeneration.forward :1245-1338)
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, pixel_values=None, pixel_values_videos=None,
                image_grid_thw=None, video_grid_thw=None, rope_deltas=None, mm_token_type_ids=None, return_dict=None, return_logits=None,
                _record=None, **kwargs):
        if inputs_embeds is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("generation / KV-cache are out of scope: training forward only")
        if pixel_values_videos is not None or video_grid_thw is not None:
            raise NotImplementedError("video inputs are out of scope (SURVEY.md section 2)")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states are not produced by the fused kernels")
        if position_ids is not None:
            raise NotImplementedError("explicit position_ids (the reference passes None: the 3-D rope index is derived from the batch)")
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        want_grads = self.training and labels is not None and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if return_logits is None:
            return_logits = not want_grads        # training: the [B,L,V] logits are never materialised unless asked for
        if want_grads:
            # stock `loss.backward()` callers: the fused forward+backward runs now, .backward() publishes the gradients (arena.FusedStep)
            loss = self._autograd_step(lambda: self.engine.step(
                input_ids, attention_mask, labels, pixel_values, image_grid_thw, grad_scale=1.0, loss_scale=1.0, compute_grads=True,
                overwrite_grads=True, need_logits=return_logits, record=_record))
            lg = self._last_logits
        else:
            out = self.engine.step(input_ids, attention_mask, labels, pixel_values, image_grid_thw, compute_grads=False,
                                   need_logits=return_logits, record=_record)
            loss = None if labels is None else out["loss"].reshape(())
            lg = out["logits"]
        logits = None if lg is None else lg.float()
        if not return_dict:
            return ((loss,) if loss is not None else ()) + ((logits,) if logits is not None else ())
        return Qwen2VLCausalLMOutputWithPast(loss=loss, logits=logits)


# ---------------------------------------------------------------------- host-side integer bookkeeping (tiny, data dependent)
def vision_hw_ids(grids, merge):
    """Patch (row, column) coordinates in the tower's row order -- merge-window major, so 4 consecutive rows are one 2x2 merge group
    (transformers/vision_utils.py get_vision_position_ids).  -> int64 [2, N]."""
    hs, ws = [], []
    for t, h, w in grids:
        hp = torch.arange(h, dtype=torch.int64)[:, None].expand(h, w)
        wp = torch.arange(w, dtype=torch.int64)[None, :].expand(h, w)
        shp = (h // merge, merge, w // merge, merge)
        hs.append(hp.reshape(shp).transpose(1, 2).reshape(-1).repeat(t))
        ws.append(wp.reshape(shp).transpose(1, 2).reshape(-1).repeat(t))
    return torch.stack([torch.cat(hs), torch.cat(ws)])


def mrope_position_ids(ids, am, grids, image_token_id, merge, segment_ids=None):
    """Qwen2VLModel.get_rope_index (:914-1018), images only -> int64 [3, B, T].  Runs of text count on; an image (t, h, w) puts
    (frame, row, column) of its merged grid on top of the running position and advances it by max(h, w) / merge; positions where
    attention_mask == 0 stay 0 and do not advance the counter.  segment_ids [B, T] (packed samples): the counter restarts at every
    sample, i.e. each sample gets the index it would get alone (images are consumed in order of appearance)."""
    B, T = ids.shape
    pos = torch.zeros((3, B, T), dtype=torch.int64)
    gi = 0
    for b in range(B):
        keep = torch.nonzero(am[b] != 0).reshape(-1)
        row = ids[b][keep]
        n = row.numel()
        if n == 0:
            continue
        img = row == image_token_id
        cut = img[1:] != img[:-1]
        restart = torch.zeros(n, dtype=torch.bool)
        if segment_ids is not None:
            seg = segment_ids[b][keep]
            sb = seg[1:] != seg[:-1]
            restart[1:] = sb
            cut = cut | sb                     # a sample boundary also ends a run (two images of two samples may touch)
        edge = torch.nonzero(cut).reshape(-1) + 1
        starts = [0] + edge.tolist()
        ends = edge.tolist() + [n]
        cur = 0
        out = torch.empty((3, n), dtype=torch.int64)
        for s, e in zip(starts, ends):
            if bool(restart[s]):
                cur = 0
            if not bool(img[s]):
                out[:, s:e] = torch.arange(cur, cur + e - s, dtype=torch.int64)[None]
                cur += e - s
            else:
                if gi >= len(grids):
                    raise ValueError("more runs of image tokens than rows in image_grid_thw")
                t, h, w = grids[gi]
                gi += 1
                gh, gw = h // merge, w // merge
                if t * gh * gw != e - s:
                    raise ValueError(f"a run of {e - s} image tokens does not match its grid {t}x{h}x{w} ({t * gh * gw} merged patches)")
                out[0, s:e] = torch.arange(t, dtype=torch.int64).repeat_interleave(gh * gw) + cur
                out[1, s:e] = torch.arange(gh, dtype=torch.int64).repeat_interleave(gw).repeat(t) + cur
                out[2, s:e] = torch.arange(gw, dtype=torch.int64).repeat(t * gh) + cur
                cur += max(h, w) // merge
        pos[:, b, keep] = out
    return pos


class Qwen2VLEngine:
    """Host side of the Qwen2-VL step: sequences the gfx950 kernels for the dynamic-resolution vision tower + merger (forward only), the
    image-token merge and -- through decoder.py -- the Qwen2 decoder, head and loss."""

    def __init__(self, model):
        self.m = model
        self.cfg = model.config
        self._verified = False
        self.w8 = None                       # Fp8Weights when model.precision == "fp8"
        # set by MantisHipTrainer before every micro-batch: True on the 2nd.. micro-batch of an accumulation window (no optimizer step
        # since the last one), so the fp8 weight copies are reused; any other caller gets a fresh quantisation every step
        self.weights_unchanged = False
        self._side = None                    # side stream of prefetch_vision
        self._prefetched = {}                # id(pixel_values object) -> (pixel_values object, image rows, done event)

    # ------------------------------------------------------------------ software pipelining of the frozen tower
    def prefetch_vision(self, inputs, after_event=None, stream=None, launch=None):
        """Enqueue the tower + merger forward of a FUTURE batch on a side stream (behind `after_event` of the compute stream).  `visual`
        is frozen, so its output does not depend on the optimizer step in between: MantisHipTrainer calls this at the end of an
        accumulation window so the MFMA-bound tower of batch i+1 runs beside the HBM-bound clip + AdamW of step i.  The result is
        consumed by the step that is handed the SAME pixel_values object; any other batch is computed in line as before."""
        pv = inputs.get("pixel_values")
        if pv is None or inputs.get("image_grid_thw") is None or not torch.cuda.is_available():
            return
        dev = self.m.device
        if stream is None:
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            stream = self._side
        grids = [tuple(int(v) for v in g) for g in torch.as_tensor(inputs["image_grid_thw"]).tolist()]
        if after_event is not None:
            stream.wait_event(after_event)
        with torch.cuda.stream(stream), launch_context(launch):
            pix = torch.as_tensor(pv).to(dev, non_blocking=True).to(torch.float32).contiguous()
            img = self.vision_forward(pix, grids)
            done = torch.cuda.Event()
            done.record(stream)
        self._prefetched[id(pv)] = (pv, img, done)
        while len(self._prefetched) > 2:                   # a prefetched batch that never arrives must not pile up
            self._prefetched.pop(next(iter(self._prefetched)))

    def step_from_batch(self, inputs, launch=None, **kw):
        """launch: the caller's `launch.LaunchContext`, in force for exactly this call (see engine.LlavaEngine.step_from_batch)"""
        if kw.get("segment_ids") is None and inputs.get("segment_ids") is not None:
            kw["segment_ids"] = inputs["segment_ids"]
        with launch_context(launch):
            return self.step(inputs["input_ids"], inputs["attention_mask"], inputs.get("labels"), inputs.get("pixel_values"),
                             inputs.get("image_grid_thw"), **kw)

    # ------------------------------------------------------------------ vision tower + patch merger (frozen, forward only)
    def vision_forward(self, pix, grids, record=None):
        """pix fp32 [N, C*tp*P*P] on the device (the processor's flattened patches), grids = [(t, h, w)] -> [N / merge^2, d]."""
        m, vc = self.m, self.cfg.vision_config
        vt = m.vt
        dev = pix.device
        dv, nh, mg = vc.embed_dim, vc.num_heads, vc.spatial_merge_size
        hd = dv // nh
        N = pix.shape[0]
        if N != sum(t * h * w for t, h, w in grids):
            raise ValueError(f"pixel_values has {N} patch rows but image_grid_thw describes {sum(t * h * w for t, h, w in grids)}")
        x = K.gemm_nt(K.cast_pad_rows(pix, K.pad8(pix.shape[1])), vt["patch_w"], k=K.pad8(pix.shape[1]))
        if record is not None:
            record["vision_patch_embed"] = x
        # 2-D rotary embedding (:239-248, :709-713): first half of the rotary dim turns with the patch row, second with the column
        quarter = hd // 4
        f = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))
        inv = torch.cat([f, f]).to(dev)
        sec = torch.cat([torch.zeros(quarter, dtype=torch.int32), torch.ones(quarter, dtype=torch.int32)]).to(dev)
        cos, sin = K.rope_table_sections(vision_hw_ids(grids, mg).to(dev), inv, sec)
        # per-image attention (cu_seqlens, :404-418): consecutive images of equal size share one launch as a batch
        groups, r0 = [], 0
        for t, h, w in grids:
            n = t * h * w
            if groups and groups[-1][2] == n:
                groups[-1][1] += 1
            else:
                groups.append([r0, 1, n])
            r0 += n
        for bi, bw in enumerate(vt["blocks"]):
            y = K.layernorm_fwd(x, bw["ln1"][0], bw["ln1"][1], 1e-6)
            qkv = K.gemm_nt(y, bw["qkv_w"], bias=bw["qkv_b"])
            K.rope_apply_(qkv, cos, sin, 2 * nh, hd)
            o = torch.empty((N, dv), dtype=x.dtype, device=dev)
            for r, cnt, n in groups:
                rows = slice(r, r + cnt * n)
                K.attn_fwd_qkv(qkv[rows, :dv], qkv[rows, dv: 2 * dv], qkv[rows, 2 * dv:], cnt, n, nh, nh, hd, None, hd ** -0.5, False,
                               want_lse=False, out=o[rows])
            x = K.gemm_nt(o, bw["proj_w"], bias=bw["proj_b"], residual=x)
            y = K.layernorm_fwd(x, bw["ln2"][0], bw["ln2"][1], 1e-6)
            hmid = K.gemm_nt(y, bw["fc1_w"], bias=bw["fc1_b"], act=vc.hidden_act)
            x = K.gemm_nt(hmid, bw["fc2_w"], bias=bw["fc2_b"], residual=x)
            if record is not None:
                record[f"vision_block{bi}_out"] = x
        if record is not None:
            record["vision_last_hidden_state"] = x
        y = K.layernorm_fwd(x, vt["ln_q"][0], vt["ln_q"][1], 1e-6).view(N // (mg * mg), dv * mg * mg)     # :288-290
        y = K.gemm_nt(y, vt["m0_w"], bias=vt["m0_b"], act="gelu")
        return K.gemm_nt(y, vt["m2_w"], bias=vt["m2_b"])

    # ------------------------------------------------------------------ full step
    def step(self, input_ids, attention_mask, labels, pixel_values, image_grid_thw=None, grad_scale=1.0, loss_scale=1.0,
             compute_grads=True, overwrite_grads=True, need_logits=False, record=None, on_bucket_ready=None, segment_ids=None):
        m, cfg, tc = self.m, self.cfg, self.cfg.text_config
        dev = m.device
        ids_cpu = input_ids.detach().to("cpu") if input_ids.device.type != "cpu" else input_ids
        B, T = ids_cpu.shape
        IMG = cfg.image_token_id
        am_cpu = attention_mask.detach().to("cpu") if attention_mask.device.type != "cpu" else attention_mask
        if labels is not None:                  # host-only check, every step
            lab_cpu = labels.detach().to("cpu") if labels.device.type != "cpu" else labels
            if bool((lab_cpu[am_cpu == 0] != -100).any()):
                raise NotImplementedError("labels != -100 where attention_mask == 0 (HF's loss would count them; the reference's collator "
                                          "never produces them)")
        ids_d = input_ids.to(dev, non_blocking=True)
        attn_d = attention_mask.to(dev, non_blocking=True).to(torch.int64)
        lab_d = None if labels is None else labels.to(dev, non_blocking=True).to(torch.int64)
        img = None
        n_rows = 0
        mg = cfg.vision_config.spatial_merge_size
        if pixel_values is not None:
            if image_grid_thw is None:
                raise ValueError("pixel_values without image_grid_thw")
            grids = [tuple(int(v) for v in g) for g in torch.as_tensor(image_grid_thw).tolist()]
            pre = self._prefetched.pop(id(pixel_values), None)
            if pre is not None and pre[0] is pixel_values and record is None:
                img = pre[1]                                   # computed ahead on the side stream (prefetch_vision)
                torch.cuda.current_stream().wait_event(pre[2])
                img.record_stream(torch.cuda.current_stream())
            else:
                pix = torch.as_tensor(pixel_values).to(dev, non_blocking=True).to(torch.float32).contiguous()
                img = self.vision_forward(pix, grids, record)
            if record is not None:
                record["vision_merged"] = img
            n_rows = img.shape[0]
            n_tok = int((ids_cpu == IMG).sum())
            if n_tok != n_rows:       # get_placeholder_mask (:1078-1082)
                raise ValueError(f"Image features and image tokens do not match, tokens: {n_tok}, features: {n_rows}")
            if bool((am_cpu[ids_cpu == IMG] == 0).any()):
                raise NotImplementedError("attention_mask == 0 on an <|image_pad|> token")
            seg_cpu = None if segment_ids is None else (segment_ids.detach().to("cpu") if segment_ids.device.type != "cpu" else segment_ids)
            pos3 = mrope_position_ids(ids_cpu, am_cpu, grids, IMG, mg, seg_cpu)
        elif segment_ids is None:
            pos3 = torch.arange(T, dtype=torch.int64)[None, None].expand(3, B, T).contiguous()   # text only: HF counts 0..T-1 itself
        else:
            seg_cpu = segment_ids.detach().to("cpu") if segment_ids.device.type != "cpu" else segment_ids
            pos3 = mrope_position_ids(ids_cpu, torch.ones_like(ids_cpu), [], IMG, mg, seg_cpu)   # arange restarting at every sample
        # masked_scatter (:1160-1166) = the packing plan with ONE slot per <|image_pad|> token, rows taken in order
        plan = K.pack_plan(ids_d, attn_d, lab_d, 1, n_rows, IMG if img is not None else -(2 ** 62), -1, -100, T)
        kstart = qend = None
        if segment_ids is not None:
            # sample packing (/root/reference/mantis/train/data.py:1546-1671): several samples in one row, block-diagonal attention
            # through O(L) segment bounds, rope index restarting per sample (above), no prediction across a sample boundary
            # (the kernel also rewrites the plan's own 1-D position buffer, which this path replaces by the 3-D index right after)
            seg_d = segment_ids.to(dev, non_blocking=True).to(torch.int32).contiguous()
            K.pack_segments(plan, ids_d, seg_d, -(2 ** 62))
            kstart, qend = plan.kstart, plan.qend
        plan.position_ids = pos3
        D.compact_ce_rows(plan, ids_cpu, am_cpu, labels, IMG, -100, dev, vocab_size=tc.vocab_size, refuse_image_targets=True)
        x = K.pack_rows_fwd(plan, ids_d, m.lm["embed"], img)
        if record is not None:
            record["merged_embeds"] = x.view(B, T, -1)
            record["position_ids"] = pos3
        # multimodal RoPE (:156-170, :207-213): rotary frequency j reads the temporal / height / width id by section
        sec = torch.repeat_interleave(torch.arange(3, dtype=torch.int32), torch.tensor(tc.rope_parameters["mrope_section"])).to(dev)
        rope = K.rope_table_sections(pos3.reshape(3, B * T).to(dev), D.inv_freq(tc.head_dim, tc.rope_theta).to(dev), sec)
        kmask = None if D.no_padding(am_cpu, 0, T) else plan.kmask      # no pad position in the batch: no key mask (host-side decision)
        x, dctx = D8.forward(K, self, m.lm, tc, x, B, T, None, kmask, kstart, compute_grads, record, rope=rope,
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
        # `visual` (tower + merger) is frozen: the gradient on the image rows ends here
        if on_bucket_ready is not None:
            on_bucket_ready("front")
        return out
