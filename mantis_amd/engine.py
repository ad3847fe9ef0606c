"""Explicit forward + backward of the Mantis LLaVA training step over the HIP operator API (`hip_ops`).

This is the host side of rows B..J of SURVEY.md section 8a: it sequences the gfx950 kernels, owns which activations are
kept for backward (and which are recomputed), and writes gradients straight into the flat gradient arena -- there is no
autograd graph.  Reference control flow being reproduced:
  LlavaForConditionalGeneration.forward        /root/reference/mantis/models/mllava/modeling_llava.py:364-549
  HF LlamaModel / LlamaDecoderLayer forward    transformers/models/llama/modeling_llama.py:284-325,367-418
  HF Siglip/CLIP vision encoder forward        transformers/models/siglip/modeling_siglip.py:250-358
  accelerator.backward(loss)                   transformers/trainer.py:1961  (autograd of all of the above)

Work the reference does that this engine does not (results identical): the last ViT layer + post_layernorm + pooling head
(dead), the [B,L,V] logits tensor (only rows that can carry a label reach lm_head), 27 kept ViT hidden states.
"""
import os

import numpy as np
import torch

from . import decoder as D
from . import decoder_fp8 as D8
from . import hip_ops as K   # the ONLY compute backend; tests may monkeypatch `engine.K` with the oracle to test host logic
from .launch import launch_context


_DEBUG_SYNC = os.environ.get("MANTIS_DEBUG_SYNC") == "1"


class PackCountError(ValueError):
    pass


_WARNED_UNEQUAL = False


def _warn_unequal_counts_once():
    """a right-padded batch whose samples hold different numbers of images: the reference's slot search (modeling_llava.py:343-345)
    puts the image rows of the shorter samples into their trailing padding; reproduced bit for bit unless the flag is set"""
    global _WARNED_UNEQUAL
    if _WARNED_UNEQUAL:
        return
    _WARNED_UNEQUAL = True
    import warnings
    warnings.warn("right-padded batch with unequal image counts: image features are placed as the reference places them, i.e. "
                  "MIS-placed for the samples with fewer images (modeling_llava.py:343-345; the reference only ever runs bs = 1). "
                  "Set LlavaConfig(fix_unequal_counts=True) for the per-sample-correct placement.", stacklevel=3)


class LlavaEngine:
    def __init__(self, model):
        self.m = model
        self.cfg = model.config
        # device-side status words (pack plan status, out-of-range CE targets) are read back -- one host sync -- on the first
        # step of an engine and on every step under MANTIS_DEBUG_SYNC=1; the hot loop otherwise never syncs.
        self._verified = False
        self.w8 = None                       # decoder_fp8.Fp8Weights after model.set_precision("fp8")
        self.weights_unchanged = False       # set by MantisHipTrainer on the 2nd.. micro-batch of an accumulation window

    def _verify_device_status(self, plan, count):
        st = plan.status.cpu().tolist()
        if st[0] == 2:
            raise RuntimeError(f"pack_plan: host and device disagree on the merged length (host L={plan.L}); plan left in the safe "
                               f"all-padding state")
        if st[0] == 1:
            raise PackCountError(
                f"The input provided to the model are wrong. The number of image tokens is {st[2]} while the number of image "
                f"given to the model is {plan.I} ({st[1]} image slots found for {plan.I * plan.N} feature rows). This prevents "
                f"correct indexing and breaks batch generation.")
        if count is not None and int(count[1]) != 0:
            raise IndexError(f"{int(count[1])} label(s) are >= vocab_size {self.cfg.text_config.vocab_size} "
                             f"(torch.nn.CrossEntropyLoss: 'Target out of bounds')")
        self._verified = True

    def step_from_batch(self, inputs, launch=None, **kw):
        """The batch dict of the Mantis collator -> step().  launch: the caller's `launch.LaunchContext` (per-launch timers, the optimizer's
        sum-of-squares collector, the GEMM CU budget), in force for exactly this call."""
        with launch_context(launch):
            return self.step(inputs["input_ids"], inputs["attention_mask"], inputs.get("labels"), inputs.get("pixel_values"), **kw)

    # ------------------------------------------------------------------------------------------------ vision tower (frozen)
    def vision_forward(self, pixels, record=None):
        """pixels fp32 [I,3,H,W] on device -> selected image features [I*N', d_v] (N' excludes CLS under "default")."""
        m, vc = self.m, self.cfg.vision_config
        vt = m.vt
        I = pixels.shape[0]
        P, dv = vc.patch_size, vc.hidden_size
        N = (pixels.shape[2] // P) * (pixels.shape[3] // P)
        patches = K.im2col(pixels, P, vt["patch_kp"])
        pe = K.gemm_nt(patches, vt["patch_w_padded"](), bias=vt["patch_b"])
        x = K.vit_assemble(pe, vt["pos"], vt["cls"], I, N)
        NT = N + (1 if vt["cls"] is not None else 0)
        eps = vc.layer_norm_eps
        if vt["pre_ln"] is not None:
            x = K.layernorm_fwd(x, vt["pre_ln"][0], vt["pre_ln"][1], eps)
        nh = vc.num_attention_heads
        hd = dv // nh
        n_layers = vc.num_hidden_layers + self.cfg.vision_feature_layer + 1   # hidden_states[-2] = output of layer N-2
        for i in range(n_layers):
            lw = vt["layers"][i]
            y = K.layernorm_fwd(x, lw["ln1_w"], lw["ln1_b"], eps)
            qkv = K.gemm_nt(y, lw["qkv_w"], bias=lw["qkv_b"])
            o, _ = K.attn_fwd(qkv, I, NT, nh, nh, hd, None, hd ** -0.5, False, want_lse=False)
            x = K.gemm_nt(o, lw["out_w"], bias=lw["out_b"], residual=x)
            y = K.layernorm_fwd(x, lw["ln2_w"], lw["ln2_b"], eps)
            hmid = K.gemm_nt(y, lw["fc1_w"], bias=lw["fc1_b"], act=vc.hidden_act)
            x = K.gemm_nt(hmid, lw["fc2_w"], bias=lw["fc2_b"], residual=x)
        strat = self.cfg.vision_feature_select_strategy
        if strat == "default":          # modeling_llava.py:460-461
            if vt["cls"] is None:
                x = K.drop_cls(x, I, N - 1)
                N = N - 1
            else:
                x = K.drop_cls(x, I, N)
        elif strat != "full":
            raise ValueError(f"Unexpected select feature strategy: {strat}")
        return x, N

    # ------------------------------------------------------------------------------------------------ software pipelining of the tower
    def _pixels_to_device(self, pixel_values, dev):
        if isinstance(pixel_values, (list, tuple)):        # modeling_llava.py:431-432 (torch.cat of the per-sample list)
            # H2D per sample, concatenation on the device: a host-side torch.cat of a few MB costs ~30 ms on a 128-core host
            # (thread-pool start-up) -- more than the whole tiny-config step
            parts = [p.to(dev, non_blocking=True) for p in pixel_values if p is not None]
            pixel_values = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
        return pixel_values.to(dev, non_blocking=True).to(torch.float32).contiguous()

    def prefetch_vision(self, inputs, after_event=None, stream=None, launch=None):
        """Enqueue the frozen tower's forward of a FUTURE batch on `stream` (default: a plain side stream) behind `after_event` of the
        compute stream.  The tower is frozen, so its output does not depend on the optimizer step in between: with `stream` confined to
        a share of the compute units (hip_ops.cu_masked_stream) and the optimizer pass on the complementary share, the MFMA-bound tower of
        batch i+1 runs beside the HBM-bound clip + AdamW of step i.  The features are consumed by the step that is handed the SAME
        pixel_values object; any other batch is computed in line."""
        pv = inputs.get("pixel_values")
        if pv is None or inputs["input_ids"].shape[1] == 1 or not torch.cuda.is_available():
            return
        dev = self.m.device
        if stream is None:
            stream = getattr(self, "_side", None)
            if stream is None:
                stream = self._side = torch.cuda.Stream(device=dev)
        if after_event is not None:
            stream.wait_event(after_event)
        with torch.cuda.stream(stream), launch_context(launch):
            feats, N = self.vision_forward(self._pixels_to_device(pv, dev))
            done = torch.cuda.Event()
            done.record(stream)
        # one slot per in-flight batch: with the early mode (MantisHipTrainer.prefetch_early) the tower of batch i+1 is enqueued before
        # step i has picked up its own features
        if not isinstance(getattr(self, "_prefetched", None), dict):
            self._prefetched = {}
        self._prefetched[id(pv)] = (pv, feats, N, done)
        while len(self._prefetched) > 2:                   # a prefetched batch that never arrives must not pile up
            self._prefetched.pop(next(iter(self._prefetched)))

    # ------------------------------------------------------------------------------------------------ full step
    def step(self, input_ids, attention_mask, labels, pixel_values, grad_scale=1.0, loss_scale=1.0, compute_grads=True,
             overwrite_grads=True, need_logits=False, record=None, on_bucket_ready=None, segment_ids=None):
        """One forward (+ backward).  Returns dict(loss=fp32[1] on device, logits=[B,L,V] or None, plan=...).

        segment_ids (int [B,T], optional): sample packing (/root/reference/mantis/train/data.py:1546-1671) -- several samples in
        one row; tokens attend inside their own sample only (the reference's block-diagonal 4-D mask, here O(L) segment bounds
        consumed by the attention kernels), position ids restart per sample, no token is predicted across a sample boundary.

        grad_scale multiplies d(loss) (1/GA for Trainer.training_step); loss_scale multiplies the returned loss.
        overwrite_grads: first micro-batch after zero_grad -> gradient kernels overwrite instead of accumulate."""
        m, cfg, tc = self.m, self.cfg, self.cfg.text_config
        dev = m.device
        ids_cpu = input_ids.detach().to("cpu") if input_ids.device.type != "cpu" else input_ids
        B, T = ids_cpu.shape
        ign = cfg.ignore_index
        pad_id = cfg.pad_token_id if cfg.pad_token_id is not None else -1
        ids_d = input_ids.to(dev, non_blocking=True)
        attn_d = attention_mask.to(dev, non_blocking=True).to(torch.int64)
        lab_d = None if labels is None else labels.to(dev, non_blocking=True).to(torch.int64)

        # ---- rows B..E: pixels -> ViT -> projector
        img = None
        saved_proj = None
        I = 0
        N = 1
        if pixel_values is not None and T != 1:
            slots = getattr(self, "_prefetched", None)
            pre = slots.pop(id(pixel_values), None) if isinstance(slots, dict) else None
            if pre is not None and pre[0] is pixel_values and record is None:
                feats, N = pre[1], pre[2]                       # computed ahead on another stream (prefetch_vision)
                torch.cuda.current_stream().wait_event(pre[3])
                feats.record_stream(torch.cuda.current_stream())
                I = feats.shape[0] // N
            else:
                pix = self._pixels_to_device(pixel_values, dev)
                I = pix.shape[0]
                feats, N = self.vision_forward(pix, record)
            if record is not None:
                record["projector_in"] = feats.view(I, N, -1)
            pw = m.proj
            h1 = K.gemm_nt(feats, pw["w1"], bias=pw["b1"])
            a1 = K.act_fwd(h1, cfg.projector_hidden_act)
            img = K.gemm_nt(a1, pw["w2"], bias=pw["b2"])
            saved_proj = (feats, h1, a1)
            if record is not None:
                record["projector_out"] = img.view(I, N, -1)
            # modeling_llava.py:347-351 -- the slot count equals (#<image> tokens)*N by construction, so the reference's
            # check reduces to #tokens == #images; evaluated on the host copy of input_ids (no device sync in the hot loop)
            n_tok = int((ids_cpu == cfg.image_token_index).sum())
            if n_tok != I:
                raise PackCountError(
                    f"The input provided to the model are wrong. The number of image tokens is {n_tok} while"
                    f" the number of image given to the model is {I}. This prevents correct indexing and breaks batch generation.")
            kmax = int((ids_cpu == cfg.image_token_index).sum(-1).max())
            L = kmax * (N - 1) + T
            grown = (ids_cpu == cfg.image_token_index).sum(-1) * (N - 1)      # rows each sample's placeholders add in the merge
            fix = bool(getattr(cfg, "fix_unequal_counts", False))
            if not fix and B > 1 and int(grown.min()) != int(grown.max()) and bool((ids_cpu[:, -1] == pad_id).any()):
                _warn_unequal_counts_once()
            plan = K.pack_plan(ids_d, attn_d, lab_d, N, I, cfg.image_token_index, pad_id, ign, L, fix_unequal_counts=fix)
        else:
            # text-only: the reference skips the merge; positions default to arange (HF LlamaModel)
            L = T
            plan = K.pack_plan(ids_d, attn_d, lab_d, 1, 0, -(2 ** 62), pad_id, ign, L)
            grown = 0
            plan.position_ids = torch.arange(T, device=dev, dtype=torch.int64)[None].expand(B, T).contiguous()
        kstart = qend = None
        if segment_ids is not None:
            if bool((ids_cpu[:, -1] == pad_id).any()):
                raise ValueError("packed rows (segment_ids) must not be right-padded: pack them to equal length or pass them one by one")
            seg_d = segment_ids.to(dev, non_blocking=True).to(torch.int32).contiguous()
            img_tok = cfg.image_token_index if (pixel_values is not None and T != 1) else -(2 ** 62)
            K.pack_segments(plan, ids_d, seg_d, img_tok)
            kstart, qend = plan.kstart, plan.qend
        am_cpu = attention_mask.detach().to("cpu") if attention_mask.device.type != "cpu" else attention_mask
        D.compact_ce_rows(plan, ids_cpu, am_cpu, labels, cfg.image_token_index if (pixel_values is not None and T != 1) else -(2 ** 62),
                          ign, dev, vocab_size=tc.vocab_size)
        emb_w = m.lm["embed"]
        x = K.pack_rows_fwd(plan, ids_d, emb_w, img)        # rows F+G fused: [B*L, d]
        if record is not None:
            record.update(merged_embeds=x.view(B, L, -1), merged_attention_mask=plan.attention_mask,
                          merged_labels=plan.labels, merged_position_ids=plan.position_ids)

        # ---- rows H, I: Llama decoder, final norm, lm_head, masked shifted CE (decoder.py, shared with the Idefics2 path)
        # a batch without a single pad position needs no key mask (decided on the host copy of the batch: no device sync): the attention
        # kernels then skip the per-tile mask words altogether
        kmask = None if D.no_padding(am_cpu, grown, L) else plan.kmask
        x, dctx = D8.forward(K, self, m.lm, tc, x, B, L, plan.position_ids, kmask, kstart, compute_grads, record,
                             checkpoint=m.is_gradient_checkpointing)
        loss, count, logits_full, hctx = D.head_and_loss(K, m.lm, tc, x, plan, B, L, labels is not None, grad_scale, loss_scale,
                                                         compute_grads, need_logits, record)
        if count is not None and (not self._verified or _DEBUG_SYNC):
            self._verify_device_status(plan, count)
        out = dict(loss=loss, logits=logits_full, plan=plan)
        if not compute_grads:
            return out

        # ================================================================================================ backward (row J)
        cb = getattr(self, "_on_backward_start", None)
        if cb is not None:               # MantisHipTrainer.prefetch_point == "backward": the next batch's tower is queued here
            self._on_backward_start = None
            cb()
        g = m.grads                      # dict name -> grad view (None if frozen)
        acc = not overwrite_grads

        def gw(key):
            return g.get(key)

        dx = D8.backward(K, self, m.lm, g, m.grads_layers, tc, dctx, hctx, plan, B, L, kmask, kstart, qend, acc, on_bucket_ready)

        # ---- rows G, F, E backward: merged-row grads -> embedding rows + image-feature rows -> projector
        if gw("embed") is not None:
            if overwrite_grads:
                gw("embed").zero_()
            K.embed_grad(dx, ids_d, plan, gw("embed"), True)
        if img is not None and (gw("w1") is not None or gw("w2") is not None):
            feats, h1, a1 = saved_proj
            dimg = K.gather_rows(dx, plan.img_slot)
            if gw("b2") is not None:
                K.colsum(dimg, gw("b2"), acc)
                K.linear_dw(dimg, a1, gw("w2"), acc)
            da1 = K.linear_dx(dimg, m.proj["w2"])
            dh1 = K.act_bwd(da1, h1, cfg.projector_hidden_act)
            if gw("b1") is not None:
                K.colsum(dh1, gw("b1"), acc)
                K.linear_dw(dh1, feats, gw("w1"), acc)
        elif overwrite_grads:
            for k in ("w1", "b1", "w2", "b2"):       # no image in this batch: projector receives exactly zero gradient
                if gw(k) is not None:
                    gw(k).zero_()
        if on_bucket_ready is not None:
            on_bucket_ready("front")
        return out
