#!/usr/bin/env python3
"""How far is the REFERENCE'S OWN bf16 run from its fp32 run?  (round-5 verdict, item 5)

The reference trains in bf16 (`--bf16 True`, /root/reference/mantis/train/scripts/train_mllava.sh:148; models loaded with
torch_dtype=bfloat16, train_mllava.py:132) while the parity oracle is fp32 on bf16-rounded weights.  SURVEY 8c proposes per-tensor gradient
bars of cosine >= 0.999 / rel-L2 <= 2e-2 for "bf16 product vs fp32 oracle"; a few tensors of the product sit above 2e-2 (the Qwen2-VL key
bias, the q / k projections and the Idefics2 connector at full width).  Whether that is the product's error or simply what bf16 arithmetic
does to those tensors can only be said with the reference's own bf16 deviation beside it.  This script records it:

  part A (tiny golden cases, the reference's OWN classes): transformers' Qwen2VLForConditionalGeneration -- the class
         /root/reference/mantis/models/qwen2_vl/modeling_qwen2_vl.py:1 resolves to -- on the four golden inputs, weights rounded to bf16:
         once in fp32, once with the module in bf16 (weights + activations, as the reference runs it); per trainable tensor
         cosine / rel-L2 of the bf16 gradient against the fp32 gradient.
  part B (full-width geometries of the GPU checks `*_full_width_vs_oracle`, depth 2 - 3): the oracle classes (LlavaRef / Idefics2Ref /
         Qwen2VLRef, pinned to the reference on the goldens) with dtype=float32 and dtype=bfloat16 on the same seeded weights
         (same init distribution and the same synthetic batch as the GPU checks; the weight VALUES differ from the GPU checks', whose
         generator lives on the device -- the envelope is a property of geometry and distribution, not of one draw).

Output: tests/golden/bf16_envelope.json = {case: {tensor: [cosine, rel_l2]}} (+ loss pairs).  The product's bar per tensor becomes
max(2e-2, 1.5 x envelope rel-L2) and min(0.999, 1 - 1.5 x (1 - envelope cosine)) (tests/helpers.py:envelope_bars).

Runs only in the build container (transformers + the CPU time: ~1 h on 8 cores for part B).  Usage:
    python tests/golden/make_bf16_envelope.py [A] [llava] [idefics2] [qwen2vl]      (no arguments: everything; results are merged into the JSON)
"""
import json
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
OUT = os.path.join(HERE, "bf16_envelope.json")


def cos_rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    if na == 0 or nb == 0:
        return (1.0 if na == nb else 0.0), (0.0 if na == nb else 1.0)
    return float(a @ b / (na * nb)), float(np.linalg.norm(a - b) / nb)


def merge(results):
    cur = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            cur = json.load(f)
    cur.update(results)
    cur["__doc__"] = ("per case and trainable tensor: [cosine, rel-L2] of the reference's bf16 gradient against its fp32 gradient on the same "
                      "bf16-rounded weights; written by tests/golden/make_bf16_envelope.py (see its docstring)")
    with open(OUT, "w") as f:
        json.dump(cur, f, indent=0, sort_keys=True)
        f.write("\n")


# ------------------------------------------------------------------------------------------------------------------- part A
def part_a():
    import make_golden_qwen2vl as G
    res = {}
    model = G.build(53)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())             # the parity tests round the weights first, too
    import copy
    model16 = copy.deepcopy(model).to(torch.bfloat16)
    for name in ("qwen2vl_b1_img2", "qwen2vl_b1_img1_tall", "qwen2vl_b2_rightpad", "qwen2vl_b1_text_only"):
        z = np.load(os.path.join(HERE, name + ".npz"))
        grads = {}
        losses = {}
        for tag, m, dt in (("fp32", model, torch.float32), ("bf16", model16, torch.bfloat16)):
            kw = {}
            if "pixel_values" in z.files:
                kw = dict(pixel_values=torch.from_numpy(z["pixel_values"]).to(dt), image_grid_thw=torch.from_numpy(z["image_grid_thw"]))
            m.zero_grad(set_to_none=True)
            m.model.rope_deltas = None
            r = m(input_ids=torch.from_numpy(z["input_ids"]), attention_mask=torch.from_numpy(z["attention_mask"]),
                  labels=torch.from_numpy(z["labels"]), mm_token_type_ids=torch.from_numpy(z["mm_token_type_ids"]), use_cache=False, **kw)
            r.loss.backward()
            losses[tag] = float(r.loss)
            grads[tag] = {n: p.grad.detach().float().numpy().copy() for n, p in m.named_parameters() if p.grad is not None}
        rep = {n: list(cos_rel(grads["bf16"][n], grads["fp32"][n])) for n in grads["fp32"]}
        rep["__loss__"] = [losses["bf16"], losses["fp32"]]
        res["reference_bf16:" + name] = rep
        worst = max(rep[n][1] for n in rep if not n.startswith("__"))
        wn = max((n for n in rep if not n.startswith("__")), key=lambda n: rep[n][1])
        print(f"A {name}: loss bf16 {losses['bf16']:.5f} fp32 {losses['fp32']:.5f}; worst rel-L2 {worst:.4f} ({wn}: cos {rep[wn][0]:.5f})", flush=True)
    # The tensors in question are tiny (the key bias of the golden geometry has 16 elements) and their gradient is a near-cancelling sum:
    # rel-L2 on one input is a noisy statistic.  So the same measurement on 24 more random inputs of the golden generator's family (one or two
    # images of random patch grids inside 24 - 40 tokens): per tensor the WORST cosine / rel-L2 the reference's bf16 run shows.
    rng = np.random.default_rng(2024)
    worst = {}
    for draw in range(24):
        T = int(rng.integers(24, 41))
        n_img = int(rng.integers(1, 3))
        grids, spans, pos = [], [], 2
        for _ in range(n_img):
            gh, gw = int(rng.choice([2, 4, 6, 8])), int(rng.choice([2, 4, 6]))
            n_tok = gh * gw // 4
            if pos + n_tok + 2 >= T - 2:
                break
            grids.append((1, gh, gw))
            spans.append((pos, n_tok))
            pos += n_tok + 2 + int(rng.integers(1, 4))
        ids = rng.integers(1, 300, size=T, dtype=np.int64)
        for s_, n in spans:
            ids[s_] = G.VSTART
            ids[s_ + 1: s_ + 1 + n] = G.IMG
            ids[s_ + 1 + n] = G.VEND
        mask = np.ones(T, np.int64)
        lab = ids.copy()
        lab[:6] = -100
        lab[ids == G.IMG] = -100
        g = np.array(grids, np.int64)
        pix = rng.standard_normal((int(sum(t * h * w for t, h, w in grids)), 3 * 2 * 14 * 14)).astype(np.float32)
        mmtt = (ids == G.IMG).astype(np.int32)
        grads = {}
        for tag, m, dt in (("fp32", model, torch.float32), ("bf16", model16, torch.bfloat16)):
            m.zero_grad(set_to_none=True)
            m.model.rope_deltas = None
            r = m(input_ids=torch.from_numpy(ids[None]), attention_mask=torch.from_numpy(mask[None]), labels=torch.from_numpy(lab[None]),
                  mm_token_type_ids=torch.from_numpy(mmtt[None]), use_cache=False, pixel_values=torch.from_numpy(pix).to(dt),
                  image_grid_thw=torch.from_numpy(g))
            r.loss.backward()
            grads[tag] = {n: p.grad.detach().float().numpy().copy() for n, p in m.named_parameters() if p.grad is not None}
        for n in grads["fp32"]:
            c, rl = cos_rel(grads["bf16"][n], grads["fp32"][n])
            w0 = worst.get(n, [1.0, 0.0])
            worst[n] = [min(w0[0], c), max(w0[1], rl)]
    res["reference_bf16:qwen2vl_tiny_random24_worst"] = worst
    wn = max(worst, key=lambda n: worst[n][1])
    print(f"A 24 random inputs: worst rel-L2 {worst[wn][1]:.4f} / cos {worst[wn][0]:.5f} ({wn}); k_proj.bias layer 1: "
          f"{worst['model.language_model.layers.1.self_attn.k_proj.bias']}", flush=True)
    merge(res)


# ------------------------------------------------------------------------------------------------------------------- part B
def oracle_pair(make_oracle, run):
    """gradients of the oracle in fp32 and in bf16 on the same weights -> {tensor: [cos, rel]}"""
    out = {}
    for tag, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        t0 = time.time()
        o = make_oracle(dt)
        o.zero_grad()
        loss = run(o, dt)
        loss.backward()
        out[tag] = ({n: w.grad.detach().float().numpy().copy() for n, w in o.w.items() if getattr(w, "grad", None) is not None}, float(loss))
        print(f"    {tag}: loss {float(loss):.5f} in {time.time() - t0:.0f} s", flush=True)
        del o
    g32, g16 = out["fp32"][0], out["bf16"][0]
    rep = {n: list(cos_rel(g16[n], g32[n])) for n in g32 if n in g16}
    rep["__loss__"] = [out["bf16"][1], out["fp32"][1]]
    return rep


def part_b_llava():
    from mantis_amd import configuration_llava as C
    from mantis_amd.modeling_llava import LlavaForConditionalGeneration
    from oracle.llava_ref import LlavaRef
    import bench
    cfg = C.mantis_8b_siglip_llama3()
    cfg.vision_config.num_hidden_layers = 3
    cfg.text_config.num_hidden_layers = 2
    model = LlavaForConditionalGeneration(cfg, device="cpu", seed=0)
    meta = dict(vision=cfg.vision_config.to_dict(), text=cfg.text_config.to_dict(), image_token_index=cfg.image_token_index,
                pad_token_id=cfg.pad_token_id, vision_feature_select_strategy=cfg.vision_feature_select_strategy,
                vision_feature_layer=cfg.vision_feature_layer, projector_hidden_act=cfg.projector_hidden_act)
    w = {n: p.detach().float().clone() for n, p in model.named_parameters()}
    train = {n for n, p in model.named_parameters() if p.requires_grad}
    del model
    b1 = bench.synthetic_batch(cfg, 2, 512, 4, cfg.vision_config.image_size, 0, 1)

    def run(o, dt):
        loss, _ = o.forward(b1["input_ids"].numpy(), [p.to(dt) for p in b1["pixel_values"]], b1["attention_mask"].numpy(), b1["labels"].numpy())
        return loss
    rep = oracle_pair(lambda dt: LlavaRef(w, meta, dtype=dt), run)
    rep = {n: v for n, v in rep.items() if n in train or n.startswith("__")}
    merge({"oracle_bf16:llava_full_width": rep})
    return rep


def part_b_idefics2():
    from mantis_amd import configuration_idefics2 as C
    from mantis_amd.modeling_idefics2 import Idefics2ForConditionalGeneration
    from oracle.idefics2_ref import Idefics2Ref
    import bench
    cfg = C.mantis_8b_idefics2()
    cfg.vision_config.num_hidden_layers = 2
    cfg.text_config.num_hidden_layers = 2
    cfg.perceiver_config.resampler_depth = 2
    model = Idefics2ForConditionalGeneration(cfg, device="cpu", seed=0)
    meta = dict(vision=cfg.vision_config.to_dict(), perceiver=cfg.perceiver_config.to_dict(), text=cfg.text_config.to_dict(),
                image_token_id=cfg.image_token_id)
    w = {n: p.detach().float().clone() for n, p in model.named_parameters()}
    train = {n for n, p in model.named_parameters() if p.requires_grad}
    del model
    batch = bench.synthetic_batch_idefics2(cfg, 1, 512, 2, 448, 0)

    def run(o, dt):
        loss, _ = o.forward(batch["input_ids"].numpy(), batch["pixel_values"].numpy(), None, batch["attention_mask"].numpy(), batch["labels"].numpy())
        return loss
    rep = oracle_pair(lambda dt: Idefics2Ref(w, meta, dtype=dt), run)
    rep = {n: v for n, v in rep.items() if n in train or n.startswith("__")}
    merge({"oracle_bf16:idefics2_full_width": rep})
    return rep


def part_b_qwen2vl():
    from mantis_amd import configuration_qwen2_vl as C
    from mantis_amd.modeling_qwen2_vl import Qwen2VLForConditionalGeneration
    from oracle.qwen2vl_ref import Qwen2VLRef
    import bench
    cfg = C.qwen2_vl_7b()
    cfg.vision_config.depth = 2
    cfg.text_config.num_hidden_layers = 2
    model = Qwen2VLForConditionalGeneration(cfg, device="cpu", seed=0)
    meta = dict(vision=cfg.vision_config.to_dict(), text=cfg.text_config.to_dict(), image_token_id=cfg.image_token_id)
    w = {n: p.detach().float().clone() for n, p in model.named_parameters()}
    train = {n for n, p in model.named_parameters() if p.requires_grad}
    del model
    batch = bench.synthetic_batch_qwen2vl(cfg, 1, 512, [(1, 16, 20)], 0)

    def run(o, dt):
        loss, _ = o.forward(batch["input_ids"].numpy(), batch["pixel_values"].numpy(), batch["image_grid_thw"].numpy(),
                            batch["attention_mask"].numpy(), batch["labels"].numpy())
        return loss
    rep = oracle_pair(lambda dt: Qwen2VLRef(w, meta, dtype=dt), run)
    rep = {n: v for n, v in rep.items() if n in train or n.startswith("__")}
    merge({"oracle_bf16:qwen2vl_full_width": rep})
    return rep


def main():
    what = sys.argv[1:] or ["A", "qwen2vl", "idefics2", "llava"]
    torch.manual_seed(0)
    for w in what:
        t0 = time.time()
        print(f"== {w}", flush=True)
        rep = {"A": part_a, "llava": part_b_llava, "idefics2": part_b_idefics2, "qwen2vl": part_b_qwen2vl}[w]()
        if rep:
            names = [n for n in rep if not n.startswith("__")]
            wn = max(names, key=lambda n: rep[n][1])
            print(f"   {w}: {len(names)} tensors, worst rel-L2 {rep[wn][1]:.4f} / cos {rep[wn][0]:.5f} ({wn}); {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
