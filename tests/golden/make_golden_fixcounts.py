#!/usr/bin/env python3
"""Golden fixture for the `fix_unequal_counts` placement: a batch with UNEQUAL image counts, right-
and left-padded, together with what the REFERENCE returns for each of its samples run alone at B = 1 (the only batch size
its collator admits: mantis/models/mllava/processing_llava.py:277-285), where the mis-placement of
mantis/models/mllava/modeling_llava.py:343-345 cannot occur.  The fixed placement must reproduce, sample by sample, these
B = 1 outputs.

Runs only in the build container (imports /root/reference through make_golden.py's shim); own RNG stream, so the fixtures of
make_golden.py are untouched.  Writes tests/golden/siglip_b2_unequal_fixed.npz.

Usage:  python tests/golden/make_golden_fixcounts.py
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G        # noqa: E402  (helpers + the reference import shim; never imported by the test-suite)


def run_b1(model, ids, mask, labels, pixels):
    """The reference at B = 1 on one unpadded sample: merged integers, logits, loss, gradients."""
    got = {}
    orig = model._merge_input_ids_with_image_features

    def merge(*a, **k):
        r = orig(*a, **k)
        got["merged_attention_mask"], got["merged_labels"], got["merged_position_ids"] = (x.detach().clone().numpy() for x in r[1:])
        return r
    model._merge_input_ids_with_image_features = merge
    model.zero_grad(set_to_none=True)
    res = model(input_ids=torch.from_numpy(ids[None]), pixel_values=[torch.from_numpy(pixels)],
                attention_mask=torch.from_numpy(mask[None]), labels=torch.from_numpy(labels[None]))
    res.loss.backward()
    del model._merge_input_ids_with_image_features
    got["loss"] = res.loss.detach().numpy()
    got["logits"] = res.logits.detach().numpy()
    for n, p in model.named_parameters():
        if p.grad is not None:
            got["grad." + n] = p.grad.detach().numpy().copy()
    return got


def main():
    rng = np.random.default_rng(4321)
    model, cfg = G.build("siglip", 11)         # the weights of weights_siglip.npz (same seed, same construction)
    T = 24
    ia, ma = G.make_ids(rng, T, [3, 10], 0)
    ib_full, _ = G.make_ids(rng, T, [2], 0)
    nb = T - 5                                  # sample b: 19 real tokens, one image
    ib = ib_full[:nb]
    la, lb = G.make_labels(ia, ma, 6), G.make_labels(ib, np.ones(nb, np.int64), 6)
    pa = rng.standard_normal((2, 3, 56, 56)).astype(np.float32)
    pb = rng.standard_normal((1, 3, 56, 56)).astype(np.float32)
    out = {}
    for tag, ids, mask, lab, px in (("s0", ia, ma, la, pa), ("s1", ib, np.ones(nb, np.int64), lb, pb)):
        r = run_b1(model, ids, mask, lab, px)
        out[f"{tag}.input_ids"], out[f"{tag}.labels"] = ids, lab
        for k, v in r.items():
            out[f"{tag}.{k}"] = v
        print(f"{tag}: B=1 reference loss {float(r['loss']):.6f} logits {r['logits'].shape}")
    pad = np.full(T - nb, G.PAD, np.int64)
    zeros, ign = np.zeros(T - nb, np.int64), np.full(T - nb, -100, np.int64)
    # the same two samples as ONE batch, right- and left-padded (what a bs > 1 collator hands over)
    out["right.input_ids"] = np.stack([ia, np.concatenate([ib, pad])])
    out["right.attention_mask"] = np.stack([ma, np.concatenate([np.ones(nb, np.int64), zeros])])
    out["right.labels"] = np.stack([la, np.concatenate([lb, ign])])
    out["left.input_ids"] = np.stack([ia, np.concatenate([pad, ib])])
    out["left.attention_mask"] = np.stack([ma, np.concatenate([zeros, np.ones(nb, np.int64)])])
    out["left.labels"] = np.stack([la, np.concatenate([ign, lb])])
    out["pixel_values"] = np.concatenate([pa, pb], 0)
    out["pixel_counts"] = np.array([2, 1], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "siglip_b2_unequal_fixed.npz"), **out)


if __name__ == "__main__":
    main()
