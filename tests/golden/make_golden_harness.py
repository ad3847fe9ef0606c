#!/usr/bin/env python3
"""Golden fixtures for the Python harness around the hot path (SURVEY.md section 8a "Python callers" and row f4), produced by
RUNNING the reference's own code (/root/reference/mantis/train/data.py) on synthetic token ids:

  label_rule.npz      ChatDataset.getitem label masking, data.py:415-466 -- the real method, LLAMA_3 and PLAIN branches
  collate_ref.npz     Collator._right_pad_inputs_with_attention_mask, data.py:1392-1527 -- right-pad ids / mask / labels
  collate_qwen_ref.npz  the same method on Qwen2-VL-style items (flattened patches + image_grid_thw: concatenated along dim 0)
  pack_batch_ref.npz  PackingDataset.pack_batch, data.py:1609-1671 -- packed ids, block-diagonal 4-D mask, position ids, labels

`mantis.train.data` imports `av` and `decord` (video decoding, absent here and irrelevant to these code paths): empty
stand-in modules are registered in sys.modules before the import (SURVEY 8c recipe).  The tokenizer / processor / conversation
template objects the methods touch are minimal fakes that return the planted token ids; the label / padding / packing code that
runs is the reference's.  Runs only in the build container; the test-suite reads the .npz files only.

Usage: python tests/golden/make_golden_harness.py
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
IMG, SEP, PAD = 298, 290, 299


def _import_reference_data():
    for n in ("av", "decord"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.path.insert(0, REF)
    import mantis.train.data as D
    from mantis.train.conversation import SeparatorStyle
    return D, SeparatorStyle


class _Tok:
    pad_token_id = PAD

    def convert_tokens_to_ids(self, tok):
        return SEP


class _Encoding(dict):
    """BatchEncoding stand-in: a dict with .pop / `in` / item assignment, which is all getitem uses."""


class _Processor:
    def __init__(self):
        self.tokenizer = _Tok()
        self.next_ids = None

    def __call__(self, *a, **k):
        ids = torch.from_numpy(self.next_ids)[None]
        return _Encoding(input_ids=ids, attention_mask=torch.ones_like(ids))


class _Conv:
    def __init__(self, style, sep):
        self.sep_style, self.sep, self.sep2, self.system, self.sep_offset = style, sep, None, "", 0
        self.roles = ("user", "assistant")
        self.messages = []

    def get_prompt(self):
        return "synthetic"


def _dataset(D, style, sep):
    ds = object.__new__(D.ChatDataset)
    ds.processor = _Processor()
    ds.conv = _Conv(style, sep)
    ds.max_image_size, ds.image_dir, ds.max_seq_len = None, None, 4096
    ds.ensure_seq_len_multiple_of = None
    ds.packing_same_mm_media = False
    ds.data_path, ds.name, ds.split = "synthetic", "synthetic", "train"
    ds.conversations = [[["user", "q"], ["assistant", "a"]]]
    ds.all_images = [[]]
    return ds


def label_rule(D, Style):
    rng = np.random.default_rng(7)
    D.set_default_image_token_id(IMG)
    llama3 = _dataset(D, Style.LLAMA_3, "<|eot_id|>")
    plain = _dataset(D, Style.PLAIN, "\n")
    out = {"sep_id": np.array(SEP), "image_id": np.array(IMG)}
    specs = [(40, [5, 12, 20, 31]), (40, [3, 9, 15]), (24, [4]), (24, []), (30, [0, 10, 29]), (33, [1, 2, 3, 4, 32]), (17, [16, 8])]
    for ci, (T, seps) in enumerate(specs):
        ids = rng.integers(0, 280, size=T, dtype=np.int64)
        for s in seps:
            ids[s] = SEP
        if 1 not in seps:
            ids[1] = IMG
        ids[T // 2 + 1 if (T // 2 + 1) not in seps else T // 2 + 2] = IMG
        llama3.processor.next_ids = ids
        enc = llama3.getitem(0)
        assert torch.equal(enc["input_ids"][0], torch.from_numpy(ids))
        plain.processor.next_ids = ids
        encp = plain.getitem(0)
        out[f"c{ci}.ids"], out[f"c{ci}.llama3"], out[f"c{ci}.plain"] = ids, enc["labels"][0].numpy().copy(), encp["labels"][0].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "label_rule.npz"), **out)
    print("label_rule.npz:", len(specs), "cases via ChatDataset.getitem")


def _item(rng, T, n_img):
    ids = rng.integers(0, 280, size=T, dtype=np.int64)
    pos = np.sort(rng.choice(T - 1, size=n_img, replace=False))
    ids[pos] = IMG
    lab = np.where(rng.random(T) < 0.6, ids, -100)
    lab[ids == IMG] = -100
    return dict(input_ids=torch.from_numpy(ids)[None], attention_mask=torch.ones(1, T, dtype=torch.int64),
                labels=torch.from_numpy(lab)[None], pixel_values=torch.from_numpy(rng.standard_normal((n_img, 3, 4, 4)).astype(np.float32)))


def collate(D):
    rng = np.random.default_rng(11)
    col = D.Collator(processor=object())          # object(): no _right_pad_... attribute -> the generic implementation runs
    col.tokenizer = _Tok()
    items = [_item(rng, 19, 2), _item(rng, 11, 1), _item(rng, 25, 3)]
    res = col(items)
    out = {"n": np.array(len(items)), "pad_token_id": np.array(PAD)}
    for i, it in enumerate(items):
        for k in ("input_ids", "attention_mask", "labels", "pixel_values"):
            out[f"s{i}.{k}"] = it[k].numpy()
    for k in ("input_ids", "attention_mask", "labels", "pixel_values"):
        out[f"out.{k}"] = res[k].numpy()
    np.savez_compressed(os.path.join(HERE, "collate_ref.npz"), **out)
    print("collate_ref.npz:", {k: tuple(v.shape) for k, v in res.items()})


def collate_qwen(D):
    """The same generic Collator on Qwen2-VL-style items (what Qwen2VLProcessor returns per sample: flattened patches [n_patches, C*tp*p*p]
    and image_grid_thw [n_images, 3]): every key other than ids / masks / labels / position ids is concatenated along dim 0."""
    rng = np.random.default_rng(17)
    col = D.Collator(processor=object())
    col.tokenizer = _Tok()
    items = []
    for T, grids in ((17, [(1, 2, 4), (1, 4, 2)]), (9, [(1, 2, 2)]), (21, [(1, 4, 4), (1, 2, 2), (1, 2, 6)])):
        ids = rng.integers(1, 280, size=T, dtype=np.int64)
        lab = ids.copy()
        lab[: T // 3] = -100
        npatch = sum(t * h * w for t, h, w in grids)
        items.append(dict(input_ids=torch.from_numpy(ids)[None], attention_mask=torch.ones(1, T, dtype=torch.int64),
                          labels=torch.from_numpy(lab)[None],
                          pixel_values=torch.from_numpy(rng.standard_normal((npatch, 12)).astype(np.float32)),
                          image_grid_thw=torch.tensor(grids, dtype=torch.int64)))
    res = col(items)
    keys = ("input_ids", "attention_mask", "labels", "pixel_values", "image_grid_thw")
    out = {"n": np.array(len(items)), "pad_token_id": np.array(PAD)}
    for i, it in enumerate(items):
        for k in keys:
            out[f"s{i}.{k}"] = it[k].numpy()
    for k in keys:
        out[f"out.{k}"] = res[k].numpy()
    np.savez_compressed(os.path.join(HERE, "collate_qwen_ref.npz"), **out)
    print("collate_qwen_ref.npz:", {k: tuple(v.shape) for k, v in res.items()})


def pack_batch(D):
    rng = np.random.default_rng(13)
    pk = object.__new__(D.PackingDataset)
    out = {}
    # equal lengths (the only shape for which the reference's dim-0 label concat is defined), one item with masked positions
    for case, lens in (("eq", [12, 12, 12]), ("ragged", [9, 14, 5, 12])):
        items = [_item(rng, T, 1 + (i % 2)) for i, T in enumerate(lens)]
        items[1]["attention_mask"][0, -3:] = 0
        if case == "ragged":
            for it in items:                          # reference: torch.cat(labels, dim=0) needs equal widths -> hand it [T] rows
                it["labels"] = it["labels"][0]
        res = pk.pack_batch(items)
        out[f"{case}.n"] = np.array(len(items))
        for i, it in enumerate(items):
            for k in ("input_ids", "attention_mask", "labels", "pixel_values"):
                out[f"{case}.s{i}.{k}"] = it[k].numpy()
        for k in ("input_ids", "attention_mask", "position_ids", "labels", "pixel_values"):
            out[f"{case}.out.{k}"] = res[k].numpy()
        print(f"pack_batch_ref.npz[{case}]:", {k: tuple(v.shape) for k, v in res.items()})
    np.savez_compressed(os.path.join(HERE, "pack_batch_ref.npz"), **out)


def main():
    D, Style = _import_reference_data()
    label_rule(D, Style)
    collate(D)
    collate_qwen(D)
    pack_batch(D)


if __name__ == "__main__":
    main()
