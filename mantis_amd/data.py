"""Batch producer for the hot path: label masking + collation (SURVEY.md section 8a "Python callers" / row f4).

Reference behaviour:
  * label mask of ChatDataset.getitem       /root/reference/mantis/train/data.py:415-466
      LLAMA_3 / SINGLE: labels = -100 everywhere, then for odd i copy ids[sep[i]+1 : sep[i+1]+1] (last span: to the end),
      sep = positions of the separator token (<|eot_id|>);  PLAIN (pre-training): labels = ids except <image> tokens.
  * collation                               /root/reference/mantis/train/data.py:1392-1527 and
    MLlavaProcessor._right_pad_inputs_with_attention_mask  mantis/models/mllava/processing_llava.py:277-285
      right-pad input_ids with pad_token_id, attention_mask with 0, labels with -100; `pixel_values` stays a LIST with one
      [n_i, 3, H, W] tensor per sample.  The reference asserts a single sample per batch (:279); this collator accepts
      any batch size (vectorised numpy), which is what configs[1]/[2] (bs = 2 per GPU) need.
  * sample packing                           /root/reference/mantis/train/data.py:1546-1671 (PackingDataset.pack_batch)
      several samples concatenated into ONE row: ids [1, sum T_i], a block-diagonal 4-D attention mask (block i = sample i's
      key mask broadcast over its query rows), position ids restarting at 0 per sample, labels concatenated.
      `pack_samples` reproduces those outputs and adds the compact form the gfx950 attention kernels consume instead of the
      O(L^2) mask: `segment_ids` [1, sum T_i] (+ `cu_seqlens`).
String-level work (chat templates, tokenisation, PIL) stays with the caller's tokenizer/processor: inputs here are token ids."""
import numpy as np
import torch

IGNORE_INDEX = -100


def llama3_label_mask(input_ids, sep_id, ignore_index=IGNORE_INDEX):
    ids = np.asarray(input_ids, dtype=np.int64)
    target = np.full_like(ids, ignore_index)
    sep = np.flatnonzero(ids == sep_id)
    for i in range(1, len(sep), 2):
        if i == len(sep) - 1:
            target[sep[i] + 1:] = ids[sep[i] + 1:]
        else:
            target[sep[i] + 1: sep[i + 1] + 1] = ids[sep[i] + 1: sep[i + 1] + 1]
    return target


def plain_label_mask(input_ids, image_token_id, ignore_index=IGNORE_INDEX):
    ids = np.asarray(input_ids, dtype=np.int64)
    return np.where(ids != image_token_id, ids, ignore_index)


class Collator:
    """samples: dicts with `input_ids` [T_i] (+ optional `labels` [T_i], `pixel_values` [n_i,3,H,W]).  Returns the batch dict
    `Trainer.training_step` consumes: input_ids / attention_mask / labels int64 [B, T_max] and pixel_values as a list.
    Qwen2-VL-style samples (`image_grid_thw` present: `pixel_values` = flattened patches [n_patches_i, C*tp*p*p]) and any other tensor
    key follow the reference Collator's generic rule (data.py:1521-1522): concatenated along dim 0."""

    def __init__(self, pad_token_id, image_token_id=None, max_length=None, pin_memory=False):
        self.pad_token_id, self.image_token_id, self.max_length, self.pin = pad_token_id, image_token_id, max_length, pin_memory

    def __call__(self, samples):
        ids = [np.asarray(s["input_ids"], dtype=np.int64).reshape(-1) for s in samples]
        if self.max_length is not None:
            ids = [x[: self.max_length] for x in ids]
        T = max(len(x) for x in ids)
        B = len(ids)
        out_ids = np.full((B, T), self.pad_token_id, dtype=np.int64)
        mask = np.zeros((B, T), dtype=np.int64)
        labels = np.full((B, T), IGNORE_INDEX, dtype=np.int64)
        for b, x in enumerate(ids):
            out_ids[b, : len(x)] = x
            mask[b, : len(x)] = 1
            if samples[b].get("labels") is not None:
                lab = np.asarray(samples[b]["labels"], dtype=np.int64).reshape(-1)[: len(x)]
                labels[b, : len(lab)] = lab
        known = ("input_ids", "attention_mask", "labels", "pixel_values")
        extra = {}
        for k in samples[0]:
            if k in known or samples[0][k] is None:
                continue
            extra[k] = _pack_extra([s[k] for s in samples if s.get(k) is not None])
        first_pv = next((s["pixel_values"] for s in samples if s.get("pixel_values") is not None), None)
        idefics2_style = first_pv is not None and torch.as_tensor(first_pv).dim() == 5      # [1, n_images, 3, H, W] per sample
        if "image_grid_thw" in extra or idefics2_style:
            # Qwen2-VL: patches of all samples in one [sum n_patches, C*tp*p*p] tensor; Idefics2: [B, n_images, 3, H, W]
            pvs = [torch.as_tensor(s["pixel_values"], dtype=torch.float32) for s in samples if s.get("pixel_values") is not None]
            pv = torch.cat(pvs, dim=0) if pvs else None
            if pv is not None and self.pin and torch.cuda.is_available():
                pv = pv.pin_memory()
            batch = dict(input_ids=torch.from_numpy(out_ids), attention_mask=torch.from_numpy(mask), labels=torch.from_numpy(labels),
                         pixel_values=pv)
            batch.update(extra)
            return batch
        pix = []
        for b, s in enumerate(samples):
            pv = s.get("pixel_values")
            if pv is None:
                continue
            pv = torch.as_tensor(pv, dtype=torch.float32)
            if self.image_token_id is not None:
                # processing_llava.py:240-246: images whose <image> token was truncated away are dropped
                keep = int((out_ids[b] == self.image_token_id).sum())
                pv = pv[:keep]
            pix.append(pv.pin_memory() if self.pin and torch.cuda.is_available() else pv)
        batch = dict(input_ids=torch.from_numpy(out_ids), attention_mask=torch.from_numpy(mask), labels=torch.from_numpy(labels))
        batch["pixel_values"] = pix if pix else None
        batch.update(extra)
        return batch


def _row(x, dtype=np.int64):
    a = np.asarray(x.numpy() if isinstance(x, torch.Tensor) else x, dtype=dtype)
    return a.reshape(-1)


def pack_samples(samples, materialize_mask=True):
    """PackingDataset.pack_batch (data.py:1609-1671) for samples with `input_ids` [1,T_i] or [T_i], optional `attention_mask`,
    `labels`, `pixel_values`.  Returns
        input_ids      int64 [1, S]            S = sum T_i                                   (:1612)
        attention_mask int32 [1, 1, S, S]      block-diagonal; block i = sample i's key mask on every query row (:1627-1638);
                                               None when materialize_mask=False (S = 5624 would be 126 MB nobody reads)
        position_ids   int64 [S]               arange(T_i) per sample                        (:1641-1648)
        labels         int64 [1, S]            the reference concatenates along dim 0, which for its [1,T] items only works for
                                               equal T_i and yields [n, T]; row-major that is this [1, S] vector
        pixel_values   tensor [sum n_i, ...] (tensors concatenated, :1619-1621) / list / None
        segment_ids    int32 [1, S], cu_seqlens int32 [n+1], key_mask int64 [1, S]  -- compact equivalents of the 4-D mask"""
    ids = [_row(s["input_ids"]) for s in samples]
    lens = [len(x) for x in ids]
    S = int(sum(lens))
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    keym = [(_row(s["attention_mask"]) if s.get("attention_mask") is not None else np.ones(t, np.int64)) for s, t in zip(samples, lens)]
    labels = [(_row(s["labels"]) if s.get("labels") is not None else np.full(t, IGNORE_INDEX, np.int64)) for s, t in zip(samples, lens)]
    out = dict(input_ids=torch.from_numpy(np.concatenate(ids))[None],
               position_ids=torch.from_numpy(np.concatenate([np.arange(t, dtype=np.int64) for t in lens])),
               labels=torch.from_numpy(np.concatenate(labels))[None],
               segment_ids=torch.from_numpy(np.repeat(np.arange(len(lens), dtype=np.int32), lens))[None],
               cu_seqlens=torch.from_numpy(cu), key_mask=torch.from_numpy(np.concatenate(keym))[None])
    mask = None
    if materialize_mask:
        mask = torch.zeros((1, 1, S, S), dtype=torch.int32)
        for i, t in enumerate(lens):
            a, b = int(cu[i]), int(cu[i + 1])
            mask[0, 0, a:b, a:b] = torch.from_numpy(keym[i]).to(torch.int32)[None, :].expand(t, t)
    out["attention_mask"] = mask
    pv = [s.get("pixel_values") for s in samples]
    if all(p is None for p in pv):
        out["pixel_values"] = None
    elif isinstance(next(p for p in pv if p is not None), (list, tuple)):
        out["pixel_values"] = sum([list(p or []) for p in pv], []) or None
    else:
        out["pixel_values"] = torch.cat([torch.as_tensor(p) for p in pv if p is not None], dim=0)
    for k in samples[0]:                  # e.g. Qwen2-VL's image_grid_thw: one row per image, images in order of appearance
        if k not in out and k != "pixel_values" and samples[0][k] is not None:
            out[k] = _pack_extra([s[k] for s in samples if s.get(k) is not None])
    return out


def _pack_extra(vals):
    """The reference's `rest_keys` rule (data.py:1659-1666): tensors (and arrays) concatenated along dim 0, lists concatenated,
    anything else (strings, ids, dicts) collected into a list."""
    v0 = vals[0]
    if isinstance(v0, (torch.Tensor, np.ndarray)):
        return torch.cat([torch.as_tensor(v) for v in vals], dim=0)
    if isinstance(v0, list):
        return sum(vals, [])
    return list(vals)


def segments_from_packed(batch):
    """(segment_ids int32 [1,S], key_mask int64 [1,S]) from a batch in the reference's packed format (PackingDataset.pack_batch,
    data.py:1609-1671): `position_ids` restart at 0 at every sample start (:1641-1648) and the diagonal of the 4-D block-diagonal
    mask is the per-token key mask (:1627-1638).  `pack_samples` output carries `segment_ids` / `key_mask` directly."""
    if batch.get("segment_ids") is not None and batch.get("key_mask") is not None:
        return batch["segment_ids"], batch["key_mask"]
    if batch.get("position_ids") is None:
        raise ValueError("a packed batch needs `position_ids` (restarting at 0 per sample) or `segment_ids` + `key_mask`")
    ids = torch.as_tensor(batch["input_ids"])
    m = torch.as_tensor(batch["attention_mask"])
    # the reference's packed batch is ONE row (pack_batch concatenates along the sequence, :1612; its collator never stacks packed rows)
    if ids.dim() != 2 or ids.shape[0] != 1 or m.dim() != 4 or m.shape[0] != 1:
        raise ValueError(f"a packed batch is one row: input_ids [1, S] and a [1, 1, S, S] mask; got {tuple(ids.shape)} / {tuple(m.shape)}")
    pos = torch.as_tensor(batch["position_ids"]).reshape(-1)
    if pos.numel() != ids.shape[1]:
        raise ValueError(f"packed position_ids have {pos.numel()} entries for a row of {ids.shape[1]} tokens")
    seg = (torch.cumsum((pos == 0).to(torch.int32), 0) - 1).to(torch.int32)[None]
    key = torch.diagonal(m[0, 0], 0).to(torch.int64)[None]
    return seg, key
