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
    `Trainer.training_step` consumes: input_ids / attention_mask / labels int64 [B, T_max] and pixel_values as a list."""

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
        return batch
