"""ORACLE (test infrastructure, never imported by the product path).

numpy restatement of the integer side of
``LlavaForConditionalGeneration._merge_input_ids_with_image_features``
(/root/reference/mantis/models/mllava/modeling_llava.py:293-360), in the
index-only formulation of SURVEY.md appendix A.  Pinned against the merged
attention-mask / labels / position-ids recorded from the reference in
tests/golden/*.npz (tests/test_oracle_vs_golden.py), including the
unequal-image-count quirk case.

Outputs a *source map* instead of copying embeddings, so the same plan drives
the bf16 row copies (bit-exact by construction, a copy is a copy):
  src_kind[b, p]  0 = padding slot (zero row), 1 = text token, 2 = image-feature row
  src_idx[b, p]   text: t (column of input_ids)    image: global feature row (batch-major)
"""
import numpy as np

PAD_SLOT, TEXT, IMAGE = 0, 1, 2


class PackCountError(ValueError):
    """same class of failure as modeling_llava.py:347-351 (ValueError)."""


def pack_plan(input_ids, attention_mask, labels, num_images, num_patches, image_token_index, pad_token_id,
              ignore_index=-100, fix_unequal_counts=False):
    """fix_unequal_counts=False: the reference's placement, including its mis-placement of image rows for a right-padded
    batch with unequal image counts.  True: SURVEY appendix A's index-only formulation -- image j of
    sample b occupies [p[b,t_j]-(N-1), p[b,t_j]] whatever the padding side; every other unwritten slot stays padding.  Equal
    to the reference whenever all samples hold the same number of images (and, sample by sample, to the reference run at
    B = 1: tests/golden/make_golden_fixcounts.py)."""
    ids = np.asarray(input_ids, dtype=np.int64)
    attn = np.asarray(attention_mask, dtype=np.int64)
    B, T = ids.shape
    N = int(num_patches)
    # :296  left_padding = not any(ids[:, -1] == pad)
    left_padding = not bool(np.sum(ids[:, -1] == pad_token_id))
    # :298-301
    m = ids == image_token_index
    k = m.sum(-1)
    L = int(k.max()) * (N - 1) + T
    # :309-312
    p = np.cumsum(m * (N - 1) + 1, -1) - 1
    nb_image_pad = L - 1 - p[:, -1]
    if left_padding:
        p = p + nb_image_pad[:, None]
    src_kind = np.zeros((B, L), dtype=np.int32)
    src_idx = np.zeros((B, L), dtype=np.int64)
    out_mask = np.zeros((B, L), dtype=np.int64)
    out_lab = np.full((B, L), ignore_index, dtype=np.int64)
    # :338-341 scatter text
    bi, ti = np.nonzero(~m)
    tp = p[bi, ti]
    src_kind[bi, tp] = TEXT
    src_idx[bi, tp] = ti
    out_mask[bi, tp] = attn[bi, ti]
    if labels is not None:
        out_lab[bi, tp] = np.asarray(labels, dtype=np.int64)[bi, ti]
    # :344-345 image slots = unwritten rows minus the first nb_image_pad of them, per sample
    unwritten = src_kind == 0
    if fix_unequal_counts:
        slots = np.zeros((B, L), dtype=bool)
        ib, it = np.nonzero(m)                     # row-major: batch-major, in-sample order = the feature rows' order
        for b_, t_ in zip(ib, it):
            slots[b_, p[b_, t_] - (N - 1): p[b_, t_] + 1] = True
    else:
        rank = np.cumsum(unwritten, -1) - 1
        slots = unwritten & (rank >= nb_image_pad[:, None])
    # :347-351
    if int(slots.sum()) != int(num_images) * N:
        raise PackCountError(
            f"The input provided to the model are wrong. The number of image tokens is {int(m.sum())} while"
            f" the number of image given to the model is {int(num_images)}.")
    # :353 fill in row-major order
    sb, sp = np.nonzero(slots)
    src_kind[sb, sp] = IMAGE
    src_idx[sb, sp] = np.arange(sb.size, dtype=np.int64)
    # :354-355
    out_mask |= slots.astype(np.int64)
    pos = np.cumsum(out_mask, -1) - 1
    pos[out_mask == 0] = 1
    return dict(L=L, src_kind=src_kind, src_idx=src_idx, attention_mask=out_mask,
                labels=out_lab if labels is not None else None, position_ids=pos, text_pos=p,
                left_padding=left_padding, nb_image_pad=nb_image_pad)


def pack_rows(plan, text_embeds, image_features):
    """Apply the plan to numpy arrays: text_embeds [B,T,d], image_features [I,N,d] -> [B,L,d]."""
    B, L = plan["src_kind"].shape
    d = text_embeds.shape[-1]
    out = np.zeros((B, L, d), dtype=text_embeds.dtype)
    feats = image_features.reshape(-1, d)
    for b in range(B):
        tk = plan["src_kind"][b] == TEXT
        out[b, tk] = text_embeds[b, plan["src_idx"][b, tk]]
        ik = plan["src_kind"][b] == IMAGE
        out[b, ik] = feats[plan["src_idx"][b, ik]]
    return out


def llama3_label_mask(ids, sep_id, ignore_index=-100):
    """Label rule of ChatDataset.getitem for SeparatorStyle.LLAMA_3 / SINGLE
    (/root/reference/mantis/train/data.py:415,432-442)."""
    ids = np.asarray(ids, dtype=np.int64)
    target = np.full_like(ids, ignore_index)
    sep = np.nonzero(ids == sep_id)[0].tolist()
    for i in range(len(sep)):
        if i % 2 == 0:
            continue
        if i == len(sep) - 1:
            target[sep[i] + 1:] = ids[sep[i] + 1:]
        else:
            target[sep[i] + 1:sep[i + 1] + 1] = ids[sep[i] + 1:sep[i + 1] + 1]
    return target


def plain_label_mask(ids, image_token_id, ignore_index=-100):
    """SeparatorStyle.PLAIN branch (/root/reference/mantis/train/data.py:457-461)."""
    ids = np.asarray(ids, dtype=np.int64)
    target = np.full_like(ids, ignore_index)
    keep = ids != image_token_id
    target[keep] = ids[keep]
    return target


def pack_segments(plan, input_ids, segment_ids, num_patches, image_token_index):
    """Packed samples (/root/reference/mantis/train/data.py:1546-1671, PackingDataset.pack_batch: block-diagonal mask :1627-1638,
    position ids restarting per sample :1641-1648) carried through the image-token expansion of modeling_llava.py:293-360.
    plan: the dict returned by pack_plan for the packed row(s).  Returns (kstart, qend, position_ids, sample_start_tokens):
      kstart[b, p] / qend[b, p]   first / one-past-last merged position of p's sample (the O(L) form of the 4-D mask)
      position_ids                cumsum(mask) restarted at every sample start; 1 where mask == 0 (modeling_llava.py:355)
      sample_start_tokens[b, t]   True for the first token of a sample (it must not be predicted from the previous sample)."""
    ids = np.asarray(input_ids, dtype=np.int64)
    seg = np.asarray(segment_ids, dtype=np.int64)
    B, T = ids.shape
    L = plan["L"]
    N = int(num_patches)
    mask = plan["attention_mask"]
    kstart = np.zeros((B, L), dtype=np.int32)
    qend = np.full((B, L), L, dtype=np.int32)
    pos = np.ones((B, L), dtype=np.int64)
    first = np.zeros((B, T), dtype=bool)
    for b in range(B):
        lens = np.where(ids[b] == image_token_index, N, 1)
        ends = np.cumsum(lens)
        starts = ends - lens + (L - int(ends[-1]))
        first[b] = np.concatenate([[True], seg[b, 1:] != seg[b, :-1]])
        marks = sorted(int(s) for s in starts[first[b]] if 0 <= s < L)
        edges = [0] + [m for m in marks if m > 0] + [L]
        for a, e in zip(edges[:-1], edges[1:]):
            kstart[b, a:e] = a
            qend[b, a:e] = e
        c = np.cumsum(mask[b])
        base = np.where(kstart[b] > 0, c[np.maximum(kstart[b] - 1, 0)], 0)
        pos[b] = np.where(mask[b] == 0, 1, c - base - 1)
    return kstart, qend, pos, first
