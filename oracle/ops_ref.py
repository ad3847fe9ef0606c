"""ORACLE (test infrastructure, never imported by the product path).

torch-CPU restatement of every operator in `mantis_amd/hip_ops.py`, with the SAME function signatures, so that
  * `tests/test_hip_ops_gpu.py` (-m gpu) can compare each gfx950 kernel with its reference on identical inputs, and
  * `tests/test_engine_host_logic.py` (CPU) can drive the product's host logic (mantis_amd/engine.py: kernel sequencing,
    what is saved / recomputed, gradient plumbing) with this module monkeypatched in place of the HIP backend and
    check it against the golden vectors recorded from the reference.
Math is done in fp32 on the (bf16 or fp32) inputs and rounded once to the input dtype, except where the reference's own
bf16 graph rounds more often (RMSNorm, SwiGLU, RoPE: noted inline).  Per-op formulas follow oracle/llava_ref.py, which
is pinned against the reference by tests/test_oracle_vs_golden.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import llava_ref as R
from . import pack_ref


def pad8(n):
    return (n + 7) // 8 * 8


def _f(x):
    return x.float()


def gemm_nt(a, b, bias=None, act=None, residual=None, out=None, accumulate=False, n_valid=None, k=None, ldc=None):
    K = a.shape[1] if k is None else k
    N = b.shape[0] if n_valid is None else n_valid
    av, bv = _f(a)[:, :K], _f(b)[:N, :K]
    if av.shape[1] < K:      # caller relies on zero padding of the shorter operand
        av = F.pad(av, (0, K - av.shape[1]))
    if bv.shape[1] < K:
        bv = F.pad(bv, (0, K - bv.shape[1]))
    y = av @ bv.t()
    dt = a.dtype
    if bias is not None:
        y = y + _f(bias)[:N]
    if act is not None:
        y = R.ACT[act](y.to(dt).float())
    if residual is not None:
        y = y.to(dt).float() + _f(residual)[:, :N]
    if out is None:
        out = torch.zeros((a.shape[0], N if ldc is None else ldc), dtype=dt)
        out[:, :N] = y.to(dt)
        return out
    if accumulate:
        y = y + _f(out)[:, :N]
    out[:, :N] = y.to(out.dtype)
    return out


# ---- fp8 linears: exact restatement of csrc/gemm_fp8.hip's arithmetic (same scales, same round-to-nearest-even conversion)
FP8_E4M3, FP8_E5M2 = 0, 1
_F8 = {0: (torch.float8_e4m3fn, 448.0), 1: (torch.float8_e5m2, 57344.0)}


class Fp8Tensor:
    def __init__(self, q, qt, state, fmt, rows, cols, row_dequant=None, col_dequant=None):
        self.q, self.qt, self.state, self.fmt, self.rows, self.cols = q, qt, state, fmt, rows, cols
        self.row_dequant, self.col_dequant = row_dequant, col_dequant

    @property
    def rowwise(self):
        return self.state is None

    @property
    def dequant(self):
        return self.row_dequant if self.state is None else self.state[2:3]

    @property
    def dequant_t(self):
        return self.col_dequant if self.state is None else self.state[2:3]


def _fp8_quantize_2d(x, fmt, transposed, rowmajor):
    """csrc/gemm_fp8.hip fp8_amax2d_kernel / fp8_cast2d_kernel: one scale per row for q, one per column of x for qt"""
    dt, fmax = _F8[fmt]
    xf = _f(x)
    rows, cols = x.shape
    fm = torch.tensor(fmax, dtype=torch.float32)
    one = torch.tensor(1.0)
    q = qt = rd = cd = None
    if rowmajor:
        am = xf.abs().amax(dim=1)
        sc = torch.where(am > 0, fm / am, one)
        rd = torch.where(am > 0, am / fm, one)
        q = (xf * sc[:, None]).clamp(-fmax, fmax).to(dt).view(torch.uint8)
    if transposed:
        am = xf.abs().amax(dim=0)
        sc = torch.where(am > 0, fm / am, one)
        cd = torch.where(am > 0, am / fm, one)
        rp = (rows + 15) // 16 * 16
        qt = torch.zeros((cols, rp), dtype=torch.uint8)
        qt[:, :rows] = (xf * sc[None, :]).clamp(-fmax, fmax).to(dt).view(torch.uint8).t()
    return Fp8Tensor(q, qt, None, fmt, rows, cols, rd, cd)


def fp8_quantize(x, fmt=FP8_E4M3, transposed=True, rowmajor=True, amax=None, rowwise=False):
    if rowwise:
        return _fp8_quantize_2d(x, fmt, transposed, rowmajor)
    assert rowmajor, "the per-tensor quantiser always writes the row-major copy"
    dt, fmax = _F8[fmt]
    xf = _f(x)
    if amax is not None:
        assert float(amax.max()) == float(xf.abs().max()), "a producer-side amax must equal the tensor's own"
    amax = xf.abs().max()
    fm = torch.tensor(fmax, dtype=torch.float32)
    sc = fm / amax if float(amax) > 0 else torch.tensor(1.0)
    dq = amax / fm if float(amax) > 0 else torch.tensor(1.0)
    q = (xf * sc).clamp(-fmax, fmax).to(dt).view(torch.uint8)
    rows, cols = x.shape
    qt = None
    if transposed:
        rp = (rows + 15) // 16 * 16
        qt = torch.zeros((cols, rp), dtype=torch.uint8)
        qt[:, :rows] = q.t()
    return Fp8Tensor(q, qt, torch.stack([amax, sc, dq]).float(), fmt, rows, cols)


def gemm_fp8_nt(a8, a_dequant, b8, b_dequant, fmt_a=FP8_E4M3, bias=None, residual=None, out=None, accumulate=False, k=None, variant=0,
                rowwise=False):
    K = a8.shape[1] if k is None else k
    av = a8[:, :K].view(_F8[fmt_a][0]).float()
    bv = b8[:, :K].view(_F8[0][0]).float()
    if rowwise:
        assert a_dequant.numel() == a8.shape[0] and b_dequant.numel() == b8.shape[0]
        y = (av @ bv.t()) * (a_dequant.float()[:, None] * b_dequant.float()[None, :])
    else:
        assert a_dequant.numel() == 1 and b_dequant.numel() == 1
        y = (av @ bv.t()) * (a_dequant.float() * b_dequant.float())
    if bias is not None:
        y = y + _f(bias)
    if residual is not None:
        y = y.to(torch.bfloat16).float() + _f(residual)
    if out is None:
        return y.to(torch.bfloat16)
    if accumulate:
        y = y + _f(out)
    out.copy_(y.to(out.dtype))
    return out


def gemm_fp8_dx_swiglu(dy8, dy_dequant, wt8, w_dequant, gu, fmt_a=FP8_E5M2):
    dgu = swiglu_bwd(gemm_fp8_nt(dy8, dy_dequant, wt8, w_dequant, fmt_a), gu)
    return dgu, dgu.float().abs().max().reshape(1)


def transpose(x, rpad=None):
    Rr, C = x.shape
    Rp = pad8(Rr) if rpad is None else rpad
    out = torch.zeros((C, Rp), dtype=x.dtype)
    out[:, :Rr] = x.t()
    return out


def linear_fwd(x, w, bias=None, act=None, residual=None):
    return gemm_nt(x, w, bias=bias, act=act, residual=residual)


def linear_dx(dy, w, k=None):
    n = w.shape[0]
    return (_f(dy)[:, :n] @ _f(w)).to(dy.dtype)


def side_stream(default="0"):
    return None


def linear_gu_swiglu(x, w_gu, variant=0, amax_parts=None):
    gu = gemm_nt(x, w_gu)
    return gu, swiglu_fwd(gu)


def linear_qkv_rope(x, w_qkv, bias, cos, sin, n_rope_heads, hd, variant=0):
    return rope_apply_(gemm_nt(x, w_qkv, bias=bias), cos, sin, n_rope_heads, hd)


def linear_dx_swiglu(dy, w_down, gu):
    return swiglu_bwd(linear_dx(dy, w_down), gu)


def linear_dw(dy, x, grad_w, accumulate):
    n = grad_w.shape[0]
    g = _f(dy)[:, :n].t() @ _f(x)
    if accumulate:
        g = g + _f(grad_w)
    grad_w.copy_(g.to(grad_w.dtype))


def colsum(x, grad, accumulate):
    g = _f(x).sum(0)
    if accumulate:
        g = g + _f(grad)
    grad.copy_(g.to(grad.dtype))


def amax_parts_buffer(device):
    return torch.zeros(2048, dtype=torch.float32)


def _put_amax(parts, y):
    if parts is not None:
        parts.zero_()
        parts[0] = y.float().abs().max()


def rmsnorm_fwd(x, w, eps, want_rstd=True, amax_parts=None):
    xf = _f(x)
    rstd = torch.rsqrt(xf.pow(2).mean(-1) + eps)
    y = (_f((xf * rstd[:, None]).to(x.dtype)) * _f(w)).to(x.dtype)      # two roundings, as LlamaRMSNorm does in bf16
    _put_amax(amax_parts, y)
    return y, (rstd if want_rstd else None)


def rmsnorm_bwd(dy, x, w, rstd, dres, grad_w, accumulate, amax_parts=None):
    xf, g = _f(x), _f(dy) * _f(w)
    xhat = xf * rstd[:, None]
    dx = rstd[:, None] * (g - xhat * (g * xhat).mean(-1, keepdim=True))
    if dres is not None:
        dx = dx + _f(dres)
    if grad_w is not None:
        gw = (_f(dy) * xhat).sum(0)
        if accumulate:
            gw = gw + _f(grad_w)
        grad_w.copy_(gw.to(grad_w.dtype))
    _put_amax(amax_parts, dx.to(x.dtype))
    return dx.to(x.dtype)


def layernorm_fwd(x, w, b, eps):
    return F.layer_norm(_f(x), (x.shape[-1],), _f(w), _f(b), eps).to(x.dtype)


def swiglu_fwd(gu, amax_parts=None):
    I = gu.shape[1] // 2
    g, u = _f(gu[:, :I]), _f(gu[:, I:])
    out = (_f(F.silu(g).to(gu.dtype)) * u).to(gu.dtype)
    _put_amax(amax_parts, out)
    return out


def swiglu_bwd(dact, gu):
    I = gu.shape[1] // 2
    g, u, d = _f(gu[:, :I]), _f(gu[:, I:]), _f(dact)
    s = torch.sigmoid(g)
    silu = g * s
    return torch.cat([d * u * (s + silu * (1 - s)), d * silu], dim=1).to(gu.dtype)


def act_fwd(x, kind):
    return R.ACT[kind](_f(x)).to(x.dtype)


def act_bwd(dy, x, kind):
    xf = _f(x).detach().requires_grad_(True)
    with torch.enable_grad():
        y = R.ACT[kind](xf)
    (g,) = torch.autograd.grad(y, xf, _f(dy))
    return g.to(x.dtype)


def add(a, b):
    return (_f(a) + _f(b)).to(a.dtype)


def rope_table(position_ids, inv_freq):
    f = position_ids.reshape(-1, 1).float() * inv_freq.float()[None]
    return f.cos().to(torch.bfloat16), f.sin().to(torch.bfloat16)


def rope_table_sections(position_ids, inv_freq, section_of_freq):
    sel = position_ids[section_of_freq.long()].t().float()                 # [R, half]: frequency j reads row section_of_freq[j]
    f = sel * inv_freq.float()[None]
    return f.cos().to(torch.bfloat16), f.sin().to(torch.bfloat16)


def rope_apply_(x, cos, sin, nheads, hd, backward=False):
    Rr = x.shape[0]
    half = hd // 2
    v = x[:, : nheads * hd].reshape(Rr, nheads, hd)
    x1, x2 = _f(v[..., :half]), _f(v[..., half:])
    c, s = _f(cos)[:, None, :], _f(sin)[:, None, :]
    dt = x.dtype
    if not backward:
        r = lambda t: t.to(dt).float()      # the reference's bf16 graph rounds each product and the sum
        o1 = r(x1 * c) - r(x2 * s)
        o2 = r(x2 * c) + r(x1 * s)
    else:
        o1 = x1 * c + x2 * s
        o2 = x2 * c - x1 * s
    x[:, : nheads * hd] = torch.cat([o1, o2], dim=-1).reshape(Rr, nheads * hd).to(dt)
    return x


def _split_qkv(qkv, B, L, H, Hkv, hd):
    q = qkv[:, : H * hd].reshape(B, L, H, hd).transpose(1, 2)
    k = qkv[:, H * hd: (H + Hkv) * hd].reshape(B, L, Hkv, hd).transpose(1, 2)
    v = qkv[:, (H + Hkv) * hd: (H + 2 * Hkv) * hd].reshape(B, L, Hkv, hd).transpose(1, 2)
    return q, k, v


def _scores(q, k, scale, causal, kmask, H, Hkv, kstart=None):
    rep = H // Hkv
    kk = k.repeat_interleave(rep, dim=1) if rep > 1 else k
    s = torch.matmul(_f(q), _f(kk).transpose(-1, -2)) * scale
    L, Lk = s.shape[-2:]
    if causal:
        s = s.masked_fill(~torch.ones(L, Lk, dtype=torch.bool).tril(), float("-inf"))
    if kmask is not None:
        s = s.masked_fill(kmask[:, None, None, :] == 0, float("-inf"))
    if kstart is not None:          # packed samples: query q sees keys >= kstart[b, q] only (block-diagonal mask, data.py:1627-1638)
        keys = torch.arange(Lk)[None, None, None, :]
        s = s.masked_fill(keys < kstart.long()[:, None, :, None], float("-inf"))
    return s


def attn_fwd(qkv, B, Lseq, H, Hkv, hd, kmask, scale, causal, want_lse=True, kstart=None):
    q, k, v = _split_qkv(qkv, B, Lseq, H, Hkv, hd)
    s = _scores(q, k, scale, causal, kmask, H, Hkv, kstart)
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None]).nan_to_num(0.0)
    rep = H // Hkv
    vv = v.repeat_interleave(rep, dim=1) if rep > 1 else v
    o = torch.matmul(p, _f(vv)).transpose(1, 2).reshape(B * Lseq, H * hd).to(qkv.dtype)
    lse = torch.where(torch.isinf(lse) & (lse < 0), torch.full_like(lse, float("inf")), lse)
    return o, (lse if want_lse else None)


def attn_bwd(qkv, o, do, lse, B, Lseq, H, Hkv, hd, kmask, scale, causal, kstart=None, qend=None):
    q, k, v = _split_qkv(qkv, B, Lseq, H, Hkv, hd)
    rep = H // Hkv
    s = _scores(q, k, scale, causal, kmask, H, Hkv, kstart)
    p = torch.exp(s - lse[..., None]).nan_to_num(0.0)
    dO = _f(do).reshape(B, Lseq, H, hd).transpose(1, 2)
    O = _f(o).reshape(B, Lseq, H, hd).transpose(1, 2)
    vv = _f(v).repeat_interleave(rep, dim=1) if rep > 1 else _f(v)
    kk = _f(k).repeat_interleave(rep, dim=1) if rep > 1 else _f(k)
    dsum = (dO * O).sum(-1, keepdim=True)
    dP = torch.matmul(dO, vv.transpose(-1, -2))
    dS = p * (dP - dsum) * scale
    dq = torch.matmul(dS, kk)
    dk = torch.matmul(dS.transpose(-1, -2), _f(q))
    dv = torch.matmul(p.transpose(-1, -2), dO)
    if rep > 1:
        dk = dk.reshape(B, Hkv, rep, Lseq, hd).sum(2)
        dv = dv.reshape(B, Hkv, rep, Lseq, hd).sum(2)
    out = torch.zeros_like(qkv)
    out[:, : H * hd] = dq.transpose(1, 2).reshape(B * Lseq, H * hd).to(qkv.dtype)
    out[:, H * hd: (H + Hkv) * hd] = dk.transpose(1, 2).reshape(B * Lseq, Hkv * hd).to(qkv.dtype)
    out[:, (H + Hkv) * hd: (H + 2 * Hkv) * hd] = dv.transpose(1, 2).reshape(B * Lseq, Hkv * hd).to(qkv.dtype)
    return out


def attn_fwd_cross(q, k, v, B, Lq, Lk, H, Hkv, hd, kmask, scale):
    """Restates the perceiver attention of /root/reference/mantis/models/idefics2/modeling_idefics2.py:812-912 (eager path :873-905): Lq
    queries per batch entry over Lk keys, GQA by repeat, fp32 softmax, non-causal, key mask."""
    qh = q[:, : H * hd].reshape(B, Lq, H, hd).transpose(1, 2)
    kh = k[:, : Hkv * hd].reshape(B, Lk, Hkv, hd).transpose(1, 2)
    vh = v[:, : Hkv * hd].reshape(B, Lk, Hkv, hd).transpose(1, 2)
    s = _scores(qh, kh, scale, False, kmask, H, Hkv)
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None]).nan_to_num(0.0)
    rep = H // Hkv
    vv = vh.repeat_interleave(rep, dim=1) if rep > 1 else vh
    o = torch.matmul(p, _f(vv)).transpose(1, 2).reshape(B * Lq, H * hd).to(q.dtype)
    lse = torch.where(torch.isinf(lse) & (lse < 0), torch.full_like(lse, float("inf")), lse)
    return o, lse


def attn_bwd_cross(q, k, v, o, do, lse, dq, dk, dv, B, Lq, Lk, H, Hkv, hd, kmask, scale):
    qh = q[:, : H * hd].reshape(B, Lq, H, hd).transpose(1, 2)
    kh = k[:, : Hkv * hd].reshape(B, Lk, Hkv, hd).transpose(1, 2)
    vh = v[:, : Hkv * hd].reshape(B, Lk, Hkv, hd).transpose(1, 2)
    rep = H // Hkv
    s = _scores(qh, kh, scale, False, kmask, H, Hkv)
    p = torch.exp(s - lse[..., None]).nan_to_num(0.0)
    dO = _f(do).reshape(B, Lq, H, hd).transpose(1, 2)
    O = _f(o).reshape(B, Lq, H, hd).transpose(1, 2)
    vv = _f(vh).repeat_interleave(rep, dim=1) if rep > 1 else _f(vh)
    kk = _f(kh).repeat_interleave(rep, dim=1) if rep > 1 else _f(kh)
    dsum = (dO * O).sum(-1, keepdim=True)
    dP = torch.matmul(dO, vv.transpose(-1, -2))
    dS = p * (dP - dsum) * scale
    gq = torch.matmul(dS, kk)
    gk = torch.matmul(dS.transpose(-1, -2), _f(qh))
    gv = torch.matmul(p.transpose(-1, -2), dO)
    if rep > 1:
        gk = gk.reshape(B, Hkv, rep, Lk, hd).sum(2)
        gv = gv.reshape(B, Hkv, rep, Lk, hd).sum(2)
    dq.copy_(gq.transpose(1, 2).reshape(B * Lq, H * hd).to(dq.dtype))
    dk.copy_(gk.transpose(1, 2).reshape(B * Lk, Hkv * hd).to(dk.dtype))
    dv.copy_(gv.transpose(1, 2).reshape(B * Lk, Hkv * hd).to(dv.dtype))


def attn_fwd_qkv(q, k, v, B, Lseq, H, Hkv, hd, kmask, scale, causal, want_lse=True, kstart=None, out=None):
    o, lse = attn_fwd(torch.cat([q[:, : H * hd], k[:, : Hkv * hd], v[:, : Hkv * hd]], dim=1), B, Lseq, H, Hkv, hd, kmask, scale, causal,
                      want_lse, kstart=kstart)
    if out is not None:
        out.copy_(o)
        o = out
    return o, lse


def attn_bwd_qkv(q, k, v, o, do, lse, dq, dk, dv, B, Lseq, H, Hkv, hd, kmask, scale, causal, kstart=None, qend=None):
    d = attn_bwd(torch.cat([q[:, : H * hd], k[:, : Hkv * hd], v[:, : Hkv * hd]], dim=1), o, do, lse, B, Lseq, H, Hkv, hd, kmask, scale, causal,
                 kstart=kstart)
    dq.copy_(d[:, : H * hd]), dk.copy_(d[:, H * hd: (H + Hkv) * hd]), dv.copy_(d[:, (H + Hkv) * hd:])


class PackPlan:
    pass


def pack_plan(input_ids, attention_mask, labels, num_patches, num_images, image_token_index, pad_token_id, ignore_index, L,
              fix_unequal_counts=False):
    ids = input_ids.numpy()
    pl = pack_ref.pack_plan(ids, attention_mask.numpy(), None if labels is None else labels.numpy(), num_images,
                            num_patches, image_token_index, pad_token_id, ignore_index, fix_unequal_counts=fix_unequal_counts)
    assert pl["L"] == L
    B, T = ids.shape
    out = PackPlan()
    out.B, out.T, out.L, out.N, out.I = B, T, L, num_patches, num_images
    kind, sidx = pl["src_kind"], pl["src_idx"]
    src = np.where(kind == pack_ref.TEXT, sidx, np.where(kind == pack_ref.IMAGE, sidx | (1 << 30), -1)).astype(np.int32)
    out.src = torch.from_numpy(src)
    out.attention_mask = torch.from_numpy(pl["attention_mask"])
    out.labels = torch.from_numpy(pl["labels"] if pl["labels"] is not None else np.full((B, L), ignore_index, np.int64))
    out.position_ids = torch.from_numpy(pl["position_ids"])
    out.kmask = (out.attention_mask != 0).to(torch.int32)
    m = ids == image_token_index
    tp = np.where(m, -1, pl["text_pos"]).astype(np.int32)
    out.text_pos = torch.from_numpy(tp)
    slot = np.full((max(1, num_images * num_patches),), -1, np.int32)
    bb, pp = np.nonzero(kind == pack_ref.IMAGE)
    slot[sidx[bb, pp]] = (bb * L + pp).astype(np.int32)
    out.img_slot = torch.from_numpy(slot)
    ce_row = np.full((B * T,), -1, np.int32)
    ce_tgt = np.full((B * T,), -100, np.int32)
    am = attention_mask.numpy()
    lab = None if labels is None else labels.numpy()
    for b in range(B):
        for t in range(T):
            if tp[b, t] >= 1:
                ce_row[b * T + t] = b * L + tp[b, t] - 1
                if lab is not None and am[b, t] != 0 and lab[b, t] != ignore_index:
                    ce_tgt[b * T + t] = lab[b, t]
    out.ce_row, out.ce_tgt = torch.from_numpy(ce_row), torch.from_numpy(ce_tgt)
    out.status = torch.zeros(4, dtype=torch.int32)
    out.kstart = out.qend = None
    return out


def pack_segments(plan, input_ids, segment_ids, image_token_index):
    pl = dict(L=plan.L, attention_mask=plan.attention_mask.numpy())
    ks, qe, pos, first = pack_ref.pack_segments(pl, input_ids.numpy(), segment_ids.numpy(), plan.N, image_token_index)
    plan.kstart, plan.qend = torch.from_numpy(ks), torch.from_numpy(qe)
    plan.position_ids = torch.from_numpy(pos)
    f = torch.from_numpy(first.reshape(-1))
    plan.ce_row[f] = -1
    plan.ce_tgt[f] = -100
    return plan


def pack_rows_fwd(plan, input_ids, embed_weight, image_features):
    d = embed_weight.shape[1]
    out = torch.zeros((plan.B * plan.L, d), dtype=embed_weight.dtype)
    src = plan.src.reshape(-1)
    rows = torch.arange(plan.B * plan.L)
    b = rows // plan.L
    txt = (src >= 0) & (src < (1 << 30))
    out[txt] = embed_weight[input_ids[b[txt], src[txt].long()]]
    im = src >= (1 << 30)
    if im.any():
        out[im] = image_features[(src[im] & ((1 << 30) - 1)).long()]
    return out


def gather_rows(x, idx):
    out = torch.zeros((idx.numel(), x.shape[1]), dtype=x.dtype)
    ok = idx >= 0
    out[ok] = x[idx[ok].long()]
    return out


def scatter_rows(x, idx, nrows_out, out=None):
    if out is None:
        out = torch.zeros((nrows_out, x.shape[1]), dtype=x.dtype)
    ok = idx >= 0
    out[idx[ok].long()] = x[ok]
    return out


def embed_grad(dmerged, input_ids, plan, grad_weight, accumulate):
    g = _f(grad_weight)
    tp = plan.text_pos.reshape(-1)
    ids = input_ids.reshape(-1)
    ok = tp >= 0
    if not accumulate:          # overwrite mode only touches the rows that occur
        g[ids[ok]] = 0
    b = torch.arange(ids.numel()) // plan.T
    rows = (b * plan.L + tp.long())[ok]
    g.index_add_(0, ids[ok], _f(dmerged)[rows])
    grad_weight.copy_(g.to(grad_weight.dtype))


def ce_fwd_bwd(logits, targets, V, grad_scale, loss_scale, write_grad=True):
    x = _f(logits)[:, :V]
    t = targets.long()
    valid = (t >= 0) & (t < V)          # t >= V: torch's CrossEntropyLoss raises; the product excludes and reports them
    cnt = int(valid.sum())
    lse = torch.logsumexp(x, dim=-1)
    picked = x.gather(1, t.clamp(min=0, max=V - 1)[:, None])[:, 0]
    row_loss = torch.where(valid, lse - picked, torch.zeros_like(lse))
    loss = loss_scale * row_loss.sum() / cnt if cnt else torch.tensor(float("nan"))
    if write_grad:
        g = torch.softmax(x, -1)
        g[torch.arange(x.shape[0])[valid], t[valid]] -= 1.0
        g = g * (grad_scale / cnt if cnt else float("inf"))
        g[~valid] = 0
        logits.zero_()
        logits[:, :V] = g.to(logits.dtype)
    return loss.reshape(1).float(), torch.tensor([cnt, int((t >= V).sum())], dtype=torch.int32)


def im2col(pixels, patch, kp):
    I, C, H, W = pixels.shape
    cols = F.unfold(pixels.float(), kernel_size=patch, stride=patch)       # [I, C*P*P, N]
    out = torch.zeros((I * cols.shape[2], kp), dtype=torch.bfloat16)
    out[:, : cols.shape[1]] = cols.transpose(1, 2).reshape(-1, cols.shape[1]).to(torch.bfloat16)
    return out


def cast_pad_rows(x, kp):
    out = torch.zeros((x.shape[0], kp), dtype=torch.bfloat16)
    out[:, : x.shape[1]] = x.to(torch.bfloat16)
    return out


def vit_assemble(patch_out, pos_emb, cls_emb, I, N):
    d = patch_out.shape[1]
    x = _f(patch_out).reshape(I, N, d)
    if cls_emb is not None:
        x = torch.cat([_f(cls_emb).expand(I, 1, d), x], dim=1)
    x = x + _f(pos_emb)[None]
    return x.reshape(-1, d).to(patch_out.dtype)


def navit_prepare(pixels, pixel_mask, patch, side, bucket):
    """Restates /root/reference/mantis/models/idefics2/modeling_idefics2.py:1636-1639 (padding images are all zero), :1653-1658 (pixel mask
    -> patch mask) and :190-210 (bucketised position ids) through oracle/idefics2_ref.py; `bucket` (the product's table) is NOT used."""
    from . import idefics2_ref as I2
    n = pixels.shape[0]
    nb = pixels.shape[1:].numel()
    real = ((pixels == 0.0).reshape(n, -1).sum(dim=1) != nb).to(torch.int32)
    pmask = torch.ones((n,) + tuple(pixels.shape[2:]), dtype=torch.bool) if pixel_mask is None else pixel_mask.bool()
    pm = I2.patch_mask_from_pixel_mask(pmask, patch)
    status = torch.zeros(n, dtype=torch.int32)
    pos = torch.zeros((n, pm.shape[1] * pm.shape[2]), dtype=torch.int32)
    for i in range(n):
        nh, nw = int(pm[i][:, 0].sum()), int(pm[i][0].sum())
        if int(pm[i].sum()) != nh * nw:
            status[i] = 1
            continue
        if nh * nw:
            pos[i] = I2.bucketized_position_ids(pm[i:i + 1], side)[0].to(torch.int32)
    return real, pm.reshape(n, -1).to(torch.int32), pos, status


def drop_cls(x, I, N):
    d = x.shape[1]
    return x.reshape(I, N + 1, d)[:, 1:].reshape(I * N, d).contiguous()


def adamw_flat(param, grad, master, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=None):
    g = _f(grad) * (float(grad_scale) if grad_scale is not None else 1.0)
    master.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    master.addcdiv_(m, (v / bc2).sqrt() + eps, value=-lr / bc1)
    param.copy_(master.to(param.dtype))


def _u16(x):
    return x.view(torch.int16).to(torch.int32) & 0xFFFF


def master_join(param, master_lo, v, out=None):
    """fp32 master <- (bf16 parameter = its round-to-nearest-even upper half, low 16 bits, tie bit in the sign of v): the storage of
    mantis_adamw_split (include/mantis_hip.h), restated with integer tensor arithmetic."""
    p, lo = _u16(param), _u16(master_lo)
    tie = v.view(torch.int32) < 0
    up = (lo > 0x8000) | ((lo == 0x8000) & tie)
    hi = (p - up.to(torch.int32)) & 0xFFFF
    bits = (hi.to(torch.int64) << 16) | lo.to(torch.int64)
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)
    res = bits.view(torch.float32)
    if out is not None:
        out.copy_(res)
        return out
    return res


def master_split(master, param, master_lo, v):
    bits = master.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    lo, hi = bits & 0xFFFF, bits >> 16
    param.copy_(master.to(param.dtype))
    lo16 = torch.where(lo >= 2 ** 15, lo - 2 ** 16, lo).to(torch.int16)
    master_lo.copy_(lo16.view(master_lo.dtype) if master_lo.dtype != torch.int16 else lo16)
    tie = (lo == 0x8000) & ((hi & 1) == 1)
    vb = v.view(torch.int32)
    vb.copy_(torch.where(tie, vb | -2 ** 31, vb & 0x7FFFFFFF))


def adamw_split_flat(param, grad, master_lo, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=None):
    """AdamW on the split master: join, the fp32 update of adamw_flat, split."""
    master = master_join(param, master_lo, v).clone()
    v.copy_(v.abs())
    adamw_flat(param, grad, master, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=grad_scale)
    master_split(master, param, master_lo, v)


def grad_sumsq(x, out, accumulate=False, ws=None):
    s = _f(x).pow(2).sum()
    out[0] = out[0] + s if accumulate else s


def sumsq_ranges(x, off_len, partials):
    for r, (o, n) in enumerate(off_len.tolist()):
        partials[r] = _f(x[o:o + n]).pow(2).sum()


def sum_f32(x, out, accumulate=False):
    s = x.float().sum()
    out[0] = out[0] + s if accumulate else s


def clip_scale(sumsq, max_norm):
    norm = sumsq.sqrt()
    return torch.clamp(max_norm / (norm + 1e-6), max=1.0).reshape(1), norm.reshape(1)


def synchronize():
    pass
