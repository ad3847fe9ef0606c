#!/usr/bin/env python3
"""Attention micro-benchmark on the GPU box at the Mantis-8B step shape (B=2, L=2812, 32/8 heads x 128, causal + key mask):
HIP-event time of the forward and of the whole backward (dsum + dQ + dK/dV + group reduce).  `python tools/attn_bench.py [iters]`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def main():
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    mode = sys.argv[2] if len(sys.argv) > 2 else "mask"        # mask | nomask: with / without the (all-ones) key mask
    B, L, H, Hkv, hd = 2, 2812, 32, 8, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * L, (H + 2 * Hkv) * hd, generator=g, device="cuda").to(torch.bfloat16)
    do = torch.randn(B * L, H * hd, generator=g, device="cuda").to(torch.bfloat16)
    kmask = torch.ones(B, L, dtype=torch.int32, device="cuda") if mode == "mask" else None
    scale = hd ** -0.5
    o, lse = K.attn_fwd(qkv, B, L, H, Hkv, hd, kmask, scale, True)
    K.attn_bwd(qkv, o, do, lse, B, L, H, Hkv, hd, kmask, scale, True)
    torch.cuda.synchronize()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    fl = 4.0 * L * L * hd / 2 * B * H          # causal forward FLOPs
    for _ in range(5):
        K.attn_fwd(qkv, B, L, H, Hkv, hd, kmask, scale, True)
    tf = timed(lambda: K.attn_fwd(qkv, B, L, H, Hkv, hd, kmask, scale, True))
    tf_nolse = timed(lambda: K.attn_fwd(qkv, B, L, H, Hkv, hd, kmask, scale, True, want_lse=False))
    tb = timed(lambda: K.attn_bwd(qkv, o, do, lse, B, L, H, Hkv, hd, kmask, scale, True))
    print(f"[{mode}] attn fwd {tf:7.1f} us = {fl / tf / 1e6:6.0f} TF (no lse {tf_nolse:7.1f} us) | bwd (dq incl. rowsum(dO*O) + dkv + reduce) {tb:7.1f} us = {2.5 * fl / tb / 1e6:6.0f} TF")


if __name__ == "__main__":
    main()
