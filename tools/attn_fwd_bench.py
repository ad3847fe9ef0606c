#!/usr/bin/env python3
"""Micro-benchmark of the attention forward kernel on the vision towers' shapes (non-causal) and the decoders' (causal).
Usage (GPU box): python tools/attn_fwd_bench.py      (MANTIS_HIP_LIB=<other .so> for A/B on one box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    from mantis_amd import hip_ops as K
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    cases = [("Qwen2-VL tower  hd 80", 2, 6256, 16, 16, 80, False), ("SigLIP-336 tower hd 72", 8, 576, 16, 16, 72, False),
             ("SigLIP-so400m 448 hd 72", 16, 1024, 16, 16, 72, False), ("CLIP-L/14-336  hd 64", 8, 577, 16, 16, 64, False),
             ("perceiver      hd 96", 16, 1088, 16, 4, 96, False), ("Llama-3 8B     hd 128", 2, 2812, 32, 8, 128, True),
             ("Qwen2-7B       hd 128", 1, 4096, 28, 4, 128, True)]
    for name, B, L, H, Hkv, hd, causal in cases:
        qkv = torch.randn(B * L, (H + 2 * Hkv) * hd, device=dev, generator=g).to(torch.bfloat16)
        t = timeit(lambda: K.attn_fwd(qkv, B, L, H, Hkv, hd, None, hd ** -0.5, causal, want_lse=False))
        fl = 4.0 * B * L * L * H * hd * (0.5 if causal else 1.0)
        print(f"{name:26s} B={B:2d} L={L:5d} H={H:2d}/{Hkv:2d}: {1e3 * t:8.1f} us  {fl / t / 1e9:6.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
