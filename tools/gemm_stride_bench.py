#!/usr/bin/env python3
"""Experiment: does the power-of-two row stride of the NT operands (K = 4096 bf16 = 8 KiB) cost the bf16 ring GEMM bandwidth (L2 / fabric
channel conflicts)?  Times C = A . B^T on the headline step's forward shapes with the row stride of A and / or B padded by 128 B.
Usage (GPU box): python tools/gemm_stride_bench.py"""
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
    import __graft_entry__
    __graft_entry__.build()
    from mantis_amd import hip_ops as K
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    M = 5624
    for name, N, Kk in [("qkv fwd", 6144, 4096), ("o fwd", 4096, 4096), ("gate|up fwd", 28672, 4096), ("down fwd", 4096, 14336)]:
        line = f"{name:12s} {M}x{N}x{Kk}:"
        for pa in (0, 64):
            for pb in (0, 64):
                a = torch.randn(M, Kk + pa, device=dev, generator=g).to(torch.bfloat16)[:, :Kk]
                b = (torch.randn(N, Kk + pb, device=dev, generator=g) * 0.05).to(torch.bfloat16)[:, :Kk]
                t = timeit(lambda: K.gemm_nt(a, b))
                line += f"  padA={pa:2d} padB={pb:2d}: {1e3 * t:7.1f} us {2.0 * M * N * Kk / t / 1e9:5.0f} TF |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
