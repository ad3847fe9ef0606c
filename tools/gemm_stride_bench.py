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
        # the same GEMM with the weight (and, separately, the activation) starting 96 bytes into a 128-byte line, as every decoder weight
        # of the 16-byte aligned parameter arena did (profiles/r04_experiments.md 12)
        line = f"{name:12s} base offset:"
        a0 = torch.randn(M * Kk + 64, device=dev, generator=g).to(torch.bfloat16)
        b0 = (torch.randn(N * Kk + 64, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        for oa in (0, 48):
            for ob in (0, 48):
                a = a0[oa: oa + M * Kk].view(M, Kk)
                b = b0[ob: ob + N * Kk].view(N, Kk)
                t = timeit(lambda: K.gemm_nt(a, b))
                line += f"  offA={2 * oa:2d}B offB={2 * ob:2d}B: {1e3 * t:7.1f} us |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
