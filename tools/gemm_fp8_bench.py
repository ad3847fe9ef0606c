#!/usr/bin/env python3
"""Micro-benchmark: fp8 MFMA GEMM (csrc/gemm_fp8.hip) vs the bf16 GEMM family on the Qwen2-VL-7B decoder's linear shapes (4096 tokens),
plus the quantiser's bandwidth.  Random operands (the matrix pipe's power roof depends on the data, profiles/r02_experiments.md).
Usage (GPU box): python tools/gemm_fp8_bench.py [--variants 1,2]"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="1,2")
    ap.add_argument("--tokens", type=int, default=4096)
    args = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    from mantis_amd import hip_ops as K
    dev = "cuda"
    T, d, I, QKV = args.tokens, 3584, 18944, 4608
    shapes = [("qkv fwd", T, QKV, d), ("o fwd / dX", T, d, d), ("gate|up fwd", T, 2 * I, d), ("down fwd", T, d, I),
              ("dX gate|up", T, d, 2 * I), ("dX down", T, I, d), ("dW qkv", QKV, d, T), ("dW o", d, d, T), ("dW gate|up", 2 * I, d, T),
              ("dW down", d, I, T), ("square 8192", 8192, 8192, 8192)]
    g = torch.Generator(device=dev).manual_seed(0)
    print(f"{'shape':14s} {'M':>6s} {'N':>6s} {'K':>6s} | bf16 us   TF  |" + "".join(f" fp8 v{v} us   TF  |" for v in args.variants.split(",")))
    for name, M, N, Kk in shapes:
        a = torch.randn(M, Kk, device=dev, generator=g).to(torch.bfloat16)
        b = (torch.randn(N, Kk, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        fl = 2.0 * M * N * Kk
        t = timeit(lambda: K.gemm_nt(a, b))
        line = f"{name:14s} {M:6d} {N:6d} {Kk:6d} | {1e3 * t:7.1f} {fl / t / 1e9:6.0f} |"
        aq, bq = K.fp8_quantize(a, 0, transposed=False), K.fp8_quantize(b, 0, transposed=False)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for v in args.variants.split(","):
            t8 = timeit(lambda: K.gemm_fp8_nt(aq.q, aq.dequant, bq.q, bq.dequant, 0, out=out, variant=int(v)))
            line += f" {1e3 * t8:8.1f} {fl / t8 / 1e9:6.0f} |"
        print(line, flush=True)
    for (r, c) in [(T, d), (T, I), (T, 2 * I), (2 * I, d)]:
        x = torch.randn(r, c, device=dev, generator=g).to(torch.bfloat16)
        for tr in (False, True):
            t = timeit(lambda: K.fp8_quantize(x, 0, transposed=tr))
            by = r * c * (2 + 2 + 1 + (1 if tr else 0))
            print(f"quantize {r}x{c} transposed={tr}: {1e3 * t:7.1f} us  {by / t / 1e9:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
