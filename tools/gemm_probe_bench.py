#!/usr/bin/env python3
"""Timing of the ring GEMM variants on a few shapes WITHOUT result checks: for the timing-probe builds of the library (MANTIS_HIP_LIB=
tools/_bin/libmantis_NO_DMA.so ...: the loop without its DMA issues / fragment reads / K-step barrier; results are wrong by construction).
usage: python tools/gemm_probe_bench.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def main():
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device="cuda") * sc).to(torch.bfloat16)
    shapes = {"gu_fwd NT 5624x28672x4096": (5624, 28672, 4096, False, False), "sq8192 NT": (8192, 8192, 8192, False, False),
              "dx_gu NN 5624x4096x28672": (5624, 4096, 28672, False, True), "dw_gu TN 28672x4096x5624": (28672, 4096, 5624, True, True)}
    for name, (M, N, Kd, akm, bkm) in shapes.items():
        a = rn(Kd, M) if akm else rn(M, Kd)
        b = rn(Kd, N, sc=0.02) if bkm else rn(N, Kd, sc=0.02)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        line = f"{name:28s}"
        for v in (12, 13, 14):
            fn = lambda: K.gemm_nt(a, b, out=out, a_kmajor=akm, b_kmajor=bkm, variant=v)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / it * 1e3
            line += f" | v{v} {us:8.1f} us {2.0 * M * N * Kd / us / 1e6:7.1f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
