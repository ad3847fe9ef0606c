#!/usr/bin/env python3
"""Schedule-efficiency probe for the GEMM kernels: zero-filled operands take the power limit out of the picture (MI355X clocks
down on random data), so TFLOP/s on zeros measures how well the instruction schedule feeds the MFMA pipe; random data beside it.
usage: python tools/gemm_sched_bench.py [variant ...]   (default 12 1)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def tf(a, b, v, akm=False, bkm=False):
    M = a.shape[1] if akm else a.shape[0]
    N = b.shape[1] if bkm else b.shape[0]
    Kd = a.shape[0] if akm else a.shape[1]
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        K.gemm_nt(a, b, out=out, variant=v, a_kmajor=akm, b_kmajor=bkm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.gemm_nt(a, b, out=out, variant=v, a_kmajor=akm, b_kmajor=bkm)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * M * N * Kd / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12


def main():
    variants = [int(x) for x in sys.argv[1:]] or [12, 1]
    for (M, N, Kd) in ((8192, 8192, 8192), (4096, 4096, 4096), (5624, 28672, 4096)):
        for lay, akm, bkm in (("NT", False, False), ("NN", False, True), ("TN", True, True)):
            row = f"{M}x{N}x{Kd} {lay}:"
            for fill in ("zeros", "randn"):
                mk = (lambda *s: torch.zeros(*s, device="cuda", dtype=torch.bfloat16)) if fill == "zeros" else \
                     (lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16))
                a = mk(Kd, M) if akm else mk(M, Kd)
                b = mk(Kd, N) if bkm else mk(N, Kd)
                for v in variants:
                    row += f"  {fill} v{v} {tf(a, b, v, akm, bkm):6.0f}"
            print(row, flush=True)


if __name__ == "__main__":
    main()
