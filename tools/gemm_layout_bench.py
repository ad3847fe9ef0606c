#!/usr/bin/env python3
"""Ring-GEMM layouts on the MLP shapes of the Mantis-8B step, random operands: NT (forward), NN (dX: weight K-major as stored), TN (dW:
both activations K-major as stored).  Run plain for HIP-event TFLOP/s, or under `rocprofv3 --pmc ...` + tools/pmc_kernel.py for the
per-instantiation counter split.   python tools/gemm_layout_bench.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def main():
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    T, d, I = 5624, 4096, 14336
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)
    x, w_gu, dgu, w_o, dx = rn(T, d), rn(2 * I, d) * 0.05, rn(T, 2 * I), rn(d, d) * 0.05, rn(T, d)
    cases = {
        "NT gu_fwd  5624x28672x4096": lambda: K.gemm_nt(x, w_gu),
        "NN dx_gu   5624x4096x28672": lambda: K.gemm_nt(dgu, w_gu, b_kmajor=True, k=w_gu.shape[0]),
        "TN dw_gu   28672x4096x5624": lambda: K.gemm_nt(dgu, x, a_kmajor=True, b_kmajor=True),
        "NT o_fwd   5624x4096x4096": lambda: K.gemm_nt(x, w_o),
        "NN dx_o    5624x4096x4096": lambda: K.gemm_nt(dx, w_o, b_kmajor=True, k=d),
        "TN dw_o    4096x4096x5624": lambda: K.gemm_nt(dx, x, a_kmajor=True, b_kmajor=True),
    }
    fl = {"gu": 2.0 * T * 2 * I * d, "o": 2.0 * T * d * d}
    for name, fn in cases.items():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        f = fl["gu" if "gu" in name else "o"]
        print(f"{name}: {ms * 1e3:8.1f} us  {f / ms / 1e9:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
