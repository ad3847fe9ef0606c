#!/usr/bin/env python3
"""How long does the fused SwiGLU-backward epilogue of dX(down_proj) take when FEW compute units run it (no HBM contention) against all of
them in lock step?  Times the 176-row kernel (variant 15, persistent) on the headline shape 5624 x 14336 x 4096 (B K-major) with a plain
epilogue and with the fused SwiGLU backward, planned for 8 ... 256 CUs: per-tile time = launch time x CUs / tiles; the difference between the
two epilogues is the epilogue's own cost per tile at that level of contention.  -> profiles/r06_experiments.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def timed(fn, it=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def main():
    M, I, d = 5624, 14336, 4096
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    dy = torch.randn(M, d, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(d, I, device="cuda", dtype=torch.bfloat16) * 0.02
    gu = torch.randn(M, 2 * I, device="cuda", dtype=torch.bfloat16)
    tiles = -(-M // (176 if variant == 15 else 256)) * (I // 256)
    print(f"variant {variant}: {tiles} tiles")
    for cus in (8, 16, 32, 64, 128, 192, 256):
        with K.launch_context(K.LaunchContext(gemm_cus=cus)):
            it = 2 if cus <= 16 else 5
            t_plain = timed(lambda: K.gemm_nt(dy, w, b_kmajor=True, variant=variant), it)
            t_sw = timed(lambda: K.linear_dx_swiglu(dy, w, gu, variant=variant), it)
        per = lambda t: t * cus / tiles
        print(f"cus={cus:4d}  plain {t_plain:9.1f} us ({per(t_plain):6.1f} us/tile)   swiglu_bwd {t_sw:9.1f} us ({per(t_sw):6.1f} us/tile)   "
              f"epilogue delta {per(t_sw) - per(t_plain):6.1f} us/tile")


if __name__ == "__main__":
    main()
