#!/usr/bin/env python3
"""Where a pipelined step of `attn_bwd_dkv_g4_kernel` (32 keys x 32 queries x one head per wave: 32 MFMAs) spends its time: in-kernel
`s_memtime` sums of a probe build.

    tools/build_probe_lib.sh attn dkvstamps -DDKV_STAMPS        (build container)
    MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_dkvstamps.so python tools/attn_dkv_anatomy.py     (GPU box)

Wave 0 of every workgroup sums the intervals step top -> (DMA of tile j + 3 issued, lse / dsum words and the first four row fragments
requested) -> MFMA 15 (S, dP of tile j + 2 || softmax backward of j + 1, first half) -> MFMA 31 (dV, dK of tile j || second half) -> behind
the step's barrier (stage_finish, vmcnt(0), s_barrier).  Llama-3 step geometry (B 2, L 2812, 32 / 8 heads x 128, causal, no key mask);
32 MFMAs of 32 cycles = 1024 cycles per step is the floor.  Every stamp drains the LDS reads in flight: read the PROPORTIONS."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def main():
    B, L, H, Hkv, hd = 2, 2812, 32, 8, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * L, (H + 2 * Hkv) * hd, generator=g, device="cuda").to(torch.bfloat16)
    do = torch.randn(B * L, H * hd, generator=g, device="cuda").to(torch.bfloat16)
    scale = hd ** -0.5
    o, lse = K.attn_fwd(qkv, B, L, H, Hkv, hd, None, scale, True)
    for _ in range(3):
        K.attn_bwd(qkv, o, do, lse, B, L, H, Hkv, hd, None, scale, True)
    torch.cuda.synchronize()
    n_wg = -(-L // 64) * Hkv * B
    buf = np.zeros((n_wg, 8), dtype=np.uint64)
    fn = K._L.mantis_probe_dkv_stamps
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    if fn(buf.ctypes.data, n_wg) != 0:
        sys.exit("mantis_probe_dkv_stamps failed (is MANTIS_HIP_LIB the -DDKV_STAMPS probe build?)")
    s = buf.astype(np.float64)
    n = s[:, 2]
    ok = n > 0
    names = ["step top: DMA issue of tile j+3, words, first fragments", "MFMA 0-15: S, dP of j+2 || softmax-bwd j+1 (first half)",
             "MFMA 16-31: dV, dK of j || softmax-bwd j+1 (second half)", "stage_finish + vmcnt(0) + barrier"]
    per = [s[ok, 3 + k] / n[ok] for k in range(4)]
    tot = sum(per)
    print(f"{n_wg} workgroups, {int(n.sum())} pipelined wave-0 steps")
    print("| interval | ticks per step, median | p10 | p90 | share |")
    print("|---|---|---|---|---|")
    for nm, v in zip(names, per):
        print(f"| {nm} | {np.median(v):.0f} | {np.percentile(v, 10):.0f} | {np.percentile(v, 90):.0f} | {np.median(v / tot) * 100:.0f} % |")
    print(f"| step total | {np.median(tot):.0f} | {np.percentile(tot, 10):.0f} | {np.percentile(tot, 90):.0f} | 100 % |")
    inloop = s[ok, 3:7].sum(axis=1)
    print(f"entry -> loop start median {np.median(s[ok, 0]):.0f} ticks (K / V fragments, masked diagonal tiles, pipeline prologue); whole workgroup median "
          f"{np.median(s[:, 1]):.0f}, of which in pipelined steps {np.median(inloop / s[ok, 1]) * 100:.0f} %; steps per workgroup {int(n[ok].min())} .. {int(n.max())}")
    # by key block (heaviest first: block 0 sees every query tile)
    bx = s[:, 7]
    for lo, hi in ((0, 4), (20, 24), (40, 44)):
        sel = ok & (bx >= lo) & (bx < hi)
        if sel.any():
            print(f"key blocks {lo}-{hi - 1}: whole workgroup median {np.median(s[sel, 1]):.0f} ticks, steps {np.median(n[sel]):.0f}, ticks per step {np.median(tot[sel[ok]]):.0f}")


if __name__ == "__main__":
    main()
