#!/usr/bin/env python3
"""Where a 64-key tile of `attn_bwd_dq64_kernel` spends its time: in-kernel `s_memtime` sums of a probe build.

    tools/build_probe_lib.sh attn_dq64 dq64stamps -DDQ64_STAMPS        (build container)
    MANTIS_ATTN_DQ64=1 MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_dq64stamps.so python tools/attn_dq64_anatomy.py     (GPU box)

Wave 0 of every workgroup sums, over its UNMASKED tiles, the intervals tile top -> MFMA 31 (S, dP of keys 0-31) -> MFMA 63 (S, dP of
keys 32-63 + softmax backward of half 0) -> MFMA 79 (dQ from half 0 + softmax backward of half 1) -> MFMA 95 (dQ from half 1) -> behind
the tile's barrier, plus entry -> loop start and entry -> exit.  The counter is the one tools/gemm_anatomy.py uses (per CU; only
differences on one CU mean anything); every stamp drains the LDS reads in flight, so read the PROPORTIONS.  Llama-3 step geometry
(B 2, L 2812, 32 / 8 heads x 128, causal, no key mask); 96 MFMAs of 32 cycles each = 3072 cycles per tile is the floor."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def main():
    if os.environ.get("MANTIS_ATTN_DQ64") != "1":
        sys.exit("run with MANTIS_ATTN_DQ64=1 (and MANTIS_HIP_LIB = the -DDQ64_STAMPS probe library)")
    B, L, H, Hkv, hd = 2, 2812, 32, 8, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * L, (H + 2 * Hkv) * hd, generator=g, device="cuda").to(torch.bfloat16)
    do = torch.randn(B * L, H * hd, generator=g, device="cuda").to(torch.bfloat16)
    scale = hd ** -0.5
    o, lse = K.attn_fwd(qkv, B, L, H, Hkv, hd, None, scale, True)
    for _ in range(3):
        K.attn_bwd(qkv, o, do, lse, B, L, H, Hkv, hd, None, scale, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.attn_bwd(qkv, o, do, lse, B, L, H, Hkv, hd, None, scale, True)
    e1.record()
    torch.cuda.synchronize()
    n_wg = -(-L // 256) * H * B
    buf = np.zeros((n_wg, 12), dtype=np.uint64)
    fn = K._L.mantis_probe_dq64_stamps
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    if fn(buf.ctypes.data, n_wg) != 0:
        sys.exit("mantis_probe_dq64_stamps failed (is MANTIS_HIP_LIB the probe build?)")
    s = buf.astype(np.float64)
    n = s[:, 2]
    ok = n > 0
    print(f"whole backward (dQ + dK/dV) {e0.elapsed_time(e1) * 1e3:.1f} us; {n_wg} workgroups, {int(n.sum())} unmasked + {int(s[:, 3].sum())} masked "
          f"+ {int(s[:, 4].sum())} skipped wave-0 tiles")
    names = ["S,dP half 0 (MFMA 0-31)", "S,dP half 1 + softmax-bwd half 0 (32-63)", "dQ half 0 + softmax-bwd half 1 (64-79)", "dQ half 1 (80-95)",
             "ring wait + barrier"]
    per = [s[ok, 5 + k] / n[ok] for k in range(5)]
    tot = sum(per)
    print("| interval | ticks per tile, median | p10 | p90 | share |")
    print("|---|---|---|---|---|")
    for nm, v in zip(names, per):
        print(f"| {nm} | {np.median(v):.0f} | {np.percentile(v, 10):.0f} | {np.percentile(v, 90):.0f} | {np.median(v / tot) * 100:.0f} % |")
    print(f"| tile total | {np.median(tot):.0f} | {np.percentile(tot, 10):.0f} | {np.percentile(tot, 90):.0f} | 100 % |")
    print(f"prologue (entry -> loop start) median {np.median(s[:, 0]):.0f} ticks; whole workgroup median {np.median(s[:, 1]):.0f} ticks "
          f"(tiles per workgroup 4 .. 44)")


if __name__ == "__main__":
    main()
