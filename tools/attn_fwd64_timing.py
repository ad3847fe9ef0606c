#!/usr/bin/env python3
"""Read the s_memtime stamps a FWD64_PROBE_TIMING build of attn_fwd64 leaves in the LSE buffer (Llama-3 step geometry):
per wave prologue / tile loop / epilogue cycles and the tile count.
    tools/build_probe_lib.sh attn_fwd64 f64_timing -DFWD64_PROBE_TIMING
    MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_f64_timing.so MANTIS_ATTN_FWD64=1 python tools/attn_fwd64_timing.py [mask|nomask]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "nomask"
    B, L, H, Hkv, hd = 2, 2812, 32, 8, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * L, (H + 2 * Hkv) * hd, generator=g, device="cuda").to(torch.bfloat16)
    kmask = torch.ones(B, L, dtype=torch.int32, device="cuda") if mode == "mask" else None
    for _ in range(3):
        o, lse = K.attn_fwd(qkv, B, L, H, Hkv, hd, kmask, hd ** -0.5, True)
    torch.cuda.synchronize()
    lse = lse.cpu()
    rows = []
    for q0 in range(0, L - 3, 64):
        w = lse[:, :, q0:q0 + 4].reshape(-1, 4)
        rows.append((q0, w.mean(0), w.max(0).values))
    print(f"[{mode}] wave rows q0: prologue / loop / epilogue cycles (mean over {B * H} heads), WG tiles, loop cycles per tile")
    for q0, m, mx in rows:
        nt = float(m[3])
        print(f"  q0={q0:5d}  pro {float(m[0]):8.0f}  loop {float(m[1]):9.0f}  epi {float(m[2]):7.0f}  tiles {nt:4.0f}  loop/tile {float(m[1]) / max(nt, 1):7.0f}"
              f"   (max pro {float(mx[0]):8.0f} loop {float(mx[1]):9.0f} epi {float(mx[2]):7.0f})")


if __name__ == "__main__":
    main()
