#!/usr/bin/env python3
"""Anatomy of a ring16 GEMM launch from in-kernel time stamps (probe build only: tools/build_probe_lib.sh gemm stamps -DRING16_STAMPS;
run with MANTIS_HIP_LIB=tools/_bin/libmantis_stamps.so).  Per workgroup: s_memtime at entry, after the prologue's DMA issue, at the
first MFMA (first K-step landed + barrier), at the end of the K loop and at exit, plus the CU it ran on.  Prints, per shape: the phases
(median over workgroups, microseconds), the gap between one workgroup's exit and the next one's entry on the SAME CU (dispatch cost
of a tile round), and how the launch's wall time splits into rounds.  -> profiles/r04_gemm_anatomy.md"""
import ctypes
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mantis_amd  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def med(x):
    return float(np.median(x)) if len(x) else float("nan")


def main():
    L = K._L
    try:
        fn = L.mantis_probe_ring16_stamps
    except AttributeError:
        raise SystemExit("needs a -DRING16_STAMPS build: MANTIS_HIP_LIB=tools/_bin/libmantis_stamps.so")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    T, d, I, QKV = 5624, 4096, 14336, 6144
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device="cuda") * sc).to(torch.bfloat16)
    x, xi, x2i, xq = rn(T, d), rn(T, I), rn(T, 2 * I), rn(T, QKV)
    w = dict(qkv=rn(QKV, d, sc=0.02), o=rn(d, d, sc=0.02), gu=rn(2 * I, d, sc=0.02), down=rn(d, I, sc=0.02))
    cos, sin = rn(T, 64), rn(T, 64)
    gw = {k: torch.empty_like(v) for k, v in w.items()}
    xv, hv = rn(4608, 1152), rn(4608, 4304)
    tw = dict(fc1=rn(4304, 1152, sc=0.02), fc2=rn(1152, 4304, sc=0.02), b1=rn(4304), b2=rn(1152), qkv=rn(3456, 1152, sc=0.02), bq=rn(3456))
    cases = [
        ("o fwd NT+res 5624x4096x4096", lambda: K.gemm_nt(x, w["o"], residual=x), 352),
        ("qkv fwd NT+rope 5624x6144x4096", lambda: K.linear_qkv_rope(x, w["qkv"], None, cos, sin, 40, 128), 528),
        ("gate|up fwd NT+swiglu 5624x28672x4096", lambda: K.linear_gu_swiglu(x, w["gu"]), 2464),
        ("down fwd NT+res 5624x4096x14336", lambda: K.gemm_nt(xi, w["down"], residual=x), 352),
        ("dX(down)+swiglu_bwd NN 5624x14336x4096", lambda: K.linear_dx_swiglu(x, w["down"], x2i), 1232),
        ("dX(gate|up) NN 5624x4096x28672", lambda: K.linear_dx(x2i, w["gu"]), 352),
        ("dW(gate|up) TN 28672x4096x5624", lambda: K.linear_dw(x2i, x, gw["gu"], False), 1792),
        ("dW(o) TN 4096x4096x5624", lambda: K.linear_dw(x, x, gw["o"], False), 256),
        ("tower fc1 NT 4608x4304x1152", lambda: K.gemm_nt(xv, tw["fc1"], bias=tw["b1"], act="gelu_pytorch_tanh"), 306),
        ("tower fc2 NT 4608x1152x4304", lambda: K.gemm_nt(hv, tw["fc2"], bias=tw["b2"], residual=xv), 90),
        ("tower qkv NT 4608x3456x1152", lambda: K.gemm_nt(xv, tw["qkv"], bias=tw["bq"]), 252),
    ]
    print("| GEMM | launch us | WGs | tick ns | prologue issue | first data + barrier | K loop | epilogue (+K-split reduce) | exit->entry gap on a CU | "
          "first WG entry spread | rounds (WGs per CU max) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    raw = []
    for name, fnc, tiles in cases:
        for _ in range(3):
            fnc()
        torch.cuda.synchronize()
        nwg = 8192
        before = np.zeros((nwg, 8), dtype=np.uint64)
        assert fn(before.ctypes.data, nwg) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fnc()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        buf = np.zeros((nwg, 8), dtype=np.uint64)
        assert fn(buf.ctypes.data, nwg) == 0
        fresh = (buf[:, 6] != before[:, 6]) & (buf[:, 6] > 0)      # rows this launch rewrote (stale rows of earlier, larger grids drop out)
        b = buf[fresh].astype(np.int64)
        raw.append((name, us, b))
        t0, t1, t2, t3, t6 = b[:, 2], b[:, 3], b[:, 4], b[:, 5], b[:, 6]
        span = t6.max() - t0.min()
        tick_ns = us * 1e3 / span                        # calibrate the s_memtime tick against the HIP-event duration (approximate)
        tu = lambda v: v * tick_ns / 1e3
        cu = (b[:, 0] << 32) | (b[:, 1] & 0xFFF00)       # XCC id + (SE, SH, CU) bits of HW_ID (wave / SIMD / pipe bits dropped)
        gaps, per_cu = [], defaultdict(list)
        for i in range(len(b)):
            per_cu[int(cu[i])].append((int(t0[i]), int(t6[i])))
        for v in per_cu.values():
            v.sort()
            for (a0, a1), (b0, b1) in zip(v, v[1:]):
                gaps.append(b0 - a1)
        first = sorted(v[0][0] for v in per_cu.values())
        spread = first[-1] - first[0] if first else 0
        print(f"| {name} | {us:.1f} | {len(b)} | {tick_ns:.2f} | {tu(med(t1 - t0)):.2f} | {tu(med(t2 - t1)):.2f} | {tu(med(t3 - t2)):.2f} | "
              f"{tu(med(t6 - t3)):.2f} (p90 {tu(float(np.percentile(t6 - t3, 90))):.2f}) | {tu(med(gaps)):.2f} (p90 {tu(float(np.percentile(gaps, 90))) if gaps else float('nan'):.2f}, n={len(gaps)}) | "
              f"{tu(spread):.2f} | {len(per_cu)} CUs, max {max(len(v) for v in per_cu.values())} |", flush=True)
    out = os.environ.get("ANATOMY_RAW")
    if out:
        np.savez_compressed(out, **{f"c{i}": b for i, (_, _, b) in enumerate(raw)}, names=np.array([n for n, _, _ in raw]),
                            us=np.array([u for _, u, _ in raw]))


if __name__ == "__main__":
    main()
