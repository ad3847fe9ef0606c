#!/usr/bin/env python3
"""GEMM micro-benchmark on the GPU box: every tile variant x the shapes of the Mantis-8B step (HIP-event timing)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402

SHAPES = {  # name: (M, N, K)
    "qkv_fwd": (5624, 6144, 4096), "o_fwd": (5624, 4096, 4096), "gu_fwd": (5624, 28672, 4096), "down_fwd": (5624, 4096, 14336),
    "dx_gu": (5624, 4096, 28672), "dx_down": (5624, 14336, 4096), "dw_gu": (28672, 4096, 5624), "dw_down": (4096, 14336, 5624),
    "dw_qkv": (6144, 4096, 5624), "dw_o": (4096, 4096, 5624), "vit_fc1": (4608, 4304, 1152), "vit_qkv": (4608, 3456, 1152),
    "lm_head": (1024, 128258, 4096), "sq4096": (4096, 4096, 4096), "sq8192": (8192, 8192, 8192),
}


def main():
    for spec in filter(None, os.environ.get("GEMM_BENCH_SHAPES", "").split(";")):      # extra shapes "name:M,N,K;..."
        nm, dims = spec.split(":")
        SHAPES[nm] = tuple(int(x) for x in dims.split(","))
    variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 12]
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(SHAPES)
    res = {}
    for name in names:
        M, N, Kd = SHAPES[name]
        a = torch.randn(M, Kd, device="cuda", dtype=torch.bfloat16)
        b = torch.randn(N, Kd, device="cuda", dtype=torch.bfloat16) * 0.05
        ref = None
        row = {}
        for v in variants:
            out = K.gemm_nt(a, b, variant=v)
            torch.cuda.synchronize()
            if ref is None:
                ref = out
                r0 = (a[:64].float() @ b[:256].float().t())
                err = float((out[:64, :256].float() - r0).norm() / r0.norm())
                assert err < 1e-2, (name, v, err)
            else:
                assert v >= 8 or torch.equal(out, ref), (name, v, "variants disagree")
            for _ in range(2):
                K.gemm_nt(a, b, out=out, variant=v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            it = 8
            for _ in range(it):
                K.gemm_nt(a, b, out=out, variant=v)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / it
            row[v] = round(2.0 * M * N * Kd / ms / 1e9, 1)
            row[f"us{v}"] = round(ms * 1e3, 1)
        if os.environ.get("GEMM_BENCH_VENDOR") == "1":     # context only: the vendor library (hipBLASLt via torch) on the same operands
            bt = b.t()
            for _ in range(2):
                torch.mm(a, bt)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                torch.mm(a, bt)
            e1.record()
            torch.cuda.synchronize()
            row["vendor"] = round(2.0 * M * N * Kd / (e0.elapsed_time(e1) / 8) / 1e9, 1)
        res[name] = row
        print(f"{name:10s} {M}x{N}x{Kd}: " + "  ".join(f"{('v' + str(v)) if not isinstance(v, str) else v}={t:7.1f}{'' if str(v).startswith('us') else 'TF'}" for v, t in row.items()), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
