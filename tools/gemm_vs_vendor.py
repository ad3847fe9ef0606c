#!/usr/bin/env python3
"""Per-shape A/B of this library's bf16 GEMM against the vendor library (hipBLASLt through torch.mm) on the fifteen GEMMs of one
Mantis-8B decoder layer + head, each in the operand layout the step really uses (NT forward, NN dX with the weight K-major as stored,
TN dW with both activations K-major as stored).  CONTEXT ONLY: the vendor library is never on the product path.  Prints a markdown table
(-> profiles/r0N_gemm_vs_vendor.md): the work-list of the GEMM round.

    python tools/gemm_vs_vendor.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def timed(fn, it):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3      # us


def main():
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    T, d, I, QKV, V, R = 5624, 4096, 14336, 6144, 128264, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device="cuda") * sc).to(torch.bfloat16)
    x = rn(T, d)
    acts = {QKV: rn(T, QKV), d: rn(T, d), 2 * I: rn(T, 2 * I), I: rn(T, I)}
    w = {"qkv": rn(QKV, d, sc=0.02), "o": rn(d, d, sc=0.02), "gu": rn(2 * I, d, sc=0.02), "down": rn(d, I, sc=0.02), "head": rn(V, d, sc=0.02)}
    xh, dlog = rn(R, d), rn(R, V)
    rows = []

    def case(name, layout, M, N, Kd, ours, vendor, n_per_step):
        u = timed(ours, it)
        v = timed(vendor, it)
        fl = 2.0 * M * N * Kd
        rows.append((name, layout, M, N, Kd, u, fl / u / 1e6, v, fl / v / 1e6, n_per_step))
        print(f"{name:10s} {layout} {M}x{N}x{Kd}: ours {u:8.1f} us {fl / u / 1e6:7.1f} TF | vendor {v:8.1f} us {fl / v / 1e6:7.1f} TF | "
              f"ours/vendor time {u / v:5.2f}", flush=True)

    # forward (NT): y = x . W^T
    for nm, a, wt in (("qkv_fwd", x, w["qkv"]), ("o_fwd", acts[d], w["o"]), ("gu_fwd", x, w["gu"]), ("down_fwd", acts[I], w["down"])):
        out = torch.empty(a.shape[0], wt.shape[0], device="cuda", dtype=torch.bfloat16)
        case(nm, "NT", a.shape[0], wt.shape[0], a.shape[1], lambda a=a, wt=wt, out=out: K.gemm_nt(a, wt, out=out),
             lambda a=a, wt=wt, out=out: torch.mm(a, wt.t(), out=out), 32)
    out = torch.empty(R, V, device="cuda", dtype=torch.bfloat16)
    case("head_fwd", "NT", R, V, d, lambda: K.gemm_nt(xh, w["head"], out=out), lambda: torch.mm(xh, w["head"].t(), out=out), 1)
    # dX (NN): dx = dy . W, weight read K-major as stored
    for nm, dy, wt in (("dx_qkv", acts[QKV], w["qkv"]), ("dx_o", acts[d], w["o"]), ("dx_gu", acts[2 * I], w["gu"]), ("dx_down", acts[d], w["down"])):
        out = torch.empty(dy.shape[0], wt.shape[1], device="cuda", dtype=torch.bfloat16)
        case(nm, "NN", dy.shape[0], wt.shape[1], wt.shape[0], lambda dy=dy, wt=wt, out=out: K.gemm_nt(dy, wt, out=out, b_kmajor=True, k=wt.shape[0]),
             lambda dy=dy, wt=wt, out=out: torch.mm(dy, wt, out=out), 32)
    out = torch.empty(R, d, device="cuda", dtype=torch.bfloat16)
    case("dx_head", "NN", R, d, V, lambda: K.gemm_nt(dlog, w["head"], out=out, b_kmajor=True, k=V), lambda: torch.mm(dlog, w["head"], out=out), 1)
    # dW (TN): dW = dy^T . x, both activations read K-major as stored
    for nm, dy, xin in (("dw_qkv", acts[QKV], x), ("dw_o", acts[d], x), ("dw_gu", acts[2 * I], x), ("dw_down", acts[d], acts[I])):
        out = torch.empty(dy.shape[1], xin.shape[1], device="cuda", dtype=torch.bfloat16)
        case(nm, "TN", dy.shape[1], xin.shape[1], dy.shape[0], lambda dy=dy, xin=xin, out=out: K.gemm_nt(dy, xin, out=out, a_kmajor=True, b_kmajor=True),
             lambda dy=dy, xin=xin, out=out: torch.mm(dy.t(), xin, out=out), 32)
    out = torch.empty(V, d, device="cuda", dtype=torch.bfloat16)
    case("dw_head", "TN", V, d, R, lambda: K.gemm_nt(dlog, xh, out=out, a_kmajor=True, b_kmajor=True), lambda: torch.mm(dlog.t(), xh, out=out), 1)

    print("\n| GEMM | layout | M x N x K | ours us | ours TF | vendor us | vendor TF | ours/vendor time | per step | ms/step ours | ms/step at vendor speed |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    tu = tv = tb = 0.0
    for nm, lay, M, N, Kd, u, utf, v, vtf, n in rows:
        tu += u * n / 1e3
        tv += v * n / 1e3
        tb += min(u, v) * n / 1e3
        print(f"| {nm} | {lay} | {M}x{N}x{Kd} | {u:.1f} | {utf:.0f} | {v:.1f} | {vtf:.0f} | {u / v:.2f} | {n} | {u * n / 1e3:.2f} | {v * n / 1e3:.2f} |")
    print(f"\nsum over the step's decoder + head GEMMs: ours {tu:.1f} ms, vendor {tv:.1f} ms, best-of-both {tb:.1f} ms")


if __name__ == "__main__":
    main()
