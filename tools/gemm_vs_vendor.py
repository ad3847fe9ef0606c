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

    with_vendor = os.environ.get("GEMM_BENCH_VENDOR", "1") == "1"

    def case(name, layout, M, N, Kd, ours, vendor, n_per_step):
        """ours(variant): 12 = 8-wave 32x32x16 ring kernel, 13 = 4-wave 16x16x32 ring kernel; both timed, the faster order-alternated"""
        u12 = timed(lambda: ours(12), it)
        u13 = timed(lambda: ours(13), it)
        u14 = timed(lambda: ours(14), it)
        u12 = min(u12, timed(lambda: ours(12), it))
        u13 = min(u13, timed(lambda: ours(13), it))
        u14 = min(u14, timed(lambda: ours(14), it))
        v = timed(vendor, it) if with_vendor else float("nan")
        fl = 2.0 * M * N * Kd
        u = u13
        rows.append((name, layout, M, N, Kd, u, fl / u / 1e6, v, fl / v / 1e6, n_per_step, u12, u14))
        print(f"{name:10s} {layout} {M}x{N}x{Kd}: v12 {u12:8.1f} us {fl / u12 / 1e6:7.1f} TF | v13 {u13:8.1f} us {fl / u13 / 1e6:7.1f} TF | "
              f"v14 {u14:8.1f} us {fl / u14 / 1e6:7.1f} TF | vendor {v:8.1f} us {fl / v / 1e6:7.1f} TF | v13/v12 {u13 / u12:5.3f} v14/v12 {u14 / u12:5.3f}", flush=True)

    # forward (NT): y = x . W^T
    for nm, a, wt in (("qkv_fwd", x, w["qkv"]), ("o_fwd", acts[d], w["o"]), ("gu_fwd", x, w["gu"]), ("down_fwd", acts[I], w["down"])):
        out = torch.empty(a.shape[0], wt.shape[0], device="cuda", dtype=torch.bfloat16)
        case(nm, "NT", a.shape[0], wt.shape[0], a.shape[1], lambda v, a=a, wt=wt, out=out: K.gemm_nt(a, wt, out=out, variant=v),
             lambda a=a, wt=wt, out=out: torch.mm(a, wt.t(), out=out), 32)
    out = torch.empty(R, V, device="cuda", dtype=torch.bfloat16)
    case("head_fwd", "NT", R, V, d, lambda v: K.gemm_nt(xh, w["head"], out=out, variant=v), lambda: torch.mm(xh, w["head"].t(), out=out), 1)
    # dX (NN): dx = dy . W, weight read K-major as stored
    for nm, dy, wt in (("dx_qkv", acts[QKV], w["qkv"]), ("dx_o", acts[d], w["o"]), ("dx_gu", acts[2 * I], w["gu"]), ("dx_down", acts[d], w["down"])):
        out = torch.empty(dy.shape[0], wt.shape[1], device="cuda", dtype=torch.bfloat16)
        case(nm, "NN", dy.shape[0], wt.shape[1], wt.shape[0], lambda v, dy=dy, wt=wt, out=out: K.gemm_nt(dy, wt, out=out, b_kmajor=True, k=wt.shape[0], variant=v),
             lambda dy=dy, wt=wt, out=out: torch.mm(dy, wt, out=out), 32)
    out = torch.empty(R, d, device="cuda", dtype=torch.bfloat16)
    case("dx_head", "NN", R, d, V, lambda v: K.gemm_nt(dlog, w["head"], out=out, b_kmajor=True, k=V, variant=v), lambda: torch.mm(dlog, w["head"], out=out), 1)
    # dW (TN): dW = dy^T . x, both activations read K-major as stored
    for nm, dy, xin in (("dw_qkv", acts[QKV], x), ("dw_o", acts[d], x), ("dw_gu", acts[2 * I], x), ("dw_down", acts[d], acts[I])):
        out = torch.empty(dy.shape[1], xin.shape[1], device="cuda", dtype=torch.bfloat16)
        case(nm, "TN", dy.shape[1], xin.shape[1], dy.shape[0], lambda v, dy=dy, xin=xin, out=out: K.gemm_nt(dy, xin, out=out, a_kmajor=True, b_kmajor=True, variant=v),
             lambda dy=dy, xin=xin, out=out: torch.mm(dy.t(), xin, out=out), 32)
    out = torch.empty(V, d, device="cuda", dtype=torch.bfloat16)
    case("dw_head", "TN", V, d, R, lambda v: K.gemm_nt(dlog, xh, out=out, a_kmajor=True, b_kmajor=True, variant=v), lambda: torch.mm(dlog.t(), xh, out=out), 1)

    print("\n| GEMM | layout | M x N x K | v12 us | v13 us | v14 us | v13/v12 | v14/v12 | vendor us | best ours TF | vendor TF | per step | ms/step v12 | ms/step v13 | ms/step v14 | ms/step vendor |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    t12 = t13 = t14 = tv = tb = 0.0
    for nm, lay, M, N, Kd, u, utf, v, vtf, n, u12, u14 in rows:
        t12 += u12 * n / 1e3
        t13 += u * n / 1e3
        t14 += u14 * n / 1e3
        tv += v * n / 1e3
        b = min(u, u12, u14)
        tb += b * n / 1e3
        print(f"| {nm} | {lay} | {M}x{N}x{Kd} | {u12:.1f} | {u:.1f} | {u14:.1f} | {u / u12:.3f} | {u14 / u12:.3f} | {v:.1f} | {2.0 * M * N * Kd / b / 1e6:.0f} | {vtf:.0f} | {n} | "
              f"{u12 * n / 1e3:.2f} | {u * n / 1e3:.2f} | {u14 * n / 1e3:.2f} | {v * n / 1e3:.2f} |")
    print(f"\nsum over the step's decoder + head GEMMs: v12 {t12:.1f} ms, v13 {t13:.1f} ms, v14 {t14:.1f} ms, best per shape {tb:.1f} ms, vendor {tv:.1f} ms")


if __name__ == "__main__":
    main()
