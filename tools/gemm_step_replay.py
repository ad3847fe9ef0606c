#!/usr/bin/env python3
"""Replay of the bf16 GEMM launches of one headline training step (Mantis-8B-SigLIP-Llama-3, 2 samples: 5624 merged rows) and NOTHING
else, in the step's own order, layouts and epilogues: 32 x [q|k|v+RoPE, o+res, gate|up+SwiGLU, down+res] forward, lm_head on 512 rows,
then 32 x [dW(down), dX(down)+SwiGLU-bwd, dW(gate|up), dX(gate|up), dW(o), dX(o), dW(q|k|v), dX(q|k|v)] backward.  The instrument between
the isolated per-shape benches (tools/gemm_vs_vendor.py: one shape in a hot loop) and the step (bench.py --gemm-table): same launch
sequence as the step, sustained for whole steps, but no attention / norm / optimizer kernels in between.

    python tools/gemm_step_replay.py [--steps 3] [--sets 4] [--table PATH] [--no-tower]

--sets N: N distinct weight sets (435 MB each) and activation sets cycled through the 32 layers: N = 1 keeps the operands of a shape
Infinity-Cache-warm between layers like the isolated bench does, N >= 4 streams them from HBM like the step does (16 GB of weights).
Per-shape table (HIP events around every launch, as bench.py) + family total.  A/B another build: MANTIS_HIP_LIB=<lib.so>."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mantis_amd  # noqa: E402,F401
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sets", type=int, default=4)
    ap.add_argument("--table", default=None)
    ap.add_argument("--no-tower", action="store_true")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--keep-forward", action="store_true",
                    help="keep every forward output alive until the end of the step, as the step does for its backward: the forward GEMMs "
                         "then write 32 distinct sets of buffers (~22 GB) instead of the allocator's one hot block per shape")
    args = ap.parse_args()
    T, d, I, QKV, V, R = 5624, 4096, 14336, 6144, 128258, 512
    Vp = (V + 7) // 8 * 8
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(torch.bfloat16)
    S = args.sets
    W = [dict(qkv=rn(QKV, d, sc=0.02), o=rn(d, d, sc=0.02), gu=rn(2 * I, d, sc=0.02), down=rn(d, I, sc=0.02)) for _ in range(S)]
    G = [dict(qkv=torch.empty(QKV, d, device=dev, dtype=torch.bfloat16), o=torch.empty(d, d, device=dev, dtype=torch.bfloat16),
              gu=torch.empty(2 * I, d, device=dev, dtype=torch.bfloat16), down=torch.empty(d, I, device=dev, dtype=torch.bfloat16)) for _ in range(S)]
    # per-set activations (what the step keeps per layer for the backward)
    A = [dict(n1=rn(T, d), qkv=rn(T, QKV), o=rn(T, d), x=rn(T, d), n2=rn(T, d), gu=rn(T, 2 * I), a=rn(T, I), dx=rn(T, d, sc=0.05),
              dqkv=rn(T, QKV, sc=0.05)) for _ in range(S)]
    cos, sin = rn(T, 64), rn(T, 64)
    head, ghead = rn(Vp, d, sc=0.02), torch.empty(Vp, d, device=dev, dtype=torch.bfloat16)
    nf, dlog = rn(R, d), rn(R, Vp, sc=0.01)
    # tower (26 SigLIP layers, M = 4608) + projector
    dv, Iv, Mv = 1152, 4304, 4608
    Iv8 = (Iv + 7) // 8 * 8
    TW = dict(qkv=rn(3 * dv, dv, sc=0.02), o=rn(dv, dv, sc=0.02), fc1=rn(Iv8, dv, sc=0.02), fc2=rn(dv, Iv8, sc=0.02),
              bq=rn(3 * dv), bo=rn(dv), b1=rn(Iv8), b2=rn(dv), p1=rn(d, dv, sc=0.02), p2=rn(d, d, sc=0.02), pb1=rn(d), pb2=rn(d))
    xv, hv = rn(Mv, dv), rn(Mv, Iv8)

    def one_step():
        keep = []
        hold = keep.append if args.keep_forward else (lambda t: None)
        if not args.no_tower:
            for _ in range(26):
                K.gemm_nt(xv, TW["qkv"], bias=TW["bq"])
                K.gemm_nt(xv, TW["o"], bias=TW["bo"], residual=xv)
                K.gemm_nt(xv, TW["fc1"], bias=TW["b1"], act="gelu_pytorch_tanh", n_valid=Iv)
                K.gemm_nt(hv, TW["fc2"], bias=TW["b2"], residual=xv, k=Iv)
            h1 = K.gemm_nt(xv, TW["p1"], bias=TW["pb1"])
            K.gemm_nt(h1, TW["p2"], bias=TW["pb2"])
        for l in range(args.layers):
            w, a = W[l % S], A[l % S]
            hold(K.linear_qkv_rope(a["n1"], w["qkv"], None, cos, sin, 40, 128))
            hold(K.gemm_nt(a["o"], w["o"], residual=a["x"]))
            hold(K.linear_gu_swiglu(a["n2"], w["gu"]))
            hold(K.gemm_nt(a["a"], w["down"], residual=a["x"]))
        K.gemm_nt(nf, head, ldc=Vp)
        K.linear_dw(dlog, nf, ghead, False)
        K.linear_dx(dlog, head, k=Vp)
        for l in reversed(range(args.layers)):
            w, a, gw = W[l % S], A[l % S], G[l % S]
            K.linear_dw(a["dx"], a["a"], gw["down"], False)
            dgu = K.linear_dx_swiglu(a["dx"], w["down"], a["gu"])
            K.linear_dw(dgu, a["n2"], gw["gu"], False)
            K.linear_dx(dgu, w["gu"])
            K.linear_dw(a["dx"], a["o"], gw["o"], False)
            K.linear_dx(a["dx"], w["o"])
            K.linear_dw(a["dqkv"], a["n1"], gw["qkv"], False)
            K.linear_dx(a["dqkv"], w["qkv"])

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    timer = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with K.launch_context(K.LaunchContext(timer=timer)):
        e0.record()
        for _ in range(args.steps):
            one_step()
        e1.record()
    torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / args.steps
    fam_ms = sum(x[3].elapsed_time(x[4]) for x in timer) / args.steps
    fam_fl = sum(x[1] for x in timer) / args.steps
    print(f"replay: sets={S} layers={args.layers} tower={not args.no_tower} lib={os.environ.get('MANTIS_HIP_LIB', 'default')} "
          f"ring={os.environ.get('MANTIS_GEMM_RING', 'auto')}: GEMM family {fam_ms:.2f} ms/step = {fam_fl / fam_ms / 1e9:.0f} TF "
          f"({fam_fl / fam_ms / 1e9 / 2500:.3f} of 2.5 PF), wall {wall:.2f} ms/step, {len(timer) // args.steps} launches/step", flush=True)
    if args.table:
        bench.write_gemm_table(args.table, timer, args.steps, f"REPLAY sets={S}", wall)


if __name__ == "__main__":
    main()
