#!/usr/bin/env python3
"""Time the AdamW pass alone: python tools/adamw_bench.py [n_params_in_millions]  (GPU box).
Prints ms and TB/s of mantis_adamw_split (26 B / parameter) and mantis_adamw (28 B) over n parameters (profiles/r05_experiments.md 11)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mantis_amd  # noqa: F401,E402
import torch  # noqa: E402
from mantis_amd import hip_ops as K  # noqa: E402


def main():
    n = int(float(sys.argv[1]) * 1e6) // 8 * 8 if len(sys.argv) > 1 else 2_000_000_000
    dev = "cuda"
    p = (torch.randn(n // 8, device=dev).repeat(8) * 0.05).to(torch.bfloat16)
    g = (torch.randn(n // 8, device=dev).repeat(8) * 0.01).to(torch.bfloat16)
    lo = torch.zeros(n, dtype=torch.int16, device=dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    out = {}
    for name, nbytes in (("split", 26), ("fp32", 28)):
        master = p.float() if name == "fp32" else None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for it in range(6):
            if it == 2:
                ev[0].record()
            if name == "split":
                K.adamw_split_flat(p, g, lo, m, v, 1e-5, 0.9, 0.999, 1e-8, 0.0, it + 1)
            else:
                K.adamw_flat(p, g, master, m, v, 1e-5, 0.9, 0.999, 1e-8, 0.0, it + 1)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 4
        out[name] = (ms, n * nbytes / ms / 1e9)
        del master
    print(f"n={n/1e9:.2f}G: " + "  ".join(f"{k} {ms:.2f} ms {tb:.2f} TB/s" for k, (ms, tb) in out.items()), flush=True)


if __name__ == "__main__":
    main()
