#!/usr/bin/env python3
"""Does RCCL's stream get its own hardware queue?  One fresh process per setting: `n_pre` torch streams are created (and used once)
before the process group's first collective creates RCCL's stream; then dp.rccl_overlap_probe() and dp.hw_queue_probe() run.
World size 1 (MANTIS_DP_FORCE-style), so it works on a 1-GPU box.  -> profiles/r04_rccl_queue_probe.md

    python tools/rccl_queue_probe.py            # sweeps n_pre = 0..9 and GPU_MAX_HW_QUEUES in (4, 8)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n_pre, op):
    sys.path.insert(0, ROOT)
    import mantis_amd  # noqa: F401
    import torch
    import torch.distributed as dist
    from mantis_amd import dp
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    torch.cuda.set_device(0)
    x = torch.zeros(64, device="cuda")
    pre = [torch.cuda.Stream() for _ in range(n_pre)]
    for s in pre:
        with torch.cuda.stream(s):
            x.add_(0.0)
    torch.cuda.synchronize()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    rop = dict(sum=dist.ReduceOp.SUM, avg=dist.ReduceOp.AVG)[op]
    ov = dp.rccl_overlap_probe(None, op=rop)
    side = dp.hw_queue_probe()
    print(f"| {os.environ.get('GPU_MAX_HW_QUEUES')} | {mantis_amd.hw_queues_at_init()} | {n_pre} | {op} | {ov} | {side} of 7 |", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), sys.argv[3])
    else:
        print("| GPU_MAX_HW_QUEUES | queues at HIP init | streams created before RCCL's | op | all-reduce ran beside busy compute stream | fresh side streams that did |")
        print("|---|---|---|---|---|---|", flush=True)
        for q in ("8", "4"):
            for op in ("avg", "sum"):
                for n_pre in ((0, 1, 2, 3, 5, 6, 7, 8, 9) if op == "avg" else (0, 7)):
                    env = dict(os.environ, GPU_MAX_HW_QUEUES=q, MASTER_PORT=str(29600 + n_pre))
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n_pre), op], env=env, capture_output=True, text=True, timeout=300)
                    out = [l for l in r.stdout.splitlines() if l.startswith("|")]
                    print(out[-1] if out else f"| {q} | ? | {n_pre} | {op} | FAILED rc={r.returncode} {r.stderr[-200:]!r} | |", flush=True)
