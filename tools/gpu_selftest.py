#!/usr/bin/env python3
"""Run every kernel parity check on the GPU box and write gpurun_out/selftest.json (does not stop at the first failure)."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import mantis_amd  # noqa: F401  (before the first GPU call: exports GPU_MAX_HW_QUEUES=8, as the pytest run does through conftest)
    import torch
    from tests import gpu_checks
    from tests import helpers as Hh
    only = sys.argv[1:]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    res = {}
    print("device:", torch.cuda.get_device_name(0), flush=True)
    for name, fn in gpu_checks.all_checks().items():
        if only and not any(o in name for o in only):
            continue
        t0 = time.time()
        Hh.CURRENT_CASE = name
        try:
            v = fn()
            torch.cuda.synchronize()
            res[name] = dict(ok=True, err=v, s=round(time.time() - t0, 2))
            print(f"PASS {name} {v}", flush=True)
        except Exception as e:  # noqa
            msg = "".join(traceback.format_exception_only(type(e), e)).strip()
            res[name] = dict(ok=False, msg=msg[-600:], s=round(time.time() - t0, 2))
            print(f"FAIL {name}: {msg[-300:]}", flush=True)
            try:
                torch.cuda.synchronize()
            except Exception as e2:  # a faulted context cannot continue
                print("device fault, aborting:", e2, flush=True)
                break
        with open(os.path.join(ROOT, "gpurun_out", "selftest.json"), "w") as f:
            json.dump(res, f, indent=1)
    if Hh.GRAD_REPORTS:
        Hh.write_grad_parity(os.path.join(ROOT, "gpurun_out", "grad_parity.md"))
    nfail = sum(1 for r in res.values() if not r["ok"])
    print(f"{len(res) - nfail}/{len(res)} passed", flush=True)
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
