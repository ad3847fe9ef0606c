#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of one `bench.py` run into the tracked JSON that bench.py reads for `roofline.traffic` and
`mfma_busy_pct` (no literals in bench.py).

Collect on the GPU box, one counter set per pass, `--kernel-trace` only beside `--pmc` (MI355X_MICROARCH.md, HBM section):

    cd /tmp && export TMPDIR=/tmp
    B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer"
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- $B
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- $B
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
              --output-format csv -d gpurun_out/pmc_sq -o p -- $B
    python tools/pmc_step_report.py --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write --sq gpurun_out/pmc_sq \
           --out profiles/r02_pmc_step.json

Units and corrections: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so
it is doubled (the guide's correction).  Both counters sit on the L2's memory side: Infinity-Cache hits are included, so the
figure is an upper bound on HBM bytes.  Calibration against a kernel of known byte count (adamw: 14 B read + 14 B written per
parameter) is printed and stored.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

GEMM_FAMILY = ("gemm_nt_ring_kernel", "gemm_nt_ring16_kernel", "gemm_nt_ring176_kernel", "gemm_nt_kernel")
GEMM_FINISH = ("gemm_ring16_finish_kernel",)       # round 5: K-split finishing pass -- its bytes belong to the family, its launches do not count
ATTN_FAMILY = ("attn_fwd_kernel", "attn_fwd64_kernel", "attn_bwd")


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:72]


def read_pass(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    agg = defaultdict(lambda: defaultdict(float))         # kernel -> counter -> sum ; plus "_n", "_us"
    seen = set()
    for row in csv.DictReader(open(files[0])):
        k = short(row["Kernel_Name"])
        if k.startswith(("at::native", "__amd_rocclr", "hipprand", "rocprim")) or "at::native" in k:
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        did = row["Dispatch_Id"]
        if did not in seen:
            seen.add(did)
            agg[k]["_n"] += 1
            agg[k]["_us"] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    return agg


def family(agg, prefixes, key):
    n = sum(v["_n"] for k, v in agg.items() if k.startswith(prefixes))
    tot = sum(v[key] for k, v in agg.items() if k.startswith(prefixes))
    return n, tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--sq")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    ap.add_argument("--gemm-family", default=",".join(GEMM_FAMILY),
                    help="comma-separated kernel-name prefixes summarised as `gemm_family` (Qwen2-VL fp8 run: gemm_fp8_ring_kernel,gemm_fp8_nt_kernel)")
    a = ap.parse_args()
    gemm_family = tuple(x for x in a.gemm_family.split(",") if x)
    f, w = read_pass(a.fetch), read_pass(a.write)
    out = dict(source=dict(fetch=a.fetch, write=a.write, sq=a.sq, note=a.note,
                           corrections="FETCH_SIZE KiB x1024 x2 (gfx950 64-B tally of 128-B requests); WRITE_SIZE KiB x1024; "
                                       "memory-side of L2 (Infinity-Cache hits included)"),
               per_kernel={})
    for k in sorted(set(f) | set(w)):
        n = int(f[k]["_n"] or w[k]["_n"])
        if not n:
            continue
        fb = f[k]["FETCH_SIZE"] * 1024 * 2 / n if k in f else None
        wb = w[k]["WRITE_SIZE"] * 1024 / n if k in w else None
        out["per_kernel"][k] = dict(launches=n, fetch_bytes_per_launch=fb, write_bytes_per_launch=wb,
                                    avg_us=(f[k]["_us"] / f[k]["_n"]) if f[k]["_n"] else None)
    for fam, pre in (("gemm_family", gemm_family), ("attn_family", ATTN_FAMILY)):
        n, fb = family(f, pre, "FETCH_SIZE")
        n2, wb = family(w, pre, "WRITE_SIZE")
        if n and n2:
            out[fam] = dict(launches=int(n), fetch_bytes_per_launch=fb * 2048 / n, write_bytes_per_launch=wb * 1024 / n2,
                            traffic_bytes_per_launch=fb * 2048 / n + wb * 1024 / n2)
            if fam == "gemm_family":
                _, ffb = family(f, GEMM_FINISH, "FETCH_SIZE")
                _, fwb = family(w, GEMM_FINISH, "WRITE_SIZE")
                fin = ffb * 2048 / n + fwb * 1024 / n2
                out[fam].update(finish_kernel_bytes_per_launch=fin, traffic_bytes_per_launch_incl_finish=out[fam]["traffic_bytes_per_launch"] + fin,
                                note="launches = GEMM kernel launches; the K-split finishing kernels' traffic (slab reads + the remainder tiles' epilogue) "
                                     "is carried separately and added in traffic_bytes_per_launch_incl_finish")
    ad = [k for k in f if k.startswith("adamw_kernel")]
    if ad:
        out["calibration"] = dict(kernel=ad[0], fetch_bytes_per_launch=out["per_kernel"][ad[0]]["fetch_bytes_per_launch"],
                                  write_bytes_per_launch=out["per_kernel"][ad[0]]["write_bytes_per_launch"],
                                  note="algorithmic: 14 B read + 14 B written per (8-padded) trainable parameter")
    if a.sq:
        s = read_pass(a.sq)
        busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in s.values())
        grbm = sum(v["GRBM_GUI_ACTIVE"] for v in s.values())
        us = sum(v["_us"] for v in s.values())
        step = dict(mfma_busy_pct=100.0 * busy / (1024.0 * grbm / 8.0) if grbm else None,
                    effective_clock_ghz=(grbm / 8.0) / (us * 1e-6) / 1e9 if us else None, kernel_time_ms=us / 1e3,
                    wave_wait_pct=100.0 * sum(v["SQ_WAIT_ANY"] for v in s.values()) / max(1.0, sum(v["SQ_WAVE_CYCLES"] for v in s.values())),
                    formula="sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8 XCDs), all product kernels of the run")
        per = {}
        for k, v in s.items():
            if v["GRBM_GUI_ACTIVE"]:
                per[k] = dict(launches=int(v["_n"]), mfma_busy_pct=100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0),
                              clock_ghz=(v["GRBM_GUI_ACTIVE"] / 8.0) / (v["_us"] * 1e-6) / 1e9, total_ms=v["_us"] / 1e3)
        out["step"], out["sq_per_kernel"] = step, per
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps({k: out[k] for k in ("gemm_family", "attn_family", "calibration", "step") if k in out}, indent=1))


if __name__ == "__main__":
    main()
