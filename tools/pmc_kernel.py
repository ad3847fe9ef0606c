#!/usr/bin/env python3
"""Per-kernel SQ counter summary of one rocprofv3 --pmc pass (CSV): python tools/pmc_kernel.py <dir-or-csv> [name-filter]
Prints, per kernel: launches, avg us, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / time), MFMA pipe busy % of the SIMD-cycles,
and the wave-cycle split (SQ_WAIT_ANY = parked on s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)[:60]


def main():
    p = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    if os.path.isdir(p):
        p = glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = defaultdict(lambda: defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if flt not in k or "at::native" in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            agg[k]["_n"] += 1
            agg[k]["_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["_us"]):
        n, us = v["_n"], v["_us"]
        g = v.get("GRBM_GUI_ACTIVE", 0.0)
        line = f"{k}: n={int(n)} avg={us / n:.1f}us"
        if g:
            line += f" clk={(g / 8) / (us * 1e-6) / 1e9:.2f}GHz mfma_busy={100 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * g / 8):.1f}%"
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            line += " | wave-cycles: " + " ".join(f"{c[3:].lower()}={100 * v.get(c, 0) / wc:.0f}%" for c in
                                                  ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if c in v)
        extra = [c for c in v if c not in ("_n", "_us", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                           "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS")]
        if extra:
            line += " | " + " ".join(f"{c}={v[c] / n:.3g}" for c in extra)
        print(line)


if __name__ == "__main__":
    main()
