#!/usr/bin/env python3
"""Per-kernel sums of one rocprofv3 --pmc pass (CSV output): python tools/pmc_summary.py <counter_collection.csv> [skip_dispatches]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md,
HBM section), so the table also prints the x2-corrected figure."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:64]


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    cname = None
    for row in csv.DictReader(open(path)):
        if int(row["Dispatch_Id"]) <= skip:
            continue
        cname = row["Counter_Name"]
        a = agg[short(row["Kernel_Name"])]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
        a[2] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    fix = 2.0 if cname == "FETCH_SIZE" else 1.0
    print(f"| kernel | launches | {cname} total MiB (raw) | corrected MiB | per launch MiB (corrected) | GB/s over kernel time |")
    print("|---|---|---|---|---|---|")
    for k, (n, v, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        mib = v / 1024.0
        print(f"| {k} | {n} | {mib:.1f} | {mib * fix:.1f} | {mib * fix / n:.2f} | {mib * fix * 1.048576e6 / (us * 1e-6) / 1e9:.0f} |")


if __name__ == "__main__":
    main()
