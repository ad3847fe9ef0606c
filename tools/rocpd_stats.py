#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel stats table (markdown).
usage: python tools/rocpd_stats.py gpurun_out/prof/r1_results.db [skip_first_n_dispatches_fraction]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = (rows[-1][2] - rows[0][1]) / 1e3 if rows else 0
    print(f"| kernel | calls | total_ms | avg_us | min_us | max_us | % |")
    print("|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {a[0]} | {a[1] / 1e3:.2f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100 * a[1] / tot:.1f} |")
    print(f"\nkernel time total {tot / 1e3:.1f} ms over a span of {span / 1e3:.1f} ms ({len(rows)} dispatches)")


if __name__ == "__main__":
    main()
