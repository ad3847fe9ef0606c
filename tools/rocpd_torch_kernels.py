#!/usr/bin/env python3
"""Where do the framework's own elementwise kernels (at::native::*) sit in a step?  For every such kernel of a rocprofv3 rocpd kernel trace:
full name, calls, total / average time, and the most frequent (previous kernel -> next kernel) context in dispatch order.
usage: python tools/rocpd_torch_kernels.py gpurun_out/prof/x_results.db"""
import collections
import re
import sqlite3
import sys


def base(n):
    return re.sub(r"^void ", "", n)


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    agg = collections.OrderedDict()
    for i, (n, s, e) in enumerate(rows):
        if "at::native" not in n and "at_cuda" not in n and "Memcpy" not in n:
            continue
        key = base(n)[:260]
        a = agg.setdefault(key, dict(calls=0, us=0.0, ctx=collections.Counter()))
        a["calls"] += 1
        a["us"] += (e - s) / 1e3
        prev = base(rows[i - 1][0])[:50] if i else "-"
        nxt = base(rows[i + 1][0])[:50] if i + 1 < len(rows) else "-"
        a["ctx"][(prev, nxt)] += 1
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        print(f"{a['calls']:5d} calls {a['us'] / 1e3:8.2f} ms avg {a['us'] / a['calls']:8.1f} us  {k}")
        for (p, n), cnt in a["ctx"].most_common(3):
            print(f"        {cnt:4d} x  after [{p}]  before [{n}]")


if __name__ == "__main__":
    main()
