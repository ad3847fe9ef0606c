#!/usr/bin/env python3
"""Idle time between consecutive kernels of the busiest queue in a rocprofv3 rocpd kernel trace: how much of a step is NO kernel running on the
compute stream (launch gaps, dependency bubbles, host stalls)?  Takes the last `steps` occurrences of a marker kernel (adamw by default) as
step boundaries.    python tools/rocpd_gaps.py p_results.db [marker-substring] """
import sqlite3
import sys


def main():
    db, marker = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "adamw")
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    qcol = next((x for x in ("queue_id", "queue", "stream_id", "stream") if x in cols), None)
    rows = c.execute(f"select {namecol}, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
    if qcol:
        from collections import Counter
        busy = Counter()
        for r in rows:
            busy[r[3]] += r[2] - r[1]
        q0 = busy.most_common(1)[0][0]
        print("queues (busy ms):", {k: round(v / 1e6, 1) for k, v in busy.most_common()})
        main_rows = [r for r in rows if r[3] == q0]
    else:
        main_rows = rows
    marks = [i for i, r in enumerate(main_rows) if marker in r[0]]
    if len(marks) < 2:
        print("marker kernel not found often enough")
        return
    for a, b in zip(marks[:-1], marks[1:]):
        seg = main_rows[a + 1:b + 1]
        span = (seg[-1][2] - main_rows[a][2]) / 1e6
        busy = sum(r[2] - r[1] for r in seg) / 1e6
        gaps = [(seg[i + 1][1] - seg[i][2]) / 1e3 for i in range(len(seg) - 1)]
        pos = [g for g in gaps if g > 0]
        big = sorted(((g, seg[i][0][:40], seg[i + 1][0][:40]) for i, g in enumerate(gaps) if g > 20), reverse=True)[:5]
        print(f"step: {len(seg)} kernels, span {span:.2f} ms, busy {busy:.2f} ms, idle {span - busy:.2f} ms; gaps: n={len(pos)} mean {sum(pos) / max(1, len(pos)):.2f} us, "
              f"median {sorted(pos)[len(pos) // 2] if pos else 0:.2f} us; largest: {[(round(g, 1), x, y) for g, x, y in big]}")


if __name__ == "__main__":
    main()
