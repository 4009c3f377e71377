#!/usr/bin/env python
"""Ordered kernel trace of ONE step from a rocprofv3 rocpd database: the dispatches between the last two launches of a marker kernel
(default adamw), with start offset, duration and the gap to the previous kernel's end.
Usage: python tools/rocpd_trace.py x_results.db [marker] [out.md]"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else "adamw"
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = marks[-2], marks[-1]
    step = rows[a + 1:b + 1]
    t0, prev = step[0][1], step[0][1]
    lines = [f"# {len(step)} dispatches, {(step[-1][2] - t0) / 1e3:.1f} us from the first start to the last end", "",
             "| # | kernel | start us | dur us | gap us |", "|---|---|---|---|---|"]
    busy = 0
    for i, (n, s, e) in enumerate(step):
        n = re.sub(r"\(.*$", "", n).replace("void ", "")[:90]
        lines.append(f"| {i} | `{n}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - prev) / 1e3:.1f} |")
        busy += e - s
        prev = e
    lines.append(f"\nkernel time {busy / 1e3:.1f} us, gaps {(step[-1][2] - t0 - busy) / 1e3:.1f} us")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out + "\n")


if __name__ == "__main__":
    main()
