#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (the default output of `rocprofv3 --kernel-trace --stats`
on ROCm 7.2) into a per-kernel table: calls, total / average / min / max duration, share.
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {namecol} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, cnt, s, a, mn, mx in rows:
        lines.append(f"| `{short(n)}` | {cnt} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / tot:.1f} |")
    lines.append(f"\nTotal kernel time: {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
