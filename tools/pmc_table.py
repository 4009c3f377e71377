#!/usr/bin/env python
"""Per-kernel HBM traffic table from the two PMC passes of tools/probes/r03_pmc.sh (tools/rocpd_pmc.py text summaries): FETCH_SIZE doubled
(gfx950: the counter tallies 128-byte requests at 64 B, MI355X_MICROARCH.md "HBM"), WRITE_SIZE at face value, KB = 1024 B, MB = 1e6 B.
Usage: python tools/pmc_table.py gpurun_out/TAG [steps_profiled=3]"""
import re
import sys


def parse(path, counter):
    d = {}
    for line in open(path):
        m = re.match(r"(.+?)\s+" + counter + r"\s+avg\s+([\d.]+)\s+\(n=(\d+)\)", line)
        if m:
            d[m.group(1).strip()] = (float(m.group(2)), int(m.group(3)))
    return d


def main():
    tag, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
    f, w = parse(f"{tag}/pmc_fetch.txt", "FETCH_SIZE"), parse(f"{tag}/pmc_write.txt", "WRITE_SIZE")
    rows, tot = [], 0.0
    for k in sorted(f):
        fe, n = f[k]
        mbf, mbw = 2 * fe * 1024 / 1e6, w.get(k, (0, 0))[0] * 1024 / 1e6
        tot += (mbf + mbw) * n / steps
        if mbf + mbw > 50:
            rows.append((k, n, mbf, mbw))
    print(f"| kernel | launches in {steps} steps | FETCH_SIZE x2 (MB / launch) | WRITE_SIZE (MB / launch) | sum |\n|---|---|---|---|---|")
    for k, n, a, b in rows:
        print(f"| `{k[:80]}` | {n} | {a:.0f} | {b:.0f} | {a + b:.0f} |")
    print(f"\nsum over all kernels: **{tot / 1e3:.1f} GB per step**")


if __name__ == "__main__":
    main()
