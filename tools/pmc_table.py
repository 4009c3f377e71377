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
    # MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (kernel cycles x 1024); kernel cycles = GRBM_GUI_ACTIVE / 8 (the
    # counter comes back summed over the 8 XCDs). north_star: "rocprof must show ... MFMA utilisation on the fused FFN+LoRA GEMM".
    busy, act = parse(f"{tag}/pmc_write.txt", "SQ_VALU_MFMA_BUSY_CYCLES"), parse(f"{tag}/pmc_write.txt", "GRBM_GUI_ACTIVE")
    rows, tot = [], 0.0
    for k in sorted(f):
        fe, n = f[k]
        mbf, mbw = 2 * fe * 1024 / 1e6, w.get(k, (0, 0))[0] * 1024 / 1e6
        tot += (mbf + mbw) * n / steps
        if mbf + mbw > 50:
            mf = 100.0 * busy[k][0] / (act[k][0] / 8.0 * 1024.0) if k in busy and k in act and act[k][0] > 0 else float("nan")
            rows.append((k, n, mbf, mbw, mf))
    print(f"| kernel | launches in {steps} steps | FETCH_SIZE x2 (MB / launch) | WRITE_SIZE (MB / launch) | sum | MFMA busy % |\n|---|---|---|---|---|---|")
    for k, n, a, b, mf in rows:
        print(f"| `{k[:80]}` | {n} | {a:.0f} | {b:.0f} | {a + b:.0f} | {mf:.1f} |")
    print(f"\nsum over all kernels: **{tot / 1e3:.1f} GB per step**")


if __name__ == "__main__":
    main()
