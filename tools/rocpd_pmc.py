#!/usr/bin/env python
"""Per-kernel average of the PMC counters in a rocprofv3 rocpd database."""
import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm"
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
print("# columns:", cols, file=sys.stderr)
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
kn = "kernel_name" if "kernel_name" in ix else [n for n in cols if "kernel" in n and "name" in n][0]
cn = "counter_name" if "counter_name" in ix else [n for n in cols if "counter" in n and "name" in n][0]
vn = "value" if "value" in ix else "counter_value"
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    k = re.sub(r"\(.*$", "", r[ix[kn]]).replace("void ", "")
    if pat not in k:
        continue
    a = acc[(k[:70], r[ix[cn]])]
    a[0] += r[ix[vn]]; a[1] += 1
for (k, cname), (s, n) in sorted(acc.items()):
    print(f"{k:70s} {cname:32s} avg {s / n:16.1f}  (n={n})")
