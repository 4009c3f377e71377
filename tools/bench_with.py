#!/usr/bin/env python
"""bench.py with decided schedule constants of gslora_hip/vit_runner.py patched for an A/B: tools/bench_with.py RANK_PAIR32=False [NAME=VALUE ...] -- <bench args>
(the constants are plain module attributes on purpose — no environment reads in the product; this is the measuring side's way in)."""
import ast, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gs-lora_amd")]
sep = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
sets, rest = sys.argv[1:sep], sys.argv[sep + 1:]
import bench  # noqa: E402
from gslora_hip import vit_runner  # noqa: E402
for kv in sets:
    k, v = kv.split("=", 1)
    if not hasattr(vit_runner, k):
        raise SystemExit(f"vit_runner has no attribute {k}")
    setattr(vit_runner, k, ast.literal_eval(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + rest
bench.main()
