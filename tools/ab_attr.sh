#!/bin/bash
# Interleaved same-box comparison of vit_runner schedule constants on the bench step: tools/ab_attr.sh OUTDIR ROUNDS NAME=VALUE NAME=VALUE2 ...
out=$1; n=$2; shift 2
mkdir -p $out
for i in $(seq 1 $n); do k=0; for a in "$@"; do k=$((k+1)); python tools/bench_with.py $a -- --no-cpu-baseline --no-eval --no-secondary --steps 10 --warmup 3 2>/dev/null | tail -1 > $out/L${k}_$i.json; done; done
python - "$out" "$@" <<'PY'
import json, glob, sys
out, libs = sys.argv[1], sys.argv[2:]
for k, lib in enumerate(libs, 1):
    v = [json.load(open(f)) for f in sorted(glob.glob(f"{out}/L{k}_*.json"))]
    print(lib, "ms/step:", [x["ms_per_step"] for x in v], "ffn1:", [x["roofline"]["avg_ms"] for x in v], "loss:", [round(x["last_step_meters"]["total"], 4) for x in v])
PY
