#!/bin/bash
# Interleaved same-box comparison of several builds of the library on the bench step: tools/ab_libs.sh OUTDIR ROUNDS lib1.so lib2.so ...
out=$1; n=$2; shift 2
mkdir -p $out
for i in $(seq 1 $n); do
  k=0
  for lib in "$@"; do
    k=$((k+1))
    GSLORA_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-eval --no-secondary --steps 10 --warmup 3 2>/dev/null | tail -1 > $out/L${k}_$i.json
  done
done
python - "$out" "$@" <<'PY'
import json, glob, sys
out, libs = sys.argv[1], sys.argv[2:]
for k, lib in enumerate(libs, 1):
    v = [json.load(open(f)) for f in sorted(glob.glob(f"{out}/L{k}_*.json"))]
    print(lib.split("/")[-1], "ms/step:", [x["ms_per_step"] for x in v], "ffn1:", [x["roofline"]["avg_ms"] for x in v],
          "ffn2dx:", [[k["avg_ms"] for k in x["roofline"]["kernels"] if "mulgrad" in k["kernel"]][0] for x in v], "loss:", [round(x["last_step_meters"]["total"], 4) for x in v])
PY
