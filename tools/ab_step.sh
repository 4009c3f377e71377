#!/bin/bash
# Interleaved same-box A/B of the bench step: tools/ab_step.sh OUTDIR "ENV_A" "ENV_B" [rounds] [extra bench args]
# e.g. tools/ab_step.sh gpurun_out/ab1 "GSLORA_FWD_STREAM=f32" "GSLORA_FWD_STREAM=bf16" 2
out=$1; a=$2; b=$3; n=${4:-2}; shift 4
mkdir -p $out
for i in $(seq 1 $n); do
  env $a python bench.py --no-cpu-baseline --no-eval --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 > $out/A_$i.json
  env $b python bench.py --no-cpu-baseline --no-eval --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 > $out/B_$i.json
done
python - <<PY
import json,glob
for k,e in (("A","$a"),("B","$b")):
    v=[json.load(open(f)) for f in sorted(glob.glob("$out/%s_*.json"%k))]
    print(k, e, "ms/step:", [x["ms_per_step"] for x in v], "ffn1 ms:", [x["roofline"]["avg_ms"] for x in v], "ffn2dx ms:", [[k["avg_ms"] for k in x["roofline"]["kernels"] if "mulgrad" in k["kernel"]][0] for x in v], "loss:", [round(x["last_step_meters"]["total"],4) for x in v])
PY
