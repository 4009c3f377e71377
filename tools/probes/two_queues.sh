#!/bin/bash
# Do two half-batch steps running concurrently (two processes = two hardware queues) beat one full-batch step?
# Prices the "overlap HBM-bound and MFMA-bound kernels of different streams" idea before building a two-stream step.
python bench.py --no-cpu-baseline --steps 20 --warmup 3 --batch 512 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one process, 512+512:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
python bench.py --no-cpu-baseline --steps 20 --warmup 3 --batch 256 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one process, 256+256:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
for n in 2 3; do
  for i in $(seq 1 $n); do
    python bench.py --no-cpu-baseline --steps 60 --warmup 5 --batch $((512 / 2)) 2>/dev/null | tail -1 > /tmp/tq_$i.json &
  done
  wait
  python - <<PY
import json
v=[json.load(open(f"/tmp/tq_{i}.json")) for i in range(1,$n+1)]
print("$n concurrent processes, 256+256 each:", [x["value"] for x in v], "img/s each; sum", sum(x["value"] for x in v), "(upper bound: the timed regions overlap only partly)")
PY
done
