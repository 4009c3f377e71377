#!/bin/bash
# Run-to-run determinism of whole training trajectories: three fresh processes per configuration, N steps each, the last step's meters must agree to the last bit
# (a race in any kernel of the step shows up as a diverging trajectory). tools/determinism_soak.sh [steps=150]
n=${1:-150}
for args in "" "--dtype bf16" "--config 4" "--config 5"; do
  for i in 1 2 3; do
    python bench.py $args --no-cpu-baseline --no-eval --no-secondary --steps $n --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$args', json.dumps(d['last_step_meters']))"
  done
done | sort | uniq -c
