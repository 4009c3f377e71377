#!/bin/bash
# tools/ab_cfg.sh "bench args" ENV_A ENV_B ... : interleaved runs (3 rounds) of bench.py under each environment setting
args=$1; shift
for r in 1 2 3; do for e in "$@"; do
  v=$(env $e python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$e: $v ms"
done; done | sort | awk -F: '{a[$1]=a[$1] $2}; END{for(k in a) print k ":" a[k]}'
