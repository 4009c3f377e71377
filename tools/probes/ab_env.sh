#!/bin/bash
# same-box A/B of one environment knob: tools/probes/ab_env.sh NAME A B [rounds] -> interleaved bench.py runs
name=$1; a=$2; b=$3; rounds=${4:-3}
for i in $(seq $rounds); do
  for v in $a $b; do
    ms=$(env $name=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
    echo "$name=$v  $ms ms/step"
  done
done
