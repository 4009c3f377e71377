#!/bin/bash
# rocprofv3 kernel trace of the default bench step, summarised on the GPU box: tools/prof_step.sh TAG [extra bench args]
tag=$1; shift
export TMPDIR=/tmp
d=/tmp/prof_$tag; rm -rf $d; mkdir -p $d gpurun_out
rocprofv3 --kernel-trace --stats -d $d -o $tag -- python bench.py --no-cpu-baseline --no-eval --no-mem-kernels --no-secondary --steps 10 --warmup 3 "$@" > $d/bench.out 2>$d/bench.err
tail -1 $d/bench.out > gpurun_out/${tag}_bench.json
db=$(find $d -name "*_results.db" | head -1)
python tools/rocpd_stats.py $db gpurun_out/${tag}_kernel_stats.md > /dev/null
head -30 gpurun_out/${tag}_kernel_stats.md; tail -2 gpurun_out/${tag}_kernel_stats.md
