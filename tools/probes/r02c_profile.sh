# round-2 evidence run at HEAD (output directory gpurun_out/r2e): bench line + rocprofv3 kernel stats of the same command
set -x
mkdir -p gpurun_out/r2e
python bench.py --steps 10 --warmup 3 > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r2e/kt -o kt -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/gpurun_out/r2e/kt_bench.json 2> $ROOT/gpurun_out/r2e/kt.err
cd $ROOT
DB=$(find gpurun_out/r2e/kt -name "*.db" | head -1); python tools/rocpd_stats.py $DB gpurun_out/r2e/kernel_stats.md > /dev/null
rm -rf gpurun_out/r2e/kt
cut -c1-300 gpurun_out/r2e/bench.json; head -30 gpurun_out/r2e/kernel_stats.md
