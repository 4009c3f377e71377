# round-2 evidence run: new GPU tests, bench line, rocprofv3 kernel stats and PMC passes (separate passes, no trace-domain mix)
set -x
mkdir -p gpurun_out/r2f
python -m pytest tests/test_hip_engines.py tests/test_hip_ops.py tests/test_hip_graph.py tests/test_hip_model.py -m gpu -q 2>&1 | grep -v "^Test \|^current\|^Perfom\|^Epoch\|^Task\|device_id\|amdgpu.ids" | tail -15 > gpurun_out/r2f/tests.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r2f/kt -o kt -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/gpurun_out/r2f/kt_bench.json 2> $ROOT/gpurun_out/r2f/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $ROOT/gpurun_out/r2f/pmc_fetch -o pf -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/r2f/pf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $ROOT/gpurun_out/r2f/pmc_write -o pw -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/r2f/pw.err
cd $ROOT
for d in kt pmc_fetch pmc_write; do find gpurun_out/r2f/$d -name "*.db" | head -3; done
DB=$(find gpurun_out/r2f/kt -name "*.db" | head -1); python tools/rocpd_stats.py $DB gpurun_out/r2f/kernel_stats.md > /dev/null
DB=$(find gpurun_out/r2f/pmc_fetch -name "*.db" | head -1); python tools/rocpd_pmc.py $DB "" > gpurun_out/r2f/pmc_fetch.txt 2>/dev/null
DB=$(find gpurun_out/r2f/pmc_write -name "*.db" | head -1); python tools/rocpd_pmc.py $DB "" > gpurun_out/r2f/pmc_write.txt 2>/dev/null
rm -rf gpurun_out/r2f/kt gpurun_out/r2f/pmc_fetch gpurun_out/r2f/pmc_write
tail -5 gpurun_out/r2f/tests.log; cut -c1-300 gpurun_out/r2f/bench.json; head -12 gpurun_out/r2f/kernel_stats.md
