# round-3 PMC passes over the step at HEAD (separate passes, kernel trace only): FETCH_SIZE, then WRITE_SIZE + MFMA busy. TAG = output dir
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-r3pmc}
mkdir -p gpurun_out/$TAG
cd /tmp
rm -rf /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eval --no-mem-kernels --no-secondary > /dev/null 2> $ROOT/gpurun_out/$TAG/pf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pw -o pw -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eval --no-mem-kernels --no-secondary > /dev/null 2> $ROOT/gpurun_out/$TAG/pw.err
cd $ROOT
python tools/rocpd_pmc.py $(find /tmp/pf -name "*.db" | head -1) "" > gpurun_out/$TAG/pmc_fetch.txt 2>/dev/null
python tools/rocpd_pmc.py $(find /tmp/pw -name "*.db" | head -1) "" > gpurun_out/$TAG/pmc_write.txt 2>/dev/null
wc -l gpurun_out/$TAG/pmc_fetch.txt gpurun_out/$TAG/pmc_write.txt
