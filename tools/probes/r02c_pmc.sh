# round-2 (c) PMC passes over the step at HEAD (separate passes, kernel trace only): FETCH_SIZE, then WRITE_SIZE + MFMA busy
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out/r2e
cd /tmp
rm -rf /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/r2e/pf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pw -o pw -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/r2e/pw.err
cd $ROOT
python tools/rocpd_pmc.py $(find /tmp/pf -name "*.db" | head -1) "" > gpurun_out/r2e/pmc_fetch.txt 2>/dev/null
python tools/rocpd_pmc.py $(find /tmp/pw -name "*.db" | head -1) "" > gpurun_out/r2e/pmc_write.txt 2>/dev/null
wc -l gpurun_out/r2e/pmc_fetch.txt gpurun_out/r2e/pmc_write.txt
