# LDS bank-conflict pass over the step: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (+ wave / wait cycles) per kernel. TAG = output dir
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-pmc_lds}
mkdir -p gpurun_out/$TAG
cd /tmp
rm -rf /tmp/pl
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pl -o pl -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eval > /dev/null 2> $ROOT/gpurun_out/$TAG/pl.err
cd $ROOT
python tools/rocpd_pmc.py $(find /tmp/pl -name "*.db" | head -1) "" > gpurun_out/$TAG/pmc_lds.txt 2>/dev/null
wc -l gpurun_out/$TAG/pmc_lds.txt
