# FETCH_SIZE / WRITE_SIZE calibration passes (tools/probes/pmc_calib.hip). TAG = output dir under gpurun_out/
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-pmc_calib}
mkdir -p gpurun_out/$TAG
cd /tmp
rm -rf /tmp/cf /tmp/cw
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/cf -o cf -- $ROOT/tools/probes/bin/pmc_calib > /dev/null 2> $ROOT/gpurun_out/$TAG/cf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/cw -o cw -- $ROOT/tools/probes/bin/pmc_calib > /dev/null 2> $ROOT/gpurun_out/$TAG/cw.err
cd $ROOT
python tools/rocpd_pmc.py $(find /tmp/cf -name "*.db" | head -1) "" > gpurun_out/$TAG/calib_fetch.txt 2>/dev/null
python tools/rocpd_pmc.py $(find /tmp/cw -name "*.db" | head -1) "" > gpurun_out/$TAG/calib_write.txt 2>/dev/null
cat gpurun_out/$TAG/calib_fetch.txt gpurun_out/$TAG/calib_write.txt
