#!/bin/bash
# HBM fetch / write bytes of the fused FFN1 GEMM under the launcher knobs (one rocprofv3 --pmc pass per setting, no trace domains mixed in)
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for cfg in "" "GSL_XCD_REMAP=0" "GSL_KROT=0" "GSL_STORE_MODE=0" "GSL_STORE_MODE=2" "GSL_XCD_REMAP=2" "GSL_XCD_REMAP=3"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/sw
    env $cfg rocprofv3 --kernel-trace --pmc $ctr -d /tmp/sw -o sw -- python $ROOT/tools/probes/ffn1_only.py > /dev/null 2>&1
    DB=$(find /tmp/sw -name "*.db" | head -1)
    echo "[$cfg] $(python $ROOT/tools/rocpd_pmc.py $DB p8_kernel 2>/dev/null)"
  done
done
