#!/bin/bash
# stall breakdown of the attention kernels: SQ counters in separate rocprofv3 --pmc passes (kernel trace only)
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/ap
  rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/ap -o ap -- python $ROOT/tools/probes/attn_only.py > /tmp/ap.log 2>&1 || tail -3 /tmp/ap.log
  DB=$(find /tmp/ap -name "*.db" | head -1)
  python $ROOT/tools/rocpd_pmc.py $DB attn_ 2>/dev/null
done
