#!/bin/bash
# attention fwd / bwd time and LDS bank conflicts for the library variants under build_variants/ (LDS row stride / swizzle)
export TMPDIR=/tmp
ROOT=$PWD
for lib in "" $(ls $ROOT/build_variants/*.so 2>/dev/null); do   # variants: hipcc ... -DGSL_ATTN_KLD=.. (historic: the macros were removed once 160-byte unswizzled rows were chosen)
  export GSLORA_HIP_LIB=$lib
  echo "== ${lib:-default}"
  B=1024 ABLS=0 python tools/bench_attn.py 2>&1 | grep abl
  (cd /tmp; rm -rf /tmp/av; rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/av -o av -- python $ROOT/tools/probes/attn_only.py > /tmp/av.log 2>&1; python $ROOT/tools/rocpd_pmc.py $(find /tmp/av -name "*.db" | head -1) attn_ 2>/dev/null)
done
