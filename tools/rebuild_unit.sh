#!/bin/bash
# Iteration helper: recompile ONE translation unit (both operand formats where it has two) into the existing object directories of the
# product and the dev build and relink both libraries: tools/rebuild_unit.sh attention   (after a full build; ~1 min instead of ~2.5)
set -e
u=$1; root=$(cd "$(dirname "$0")/.." && pwd); c=$root/gs-lora_amd/csrc; h=$root/gs-lora_amd/gslora_hip
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden"
pids=()
for kind in prod dev; do
  d=$h/build/$kind; D=""; [ $kind = dev ] && D="-DGSL_DEV"
  /opt/rocm/bin/hipcc $F $D -c $c/$u.hip -o $d/$u.o 2>$d/$u.err & pids+=($!)
  if [ $u = gemm ] || [ $u = attention ]; then /opt/rocm/bin/hipcc $F $D -DGSL_OP_F16 -c $c/$u.hip -o $d/${u}_f16.o 2>$d/${u}_f16.err & pids+=($!); fi
done
for p in "${pids[@]}"; do wait $p || { grep -h "error" $h/build/*/$u*.err | head; exit 1; }; done
for kind in prod dev; do
  d=$h/build/$kind; out=$h/libgslora_hip.so; [ $kind = dev ] && out=$h/libgslora_hip_dev.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$c/exports.map -o $out $d/gemm.o $d/norm.o $d/lora.o $d/head.o $d/attention.o $d/gemm_f16.o $d/attention_f16.o
done
ls -la $h/*.so
