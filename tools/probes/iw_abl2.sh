#!/bin/bash
# in-wave pipelined GEMM (variant 11): anti-phase rows x DMA ablations, QKV (trivial epilogue) and fused FFN1 shapes
for shape in qkv "ffn1 gelu+lora"; do
  for abl in 0 512 3 515 1 2; do
    echo -n "abl=$abl  "; GSL_PP_ABL=$abl VARIANTS=8,11 SHAPE="$shape" ROUNDS=2 python tools/bench_pp.py 2>&1 | grep -v amdgpu | cut -c1-125
  done
done
