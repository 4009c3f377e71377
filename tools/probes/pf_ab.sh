#!/bin/bash
# A/B of the next-round L2 warm-up of the 8-phase GEMM (dev build knob GSL_PF: 0 off, 1 after the K loop, 2 in K-loop step 0, 3 in step nk-3)
DEV=GSLORA_HIP_LIB=/root/repo/gs-lora_amd/gslora_hip/libgslora_hip_dev.so
tools/ab_cfg.sh "--steps 10 --warmup 3" "$DEV GSL_PF=0" "$DEV GSL_PF=1" "$DEV GSL_PF=2" "$DEV GSL_PF=3"
for p in 0 2 3; do echo "GSL_PF=$p"; env $DEV GSL_PF=$p python tools/probes/wg_timeline.py 2>&1 | tail -6; done
