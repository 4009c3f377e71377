DEV=GSLORA_HIP_LIB=/root/repo/gs-lora_amd/gslora_hip/libgslora_hip_dev.so
tools/ab_step.sh gpurun_out/mr1 "$DEV GSL_MREV=0" "$DEV GSL_MREV=0x400" 2
tools/ab_step.sh gpurun_out/mr2 "$DEV GSL_MREV=0x481" "$DEV GSL_MREV=0x40001C81" 2
tools/ab_step.sh gpurun_out/mr3 "$DEV GSL_MREV=0 GSLORA_LN_LORA=0" "$DEV GSL_MREV=0x40001C81 GSLORA_LN_LORA=0" 2
