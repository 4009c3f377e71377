mkdir -p gpurun_out/r2c
for a in 56 120 184 248; do echo "ABL=$a"; GSL_PP_ABL=$a SHAPE="ffn1 gelu+lora" VARIANTS=10 ROUNDS=2 ITERS=5 timeout 120 python tools/bench_pp.py 2>&1 | grep -v amdgpu.ids | cut -c1-110; done > gpurun_out/r2c/abl2_ffn1.log 2>&1
cat gpurun_out/r2c/abl2_ffn1.log
