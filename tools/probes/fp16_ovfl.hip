// Probe: does MODE.FP16_OVFL (bit 23) make v_cvt_pk_f16_f32 / v_cvt_f16_f32 saturate finite overflow to +-65504 while keeping Inf / NaN?
// hipcc --offload-arch=gfx950 -O3 tools/probes/fp16_ovfl.hip -o tools/probes/bin/fp16_ovfl && tools/probes/bin/fp16_ovfl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__global__ void k(const float* in, uint32_t* out, int n, int ovfl) {
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  const int i = threadIdx.x;
  if (i < n) {
    f32x2 v = {in[2 * i], in[2 * i + 1]};
    asm volatile("" : "+v"(v));
    out[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
}
int main() {
  float h[16] = {1e6f, -1e6f, 65504.f, 70000.f, INFINITY, -INFINITY, NAN, 1.0f, 65519.9f, 65520.f, -65520.f, 3e38f, 1e-8f, -0.f, 0.1f, 2.5f};
  float* d; uint32_t* o; uint32_t r[8];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 32); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 8, ovfl);
    hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d:", ovfl);
    for (int i = 0; i < 16; ++i) { uint16_t b = (r[i / 2] >> (16 * (i & 1))) & 0xffff; printf(" %g->0x%04x", h[i], b); }
    printf("\n");
  }
  return 0;
}
