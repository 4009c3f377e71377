// Probe of gfx950 ds_read_b64_tr_b16: which LDS elements does lane l receive, as a function of the per-lane addresses?
// Build: hipcc --offload-arch=gfx950 -O2 -o tr_read_probe tr_read_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int mode) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  // mode 0: lane address = l*8 bytes (contiguous 4-element pieces)
  // mode 1: 4 x 16 row-major blocks per 16-lane group: lane i of a group -> row i/4, cols (i%4)*4 ; group g -> block g (64 elements apart)
  // mode 2: like 1 but row stride 72 elements (a padded row-major [key][64+8] image), group g -> rows 4g..4g+3
  int elem;
  if (mode == 0) elem = l * 4;
  else if (mode == 1) elem = (l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4;
  else elem = ((l >> 4) * 4 + ((l & 15) >> 2)) * 72 + (l & 3) * 4;
  const uint32_t addr = (uint32_t)(uintptr_t)(lds) + elem * 2;   // LDS byte address (low 32 bits of the generic pointer are the LDS offset)
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
