// pmc_calib.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, per access width and pattern
// (the gradient-fused FFN2-dX GEMM showed 1.30x its algorithmic bytes by FETCH_SIZE x 2 + WRITE_SIZE; its epilogue
// reads the 8-bit GELU' codes 8 bytes per lane — a width the guide's "FETCH_SIZE reports half of a 16 B/lane stream" was not calibrated on).
// Every kernel touches each byte of a 1 GiB buffer (4x the 256 MB Infinity Cache) exactly once.
// Run under the profiler in two passes (tools/probes/pmc_calib.sh): --pmc FETCH_SIZE, then --pmc WRITE_SIZE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr size_t BYTES = 1ull << 30;

template <typename T> __global__ void rd_stream(const T* __restrict__ p, size_t n, T* sink, int flag) {      // coalesced: a wave reads 64 * sizeof(T) consecutive bytes
  T acc{};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = p[i];
    if constexpr (sizeof(T) == 16) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    else if constexpr (sizeof(T) == 8) { acc.x ^= v.x; acc.y ^= v.y; }
    else acc ^= v;
  }
  if ((int)threadIdx.x == flag) sink[0] = acc;
}
// 16 B per lane, 8 lanes per 128-byte row segment, rows `stride_b` bytes apart (a [rows, 64 bf16] operand panel of a row-major matrix)
__global__ void rd_rowseg16(const uint4* __restrict__ p, size_t nseg, size_t stride16, uint4* sink, int flag) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  const size_t lanes = (size_t)gridDim.x * blockDim.x;
  // segment s = (panel, row): panel-major so that all bytes are touched once: address = row * stride + panel * 128 B
  const size_t rows = nseg / (stride16 / 8), t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = t; i < nseg * 8; i += lanes) {
    const size_t seg = i >> 3, c = i & 7, panel = seg / rows, row = seg % rows;
    const uint4 v = p[row * stride16 + panel * 8 + c];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((int)threadIdx.x == flag) sink[0] = acc;
}
// 8 B per lane, 8 lanes per 64-byte row piece, 8 rows per wave-instruction = 512 consecutive bytes (the slab-major GELU' codes as the
// FFN2-dX epilogue reads them) — coalesced, narrow
__global__ void rd_lds_dma16(const uint4* __restrict__ p, size_t n, uint4* sink, int flag) {      // global_load_lds_dwordx4 stream (the GEMM operand path)
  __shared__ uint4 buf[4][64];
  const int wave = threadIdx.x >> 6;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_amdgcn_global_load_lds((gptr_t)(p + i), (lptr_t)(&buf[wave][0]), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x == flag) sink[0] = buf[0][0];
}
template <typename T> __global__ void wr_stream(T* __restrict__ p, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void wr_stream_nt16(uint4* __restrict__ p, size_t n, uint4 v) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t w = {v.x, v.y, v.z, v.w};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(p + i));
}

int main(int argc, char**) {
  const int flag = argc > 5 ? 3 : -1;      // never true at run time, unknown at compile time: the loads stay
  void* buf; uint4* sink;
  hipMalloc(&buf, BYTES); hipMalloc(&sink, 64);
  hipMemset(buf, 1, BYTES);
  hipDeviceSynchronize();
  const dim3 g(256 * 8), b(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(rd_stream<uint4>, g, b, 0, 0, (const uint4*)buf, BYTES / 16, sink, flag);
    hipLaunchKernelGGL(rd_stream<uint2>, g, b, 0, 0, (const uint2*)buf, BYTES / 8, (uint2*)sink, flag);
    hipLaunchKernelGGL(rd_stream<uint32_t>, g, b, 0, 0, (const uint32_t*)buf, BYTES / 4, (uint32_t*)sink, flag);
    hipLaunchKernelGGL(rd_rowseg16, g, b, 0, 0, (const uint4*)buf, BYTES / 128, (size_t)(4096 / 16), sink, flag);      // 128-B segments of 4 KB rows (K = 2048 bf16)
    hipLaunchKernelGGL(rd_lds_dma16, g, b, 0, 0, (const uint4*)buf, BYTES / 16, sink, flag);
    hipLaunchKernelGGL(wr_stream<uint4>, g, b, 0, 0, (uint4*)buf, BYTES / 16, make_uint4(1, 2, 3, 4));
    hipLaunchKernelGGL(wr_stream<uint2>, g, b, 0, 0, (uint2*)buf, BYTES / 8, make_uint2(1, 2));
    hipLaunchKernelGGL(wr_stream<uint32_t>, g, b, 0, 0, (uint32_t*)buf, BYTES / 4, 7u);
    hipLaunchKernelGGL(wr_stream_nt16, g, b, 0, 0, (uint4*)buf, BYTES / 16, make_uint4(1, 2, 3, 4));
  }
  hipDeviceSynchronize();
  printf("each kernel touches %zu bytes (1 GiB = 1073.7 MB) exactly once; 3 repetitions\n", BYTES);
  return 0;
}
