// vmem_shape.hip — what does ONE vector-memory instruction cost the CU's memory pipe as a function of its ADDRESS SHAPE?
// (round 5: the fragment-layout stores of the attention backward — 64 lanes x 8 B = 16 rows x 32 B — held the pipe ~115 cycles each and every
//  other load of the CU queued behind them, profiles/r05_j_attn_merged.md. The GEMM epilogues use "full 128-byte rows", 8 rows per instruction;
//  is that the cheap end of the curve?)
// Every wave issues ITER instructions of one shape: R rows per instruction (row stride 4096 B, a [M, 2048] 16-bit tensor), 64 / R lanes per row,
// B bytes per lane, walking along the rows. The footprint (32 MB for the whole chip at 4 waves per CU) is re-walked REP times, so the data comes
// out of L2 / the Infinity Cache, not HBM: the number is the pipe's, not the DRAM's. Reported: cycles per instruction per CU = the wall time of
// the kernel in shader clocks x 1 / (instructions issued by one CU) — with W waves per CU streaming concurrently.
// build: hipcc --offload-arch=gfx950 -O3 -o vmem_shape tools/probes/vmem_shape.hip ; run: ./vmem_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int B, bool STORE>
__global__ __launch_bounds__(512) void probe(char* buf, int R, int reps, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int FP = 32768;                   // footprint per wave: with 4 waves per CU the 32 CUs of an XCD cover exactly its 4 MB L2
  const int L = 64 / R;                       // lanes per row
  const int stride = FP / R;                  // row stride: the R rows of an instruction are >= 512 B apart (different lines)
  const int seg = L * B;                      // contiguous bytes per row and instruction
  const int per_row = stride / seg;           // instructions that walk the footprint once
  char* base = buf + ((size_t)(blockIdx.x * (blockDim.x >> 6) + wave)) * FP + (size_t)(lane / L) * stride + (lane % L) * B;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int rep = 0; rep < reps; ++rep) {
    for (int it = 0; it < per_row; it += 8) {
      unsigned v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        char* p = base + (size_t)(it + k) * seg;
        if constexpr (STORE) {
          if constexpr (B == 16) *reinterpret_cast<u32x4*>(p) = u32x4{(unsigned)it, 1u, 2u, 3u};
          else if constexpr (B == 8) *reinterpret_cast<u32x2*>(p) = u32x2{(unsigned)it, 1u};
          else *reinterpret_cast<unsigned*>(p) = (unsigned)it;
          v[k] = 0;
        } else {
          if constexpr (B == 16) { const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); v[k] = t[0] ^ t[3]; }
          else if constexpr (B == 8) { const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)); v[k] = t[0] ^ t[1]; }
          else v[k] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p));
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) { out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2] = t1 - t0; out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = acc; }
}

int main() {
  const int cus = (getenv("CUS") ? atoi(getenv("CUS")) : 256), reps = 32;
  char* buf; unsigned long long* out;
  const size_t bytes = (size_t)cus * 8 * 32768;
  hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
  hipMalloc(&out, cus * 8 * 2 * sizeof(unsigned long long));
  printf("| access | rows x bytes/row per instr | waves/CU | loads: cycles per instr per CU | stores: cycles per instr per CU |\n|---|---|---|---|---|\n");
  struct Sh { int B, R; };
  const Sh shapes[] = {{16, 1}, {16, 2}, {16, 4}, {16, 8}, {16, 16}, {16, 32}, {16, 64}, {8, 1}, {8, 8}, {8, 16}, {8, 64}, {4, 1}, {4, 16}, {4, 64}};
  for (const Sh& s : shapes)
    for (int waves : {4, 8}) {
      double res[2];
      for (int st = 0; st < 2; ++st) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e30f;
        for (int trial = 0; trial < 3; ++trial) {
          hipEventRecord(a);
#define LAUNCH(BB, SS) hipLaunchKernelGGL((probe<BB, SS>), dim3(cus), dim3(waves * 64), 0, 0, buf, s.R, reps, out)
          if (s.B == 16) { if (st) LAUNCH(16, true); else LAUNCH(16, false); }
          else if (s.B == 8) { if (st) LAUNCH(8, true); else LAUNCH(8, false); }
          else { if (st) LAUNCH(4, true); else LAUNCH(4, false); }
          hipEventRecord(b); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        std::vector<unsigned long long> h(cus * 8 * 2);
        hipMemcpy(h.data(), out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        // in-kernel cycles of the slowest wave / instructions issued by ONE CU
        unsigned long long mx = 0; for (int i = 0; i < cus * waves; ++i) if (h[2 * i] > mx) mx = h[2 * i];
        res[st] = (double)mx / ((double)waves * (32768 / (64 * s.B)) * reps);
        (void)best;
      }
      printf("| %2d B/lane, %2d rows | %2d x %4d B | %d | %.1f | %.1f |\n", s.B, s.R, s.R, (64 / s.R) * s.B, waves, res[0], res[1]);
    }
  return 0;
}
