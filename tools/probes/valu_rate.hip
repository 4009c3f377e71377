// valu_rate.hip — issue cost of the instruction classes the GEMM epilogues are made of (gfx950), one probe per class:
// every wave runs ITER x 64 independent instructions of one kind between two s_memtime stamps, with 1 / 2 / 4 waves per SIMD.
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/probes/valu_rate.hip && /tmp/valu_rate
// Output: cycles per instruction per WAVE and per SIMD (= per wave / waves per SIMD). profiles/r04_valu_rate.md holds a run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
enum { OP_FMA, OP_PKFMA, OP_PKMUL, OP_PKADD, OP_EXP, OP_RCP, OP_MULLO, OP_CVTBF, OP_PERM, OP_CNDMASK, OP_AND, OP_MED3, OP_CVTU8, OP_LDSGATHER, OP_LDSLIN,
       OP_MIX_FMA_EXP, OP_ADD3, OP_LSHLOR, OP_BFE, OP_CMPCND, OP_MULF, OP_N };
static const char* NAMES[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32", "v_mul_lo_u32", "v_cvt_pk_bf16_f32",
                              "v_perm_b32", "v_cndmask_b32", "v_and_b32", "v_med3_f32", "v_cvt_pk_u8_f32", "ds_read_b32 (random gather, 10 KB table)",
                              "ds_read_b32 (lane-linear)", "3 v_fma + 1 v_exp mix", "v_add3_u32", "v_lshl_or_b32", "v_bfe_u32", "v_cmp_lt_u32 + v_cndmask", "v_mul_f32"};

template <int OP>
__global__ void probe(uint64_t* out, float* sink, int iters) {
  __shared__ uint32_t tab[2560];
  for (int i = threadIdx.x; i < 2560; i += blockDim.x) tab[i] = i * 2654435761u;
  __syncthreads();
  float a[8], b = 1.0001f + threadIdx.x * 1e-7f, c = 0.5f;
  f2 p[8], pb = {b, b}, pc = {c, c};
  uint32_t u[8];
  for (int i = 0; i < 8; ++i) { a[i] = 0.001f * (threadIdx.x + i); p[i] = (f2){a[i], a[i] + 1.f}; u[i] = threadIdx.x * 747796405u + i * 2891336453u; }
  uint32_t addr[8];
  for (int i = 0; i < 8; ++i) addr[i] = ((u[i] >> 8) % 2560u) * 4u;
  uint32_t lin = (threadIdx.x & 63) * 4u;
  uint64_t t0 = __builtin_readcyclecounter();
  t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == OP_FMA) {
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));)
    } else if constexpr (OP == OP_MULF) {
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                        "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
    } else if constexpr (OP == OP_PKFMA || OP == OP_PKMUL || OP == OP_PKADD) {
#define PK3(op) REP8(asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
                        op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n" \
                        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(pb), "v"(pc));)
#define PK2(op) REP8(asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
                        op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n" \
                        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(pb));)
      if constexpr (OP == OP_PKFMA) { PK3("v_pk_fma_f32") } else if constexpr (OP == OP_PKMUL) { PK2("v_pk_mul_f32") } else { PK2("v_pk_add_f32") }
    } else if constexpr (OP == OP_EXP || OP == OP_RCP) {
#define UN1(op) REP8(asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n" \
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
      if constexpr (OP == OP_EXP) { UN1("v_exp_f32") } else { UN1("v_rcp_f32") }
    } else if constexpr (OP == OP_MIX_FMA_EXP) {
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %3, %3\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_exp_f32 %7, %7\n"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));)
    } else if constexpr (OP == OP_MULLO || OP == OP_AND) {
#define UB2(op) REP8(asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
                        op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n" \
                        : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(0x85EBCA77u));)
      if constexpr (OP == OP_MULLO) { UB2("v_mul_lo_u32") } else { UB2("v_and_b32") }
    } else if constexpr (OP == OP_BFE || OP == OP_PERM || OP == OP_ADD3 || OP == OP_LSHLOR) {
#define UB3(op, k1, k2) REP8(asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
                        op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n" \
                        : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(k1), "v"(k2));)
      if constexpr (OP == OP_BFE) { UB3("v_bfe_u32", 3u, 16u) } else if constexpr (OP == OP_PERM) { UB3("v_perm_b32", 0x12345678u, 0x07060302u) }
      else if constexpr (OP == OP_ADD3) { UB3("v_add3_u32", 12345u, 777u) } else { UB3("v_lshl_or_b32", 3u, 5u) }
    } else if constexpr (OP == OP_CVTBF) {
      REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                        "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
    } else if constexpr (OP == OP_CVTU8) {
      REP8(asm volatile("v_cvt_pk_u8_f32 %0, %8, 0, %0\n v_cvt_pk_u8_f32 %1, %8, 1, %1\n v_cvt_pk_u8_f32 %2, %8, 2, %2\n v_cvt_pk_u8_f32 %3, %8, 3, %3\n"
                        "v_cvt_pk_u8_f32 %4, %8, 0, %4\n v_cvt_pk_u8_f32 %5, %8, 1, %5\n v_cvt_pk_u8_f32 %6, %8, 2, %6\n v_cvt_pk_u8_f32 %7, %8, 3, %7\n"
                        : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(b));)
    } else if constexpr (OP == OP_MED3) {
      REP8(asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                        "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));)
    } else if constexpr (OP == OP_CNDMASK) {
      REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "vcc");)
    } else if constexpr (OP == OP_CMPCND) {
      REP8(asm volatile("v_cmp_lt_u32 vcc, %8, %0\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_u32 vcc, %8, %1\n v_cndmask_b32 %1, %1, %8, vcc\n"
                        "v_cmp_lt_u32 vcc, %8, %2\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_u32 vcc, %8, %3\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(0x1999u) : "vcc");)
    } else if constexpr (OP == OP_LDSGATHER) {
      REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %9\n ds_read_b32 %2, %10\n ds_read_b32 %3, %11\n"
                        "ds_read_b32 %4, %12\n ds_read_b32 %5, %13\n ds_read_b32 %6, %14\n ds_read_b32 %7, %15\n s_waitcnt lgkmcnt(0)\n"
                        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
                        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]) : "memory");)
    } else if constexpr (OP == OP_LDSLIN) {
      REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                        "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7]) : "v"(lin) : "memory");)
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f; uint32_t su = 0;
  for (int i = 0; i < 8; ++i) { s += a[i] + p[i].x + p[i].y; su += u[i]; }
  if (s == 12345.678f || su == 0x12345u) sink[0] = s + tab[su % 2560];
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(uint64_t* d_out, float* d_sink) {
  const int iters = 200;
  for (int threads : {256, 512, 1024}) {
    const int wps = threads / 256;
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, d_out, d_sink, iters);      // warm-up
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, d_out, d_sink, iters);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(256 * threads / 64);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; uint64_t mx = 0;
    for (auto v : h) { mean += v; mx = v > mx ? v : mx; }
    mean /= h.size();
    const double per_wave = mean / (iters * 64.0);
    printf("| %-42s | %d | %.2f | %.2f |\n", NAMES[OP], wps, per_wave, per_wave / wps);
  }
}
template <int OP> void run_all(uint64_t* o, float* s) { run<OP>(o, s); if constexpr (OP + 1 < OP_N) run_all<OP + 1>(o, s); }

int main() {
  uint64_t* d_out; float* d_sink;
  hipMalloc(&d_out, 256 * 16 * 8); hipMalloc(&d_sink, 64);
  printf("| instruction (64 independent per iteration, 8 registers) | waves / SIMD | cycles / instr / wave | cycles / instr / SIMD |\n|---|---|---|---|\n");
  run_all<0>(d_out, d_sink);
  return 0;
}
