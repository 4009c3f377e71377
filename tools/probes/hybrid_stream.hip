// hybrid_stream.hip — would the 256x256x64 K loop be better off with ONE operand through the LDS-DMA and the other straight into registers?
// profiles/r04_dma_ceiling.md: the LDS-DMA stream alone tops out at 32 B/clk/CU = exactly the K loop's demand with both operands staged.
// This probe runs a lock-step mini-GEMM (one barrier pair per K tile, all 8 waves in phase: NOT the production schedule, only a like-for-like
// comparison) in two forms:
//   FULL    A (256 x 64) and W (256 x 64) through the LDS-DMA, 24 ds_read_b128 per wave and K tile               (64 KB DMA / K tile)
//   HYBRID  A through the LDS-DMA, each wave's own W fragment (64 n x 64 k) by 8 global_load_dwordx4 into registers, double-buffered;
//           16 ds_read_b128 per wave and K tile                                                                    (32 KB DMA + 64 KB loads / K tile)
// and each in the modes  0 = operand streams only   1 = + the 64 MFMAs per wave and K tile   2 = + the fragment reads from LDS.
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O2 -w -o tools/probes/bin/hybrid_stream tools/probes/hybrid_stream.hip && tools/probes/bin/hybrid_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include <vector>
#include <algorithm>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ int g_krot;   // host-set: sibling N-tile j starts its K loop at K tile j * g_krot (mod nk), as the production kernel's krot
constexpr int BK = 64, HT = 128 * BK;      // half-tile: 128 rows x 64 bf16 = 16 KB

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// HYB: 0 = FULL, 1 = HYBRID.  AHEAD = K tiles of DMA in flight.  WSHARE: 1 = the two waves of a column (same W fragment) each load HALF of it
// and exchange nothing (mode 0 only: bytes through the vector path halved, what an LDS-free exchange would need is not modelled).
template <int HYB, int AHEAD, int MODE>
__global__ __launch_bounds__(512) void probe(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw, int K, int M, int N,
                                             unsigned long long* stamps, float* sink) {
  constexpr int SLOT = (HYB ? 2 : 4) * HT;     // bf16 elements per K tile in LDS
  constexpr int NSLOT = HYB ? 4 : 2;           // 128 KB ring either way
  static_assert(AHEAD < NSLOT || (!HYB && AHEAD == 1), "ring depth");
  __shared__ __attribute__((aligned(16))) bf16_t smem[NSLOT * SLOT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;     // wave tile 128 x 64
  const int nbn = N / 256;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nbn) * 256, n0 = (tile % nbn) * 256;
  const int nk = K / BK;
  const int lrow = lane >> 3, lc = lane & 7;
  const int fr = lane & 15, fc = lane >> 4;
  const int krot = (g_krot * (tile % nbn)) % nk;
  auto issue_dma = [&](int kt) {               // every wave issues 2 pieces of each 16 KB half-tile
    if (kt >= nk) return;
    bf16_t* dst = smem + (kt % NSLOT) * SLOT;
    const int kslot = kt;
    kt = (kt + krot) % nk;
#pragma unroll
    for (int p = 0; p < (HYB ? 2 : 4); ++p) {  // half-tiles: HYB: A-h0, A-h1; FULL: W-h0, A-h0, W-h1, A-h1
      const int isA = HYB ? 1 : (p & 1), half = HYB ? p : (p >> 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rb = wave * 2 + i, r = rb * 8 + lrow;
        const int g = isA ? min(m0 + half * 128 + r, M - 1) : min(n0 + half * 128 + r, N - 1);
        const bf16_t* src = (isA ? A + (size_t)g * lda : W + (size_t)g * ldw) + kt * BK + ((lc ^ (r & 7)) << 3);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + p * HT + rb * 8 * BK), 16, 0, 0);
      }
    }
  };
  bf16x8_t wf[8];                               // HYBRID: this wave's W fragment of ONE K tile ([ks][j]: 8 x 16 bytes); the next tile's half is
  const bf16_t* wbase = W + (size_t)(n0 + wn * 64 + fr) * ldw + fc * 8;   // requested into the same registers right after its MFMAs were issued
  auto issue_w = [&](int kt, int ks) {
    if (kt >= nk) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[ks * 4 + j] = *reinterpret_cast<const bf16x8_t*>(wbase + (size_t)j * 16 * ldw + kt * BK + ks * 32);
  };
  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t ca, cb;
  for (int i = 0; i < 8; ++i) { ca[i] = (__bf16)(0.001f * (lane + i)); cb[i] = (__bf16)(0.002f * (lane - i)); }

  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int h = 0; h < AHEAD; ++h) issue_dma(h);
  if constexpr (HYB) { issue_w(0, 0); issue_w(0, 1); }
  for (int kt = 0; kt < nk; ++kt) {
    issue_dma(kt + AHEAD);
    // younger than what this K tile needs from the DMA: FULL: AHEAD tiles x 8 DMA; HYBRID: this body's 4 DMA + W(kt, 1) (4 loads)
    if (kt + AHEAD < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HYB ? 8 : AHEAD * 8) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bf16_t* s = smem + (kt % NSLOT) * SLOT;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (MODE == 0) {
        if constexpr (HYB) {
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(wf[ks * 4 + j]));
        }
      } else {
        bf16x8_t fa[8], fb[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if constexpr (MODE == 2) {
            const int row = i * 16 + fr;       // wave row wm: half-tile wm of A
            const bf16_t* ah = s + (HYB ? wm : 2 * wm + 1) * HT;
            fa[i] = *reinterpret_cast<const bf16x8_t*>(ah + row * BK + (((ks * 4 + fc) ^ (row & 7)) << 3));
          } else fa[i] = ca;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (HYB) fb[j] = wf[ks * 4 + j];
          else if constexpr (MODE == 2) {
            const int row = (wn & 1) * 64 + j * 16 + fr;
            const bf16_t* wh = s + ((wn >> 1) * 2) * HT;
            fb[j] = *reinterpret_cast<const bf16x8_t*>(wh + row * BK + (((ks * 4 + fc) ^ (row & 7)) << 3));
          } else fb[j] = cb;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      if constexpr (HYB) issue_w(kt + 1, ks);
    }
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (sum == 123.456f) sink[0] = sum + (float)smem[tid];
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

// MODE 3 / 4 (FULL operands only): ONE barrier per K tile — wait for this wave's own DMA pieces of tile kt, barrier (tile kt visible, everyone done
// reading tile kt-1), request tile kt+1 into the other slot, then the tile's fragment reads and MFMAs. PRE = 1: both k-steps' fragments are
// requested before the first MFMA (96 fragment registers); PRE = 0: per k-step.  STAG = 1: wave row 1 runs half a K tile behind row 0 (its
// k-step order is 1, 0 ... so that after the barrier one row of each SIMD pair reads while the other still has MFMAs queued).
template <int PRE, int STAG>
__global__ __launch_bounds__(512) void probe1b(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw, int K, int M, int N,
                                               unsigned long long* stamps, float* sink) {
  constexpr int SLOT = 4 * HT;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * SLOT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nbn = N / 256;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nbn) * 256, n0 = (tile % nbn) * 256;
  const int nk = K / BK;
  const int lrow = lane >> 3, lc = lane & 7;
  const int fr = lane & 15, fc = lane >> 4;
  auto issue_dma = [&](int kt) {
    if (kt >= nk) return;
    bf16_t* dst = smem + (kt & 1) * SLOT;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int isA = p & 1, half = p >> 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rb = wave * 2 + i, r = rb * 8 + lrow;
        const int g = isA ? min(m0 + half * 128 + r, M - 1) : min(n0 + half * 128 + r, N - 1);
        const bf16_t* src = (isA ? A + (size_t)g * lda : W + (size_t)g * ldw) + kt * BK + ((lc ^ (r & 7)) << 3);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + p * HT + rb * 8 * BK), 16, 0, 0);
      }
    }
  };
  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto rd_a = [&](const bf16_t* s, int i, int ks) {
    const int row = i * 16 + fr;
    return *reinterpret_cast<const bf16x8_t*>(s + (2 * wm + 1) * HT + row * BK + (((ks * 4 + fc) ^ (row & 7)) << 3));
  };
  auto rd_w = [&](const bf16_t* s, int j, int ks) {
    const int row = (wn & 1) * 64 + j * 16 + fr;
    return *reinterpret_cast<const bf16x8_t*>(s + ((wn >> 1) * 2) * HT + row * BK + (((ks * 4 + fc) ^ (row & 7)) << 3));
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  issue_dma(0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_dma(kt + 1);
    const bf16_t* s = smem + (kt & 1) * SLOT;
    if constexpr (PRE) {
      bf16x8_t fa[2][8], fb[2][4];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[ks][j] = rd_w(s, j, ks);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[ks][i] = rd_a(s, i, ks);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ks = STAG ? (kk ^ wm) : kk;
        bf16x8_t fa[8], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = rd_w(s, j, ks);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = rd_a(s, i, ks);
        if (STAG) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (STAG) __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (sum == 123.456f) sink[0] = sum + (float)smem[tid];
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

struct Ctx { bf16_t *A, *W; unsigned long long* st; float* sink; int M, N, K, lda, ldw; };

template <int HYB, int AHEAD, int MODE>
void run(const Ctx& c, const char* label) {
  const int grid = (c.M / 256) * (c.N / 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    if constexpr (MODE >= 3) hipLaunchKernelGGL((probe1b<(MODE == 4), (MODE == 5)>), dim3(grid), dim3(512), 0, 0, c.A, c.lda, c.W, c.ldw, c.K, c.M, c.N, c.st, c.sink);
    else hipLaunchKernelGGL((probe<HYB, AHEAD, MODE>), dim3(grid), dim3(512), 0, 0, c.A, c.lda, c.W, c.ldw, c.K, c.M, c.N, c.st, c.sink);
  };
  launch(); launch();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), c.st, grid * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  double mean = 0; for (auto v : h) mean += v; mean /= grid;
  const double flops = 2.0 * c.M * c.N * c.K;
  printf("| %-34s | %s | %d | %d | %7.1f | %7.0f | %7.0f | %6.1f |\n", label, HYB ? "hybrid" : "full", AHEAD, MODE, ms * 1e3, mean / (c.K / BK), (double)h[grid / 2] / (c.K / BK),
         MODE ? flops / (ms * 1e-3) / 1e12 : 0.0);
}

int main() {
  Ctx c; c.M = 201728;
  { int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_krot), &z, sizeof(int)); }
  hipMalloc(&c.st, 65536 * 8); hipMalloc(&c.sink, 64);
  for (int shape = 0; shape < 2; ++shape) {
    c.N = shape ? 512 : 2048; c.K = shape ? 2048 : 512;
    c.lda = c.K; c.ldw = c.K;
    const size_t ea = (size_t)c.M * (c.K + 512) + 65536, ew = (size_t)c.N * (c.K + 512) + 65536;
    hipMalloc(&c.A, ea * 2); hipMalloc(&c.W, ew * 2);
    hipMemset(c.A, 0x3c, ea * 2); hipMemset(c.W, 0x3c, ew * 2);
    printf("\nshape M = %d, N = %d, K = %d (%s)\n\n", c.M, c.N, c.K, shape ? "FFN2 forward" : "FFN1 / QKV class");
    printf("| variant | operands | K tiles ahead | mode | launch us | cycles / K tile (mean) | (median) | TFLOP/s |\n|---|---|---|---|---|---|---|---|\n");
    run<0, 1, 0>(c, "full: streams only");
    run<0, 1, 1>(c, "full: + MFMAs");
    run<0, 1, 2>(c, "full: + MFMAs + fragment reads");
    run<0, 1, 3>(c, "full, 1 barrier / K tile");
    run<0, 1, 4>(c, "full, 1 barrier, fragments up front");
    run<0, 1, 5>(c, "full, 1 barrier, rows staggered");
    run<1, 1, 0>(c, "hybrid: streams only");
    run<1, 2, 0>(c, "hybrid: streams only, 2 ahead");
    run<1, 3, 0>(c, "hybrid: streams only, 3 ahead");
    run<1, 1, 1>(c, "hybrid: + MFMAs");
    run<1, 2, 1>(c, "hybrid: + MFMAs, 2 ahead");
    run<1, 2, 2>(c, "hybrid: + MFMAs + A reads, 2 ahead");
    run<1, 3, 2>(c, "hybrid: + MFMAs + A reads, 3 ahead");
    for (int pad : {64, 128, 192, 256, 320}) {      // A rows padded: does the row pitch (a power of two) cost the stream anything?
      c.lda = c.K + pad;
      char lab[64]; snprintf(lab, 64, "streams only, A pitch K + %d", pad);
      run<0, 1, 0>(c, lab);
      snprintf(lab, 64, "1 barrier staggered, pitch K + %d", pad);
      run<0, 1, 5>(c, lab);
    }
    c.lda = c.K;
    for (int pad : {64, 128, 320}) {                // W rows padded (W is L2-resident: does its power-of-two pitch alias L2 channels?)
      c.ldw = c.K + pad;
      char lab[64]; snprintf(lab, 64, "streams only, W pitch K + %d", pad);
      run<0, 1, 0>(c, lab);
    }
    c.lda = c.K + 64; c.ldw = c.K + 64;
    run<0, 1, 0>(c, "streams only, A and W pitch K + 64");
    run<0, 1, 5>(c, "1 barrier staggered, A and W pitch K + 64");
    c.lda = c.K; c.ldw = c.K;
    hipFree(c.A); hipFree(c.W);
  }
  // how the K = 2048 stream depends on the number of sibling N-tiles that share an A row panel through the XCD's L2 (HBM-side bytes per K tile)
  printf("\nK = 2048, streams only, by N (sibling tiles per A panel = N / 256)\n\n| variant | operands | K tiles ahead | mode | launch us | cycles / K tile (mean) | (median) | TFLOP/s |\n|---|---|---|---|---|---|---|---|\n");
  for (int kr : {1, 4, 16}) {
    c.N = 512; c.K = 2048; c.M = 201728; c.lda = c.K; c.ldw = c.K;
    hipMalloc(&c.A, ((size_t)c.M * c.K + 65536) * 2); hipMalloc(&c.W, ((size_t)c.N * c.K + 65536) * 2);
    hipMemset(c.A, 0x3c, ((size_t)c.M * c.K + 65536) * 2); hipMemset(c.W, 0x3c, ((size_t)c.N * c.K + 65536) * 2);
    hipMemcpyToSymbol(HIP_SYMBOL(g_krot), &kr, sizeof(int));
    char lab[64]; snprintf(lab, 64, "N = 512, sibling K offset %d", kr);
    run<0, 1, 0>(c, lab);
    run<0, 1, 5>(c, "  (1-barrier staggered loop; no rotation there)");
    hipFree(c.A); hipFree(c.W);
  }
  { int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_krot), &z, sizeof(int)); }
  for (int n : {256, 512, 1024, 2048}) {
    c.N = n; c.K = 2048; c.M = n >= 1024 ? 50432 : 201728; c.lda = c.K; c.ldw = c.K;
    hipMalloc(&c.A, ((size_t)c.M * c.K + 65536) * 2); hipMalloc(&c.W, ((size_t)c.N * c.K + 65536) * 2);
    hipMemset(c.A, 0x3c, ((size_t)c.M * c.K + 65536) * 2); hipMemset(c.W, 0x3c, ((size_t)c.N * c.K + 65536) * 2);
    char lab[64]; snprintf(lab, 64, "N = %d, M = %d", n, c.M);
    run<0, 1, 0>(c, lab);
    hipFree(c.A); hipFree(c.W);
  }
  return 0;
}
