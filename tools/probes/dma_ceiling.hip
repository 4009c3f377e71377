// dma_ceiling.hip — what bounds the L2 -> LDS operand stream of the 256x256x64 GEMM K loop.
// The probe replays the LDS-DMA stream of gemm_bf16_p8_kernel (same tiles, same XCD remap, same 128-byte row segments with the XOR chunk
// swizzle on the source address, 16 KB half-tiles, two barriers per phase) WITHOUT the GEMM around it, and then adds the K loop's other
// consumers back one at a time:
//   mode 0  DMA only            mode 1  + 16 MFMAs per phase and wave (register operands)      mode 2  + the 24 ds_read_b128 per K tile and wave
// swept over: issuing waves per workgroup (8 x 2 pieces, 4 x 4, 2 x 8, 1 x 16 per half-tile), half-tiles in flight (AHEAD 1..7 = 16..112 KB),
// cache policy bits (aux 0 / nt / sc1), and the source pattern (GEMM rows: 128-byte runs `ld` apart; "tiled": each half-tile one contiguous
// 16 KB run, what a pre-tiled operand layout would give).
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O2 -w -o tools/probes/bin/dma_ceiling tools/probes/dma_ceiling.hip && tools/probes/bin/dma_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int BK = 64, HT = 128 * BK;      // half-tile: 128 rows x 64 bf16 = 16 KB

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <int NW, int AHEAD, int AUX, int MODE, bool TILED, int NMF = 16>
__global__ __launch_bounds__(512) void probe(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw, int K, int M, int N,
                                             unsigned long long* stamps, float* sink) {
  constexpr int PPW = 16 / NW;                 // 1 KB pieces a loader wave issues per half-tile
  __shared__ __attribute__((aligned(16))) bf16_t smem[8 * HT];      // ring of 8 half-tile slots = 128 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbn = N / 256;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nbn) * 256, n0 = (tile % nbn) * 256;
  const int nk = K / BK, NH = 4 * nk;
  const int lrow = lane >> 3, lc = lane & 7;
  // half-tile h of the stream: K tile h / 4, piece h % 4 in the order B-h0, A-h0, B-h1, A-h1 (as the GEMM)
  auto issue = [&](int h) {
    if (h >= NH || wave >= NW) return;
    const int kt = h >> 2, piece = h & 3, isA = piece & 1, half = piece >> 1;
    bf16_t* dst = smem + (h & 7) * HT;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int rb = wave * PPW + i;           // 8-row block of the 128-row half-tile
      const int r = rb * 8 + lrow;
      const bf16_t* src;
      if constexpr (TILED) {                   // half-tile = one contiguous 16 KB run (tile-major operand)
        const size_t ht_id = ((size_t)(isA ? tile / nbn : tile % nbn) * 2 + half) * nk + kt;
        src = (isA ? A : W) + ht_id * HT + (size_t)r * BK + lc * 8;
      } else {
        const int g = isA ? min(m0 + half * 128 + r, M - 1) : min(n0 + half * 128 + r, N - 1);
        src = (isA ? A + (size_t)g * lda : W + (size_t)g * ldw) + kt * BK + ((lc ^ (r & 7)) << 3);
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + rb * 8 * BK), 16, 0, AUX);
    }
  };
  f32x4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.001f * (lane + i)); fb[i] = (__bf16)(0.002f * (lane - i)); }
  const int fr = lane & 15, fc = lane >> 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int h = 0; h < AHEAD; ++h) issue(h);
  for (int h = 0; h < NH; ++h) {
    issue(h + AHEAD);
    if (wave < NW) {
      if (h + AHEAD < NH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD * PPW > 63 ? 63 : AHEAD * PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (MODE == 2) {                 // the K loop's fragment reads of this half-tile: 6 ds_read_b128 per wave and phase
      const bf16_t* s = smem + (h & 7) * HT;
      bf16x8_t f[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int row = ((wave & 3) * 32 + q * 16 + fr) & 127;
        f[q] = *reinterpret_cast<const bf16x8_t*>(s + row * BK + ((((q & 1) * 4 + fc) ^ (row & 7)) << 3));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[q % 6], f[(q + 1) % 6], acc[q & 7], 0, 0, 0);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int q = 0; q < NMF; ++q) acc[q & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[q & 7], 0, 0, 0);
    } else if constexpr (MODE == 3) {          // VALU work of the same length instead of MFMAs (NMF x 16 cycles per SIMD pair ~ NMF x 6 v_fma per wave)
#pragma unroll
      for (int q = 0; q < NMF * 6; ++q) acc[q & 7][q & 3] = fmaf(acc[q & 7][q & 3], 1.0001f, 0.5f);
    } else if constexpr (MODE == 4) {          // the waves just sleep for about the MFMA time of a phase (NMF x 32 cycles)
      __builtin_amdgcn_s_sleep(NMF / 2);
    }
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) sink[0] = s + (float)smem[tid];
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

struct Ctx { bf16_t *A, *W; unsigned long long* st; float* sink; int M, N, K; };

template <int NW, int AHEAD, int AUX, int MODE, bool TILED, int NMF = 16>
void run(const Ctx& c, const char* label) {
  const int grid = (c.M / 256) * (c.N / 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() { hipLaunchKernelGGL((probe<NW, AHEAD, AUX, MODE, TILED, NMF>), dim3(grid), dim3(512), 0, 0, c.A, c.K, c.W, c.K, c.K, c.M, c.N, c.st, c.sink); };
  launch(); launch();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), c.st, grid * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  double mean = 0; for (auto v : h) mean += v; mean /= grid;
  const double bytes_wg = (double)(c.K / BK) * 65536.0;
  printf("| %-28s | %d x %2d | %3d KB | %d | %s | %d | %7.1f | %6.2f | %7.0f | %7.0f | %5.1f | %5.1f |\n", label, NW, 16 / NW, AHEAD * 16, AUX, TILED ? "tiled" : "rows", MODE,
         ms * 1e3, bytes_wg * grid / (ms * 1e-3) / 1e12, mean / (c.K / BK), (double)h[grid / 2] / (c.K / BK), 65536.0 / (mean / (c.K / BK)), bytes_wg * grid / (ms * 1e-3) / 256 / 1e9);
}

int main(int argc, char** argv) {
  Ctx c; c.M = 201728; 
  hipMalloc(&c.st, 65536 * 8); hipMalloc(&c.sink, 64);
  for (int shape = 0; shape < 2; ++shape) {
    c.N = shape ? 512 : 2048; c.K = shape ? 2048 : 512;
    const size_t ea = (size_t)c.M * c.K + 65536, ew = (size_t)c.N * c.K + 65536;
    hipMalloc(&c.A, ea * 2); hipMalloc(&c.W, ew * 2);
    hipMemset(c.A, 0x3c, ea * 2); hipMemset(c.W, 0x3c, ew * 2);
    printf("\nshape M = %d, N = %d, K = %d (%s): %d tiles of 256 x 256, %d K tiles of 64 KB each\n\n", c.M, c.N, c.K, shape ? "FFN2 forward" : "FFN1 / QKV class", (c.M / 256) * (c.N / 256), c.K / BK);
    printf("| variant | loader waves x pieces | in flight | aux | source | mode | launch us | L2->LDS TB/s (wall) | cycles / K tile (mean) | (median) | B/clk/CU (stamps) | GB/s/CU (wall) |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n");
    if (argc > 1 && argv[1][0] == '2') {       // second sweep: how the DMA stream and the waves' own work overlap
      run<8, 3, 0, 0, false>(c, "DMA only (48 KB)");
      run<8, 3, 0, 1, false, 4>(c, "+ 4 MFMA / phase");
      run<8, 3, 0, 1, false, 8>(c, "+ 8 MFMA / phase");
      run<8, 3, 0, 1, false, 16>(c, "+ 16 MFMA / phase");
      run<8, 3, 0, 1, false, 32>(c, "+ 32 MFMA / phase");
      run<8, 3, 0, 3, false, 8>(c, "+ 48 v_fma / phase");
      run<8, 3, 0, 3, false, 16>(c, "+ 96 v_fma / phase");
      run<8, 3, 0, 4, false, 16>(c, "+ s_sleep 8 / phase");
      run<8, 3, 0, 4, false, 32>(c, "+ s_sleep 16 / phase");
      run<4, 3, 0, 1, false, 16>(c, "4 loaders + 16 MFMA");
      run<4, 3, 0, 4, false, 16>(c, "4 loaders + s_sleep 8");
      run<2, 3, 0, 1, false, 16>(c, "2 loaders + 16 MFMA");
      run<1, 3, 0, 1, false, 16>(c, "1 loader + 16 MFMA");
    } else {
    run<8, 7, 0, 0, false>(c, "GEMM stream, DMA only");
    run<8, 7, 0, 1, false>(c, "+ MFMAs");
    run<8, 7, 0, 2, false>(c, "+ MFMAs + ds_reads");
    run<8, 5, 0, 0, false>(c, "in flight 80 KB");
    run<8, 3, 0, 0, false>(c, "in flight 48 KB");
    run<8, 2, 0, 0, false>(c, "in flight 32 KB");
    run<8, 1, 0, 0, false>(c, "in flight 16 KB");
    run<8, 3, 0, 1, false>(c, "48 KB + MFMAs");
    run<8, 3, 0, 2, false>(c, "48 KB + MFMAs + ds_reads");
    run<4, 7, 0, 0, false>(c, "4 loader waves");
    run<4, 7, 0, 1, false>(c, "4 loader waves + MFMAs");
    run<4, 7, 0, 2, false>(c, "4 loader waves + MFMAs + rd");
    run<2, 7, 0, 0, false>(c, "2 loader waves");
    run<2, 7, 0, 2, false>(c, "2 loader waves + MFMAs + rd");
    run<1, 3, 0, 0, false>(c, "1 loader wave");
    run<8, 7, 2, 0, false>(c, "aux nt");
    run<8, 7, 16, 0, false>(c, "aux sc1");
    run<8, 7, 1, 0, false>(c, "aux sc0");
    run<8, 7, 0, 0, true>(c, "tiled source, DMA only");
    run<8, 7, 0, 2, true>(c, "tiled + MFMAs + ds_reads");
    run<4, 7, 0, 2, true>(c, "tiled, 4 loaders, full");
    }
    hipFree(c.A); hipFree(c.W);
  }
  return 0;
}
