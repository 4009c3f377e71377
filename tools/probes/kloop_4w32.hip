// kloop_4w32.hip — VERDICT r04 next #3: a 4-wave (ONE wave per SIMD, 512-register) 256x256 GEMM K loop on v_mfma_f32_32x32x16_bf16,
// fragments register-double-buffered, ONE barrier per 32-deep k-slab, LDS-DMA stream 3 slabs (1.5 K tiles) ahead — measured against the
// production 8-wave 16x16x32 8-phase loop (2 793 cycles per 64-deep K tile at K = 512, 3 350 at K = 2048: profiles/r04_e_wg_timeline.md).
//
// Geometry: workgroup = 4 waves as 2 (M) x 2 (N), wave tile 128 x 128 = 4 x 4 MFMA blocks of 32 x 32 (256 accumulator registers).
// LDS: a ring of 4 k-slabs; a slab = A[256][32] + W[256][32] bf16 = 2 x 16 KB, rows of 64 B. One LDS-DMA instruction (1 KB) covers 16 rows:
// lane l -> row l / 4, 16-byte chunk l % 4; the chunk is XOR-swizzled with (row >> 2) & 3 on the DMA SOURCE address and again on the
// fragment read, which makes the ds_read_b128 of a 32-row fragment conflict-free (its 16-lane service groups see 16 distinct 4-bank groups).
// Operands are swapped in the MFMA (mfma(W, A)) as in the production kernels: a lane owns 4 consecutive output columns of one row.
// Phase p (k-slab p, 32 MFMAs per wave = 1 024 matrix-pipe cycles):
//   s_waitcnt vmcnt(8)            slab p+1 has landed (this wave's pieces; 8 DMA instructions per wave and slab)
//   s_barrier                     ... everybody's pieces; and everybody is done reading slab p-1
//   8 x LDS-DMA                   slab p+3 into the slot of slab p-1
//   k-step 0: 16 MFMAs on register set X | 8 ds_read_b128 -> set Y (k-step 1 of slab p)
//   k-step 1: 16 MFMAs on set Y          | 8 ds_read_b128 -> set X (k-step 0 of slab p+1: certified by this phase's barrier)
// so no LDS latency is exposed behind the barrier and a gap between two MFMAs carries at most one filler (guide: <= 5 hide).
// MODE 0: the full loop; 1: no LDS-DMA (fragments re-read from a resident slab: the loop without its stream); 2: no fragment reads either
// (MFMAs + barriers only); 3: MFMAs only (no barrier).
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/bin/kloop_4w32 tools/probes/kloop_4w32.hip && tools/probes/bin/kloop_4w32
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <type_traits>
#include <vector>
#include <algorithm>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int KS = 32;                       // k-slab depth
constexpr int SLAB = 2 * 256 * KS;           // bf16 elements: A part then W part (32 KB)
constexpr int NSLOT = 4;

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <int MODE, int PLACE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void kloop4w(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw, int K, int M, int N, bf16_t* __restrict__ C, int ldc,
             unsigned long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[NSLOT * SLAB];      // 128 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = N / 256;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nbn) * 256, n0 = (tile % nbn) * 256;
  const int ns = K / KS;                      // k-slabs
  // DMA: per slab 16 pieces of A + 16 of W (1 KB = 16 rows x 64 B each); wave w issues pieces 4 w .. 4 w + 3 of both
  const int drow = lane >> 2, dch = lane & 3;
  const bf16_t* asrc[4]; const bf16_t* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 16 + drow;
    const int sw = (dch ^ ((r >> 2) & 3)) * 8;
    asrc[i] = A + (size_t)min(m0 + r, M - 1) * lda + sw;
    wsrc[i] = W + (size_t)min(n0 + r, N - 1) * ldw + sw;
  }
  // PLACE == 2: the same requests as buffer_load ... lds — a 32-bit per-lane byte offset (loop invariant) + the slab's k offset as the
  // scalar offset: half the address bytes per request, no 64-bit address arithmetic in the loop
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)min((size_t)M * lda * 2, (size_t)0x7fffffff), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)min((size_t)N * ldw * 2, (size_t)0x7fffffff), 0x00020000);
  int avo[4], wvo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { avo[i] = (int)((asrc[i] - A) * 2); wvo[i] = (int)((wsrc[i] - W) * 2); }
  auto dma = [&](int s) {
    if (MODE >= 1 && s >= NSLOT) return;
    if (s >= ns) return;
    bf16_t* dst = smem + (s % NSLOT) * SLAB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + s * KS), (lptr_t)(dst + (wave * 4 + i) * 16 * KS), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + s * KS), (lptr_t)(dst + 256 * KS + (wave * 4 + i) * 16 * KS), 16, 0, 0);
    }
  };
  // fragment read offsets (bf16 elements inside a slab): block i of the wave's 128 rows, k-step ks, lane -> row lane % 32, chunk 2 ks + lane / 32
  const int fr = lane & 31, fh = lane >> 5;
  int aoff[4][2], boff[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ra = wm * 128 + i * 32 + fr, rb = wn * 128 + i * 32 + fr;
      aoff[i][ks] = ra * KS + (((2 * ks + fh) ^ ((ra >> 2) & 3)) << 3);
      boff[i][ks] = 256 * KS + rb * KS + (((2 * ks + fh) ^ ((rb >> 2) & 3)) << 3);
    }
  f32x16_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned long long t0 = __builtin_readcyclecounter();
  dma(0); dma(1); dma(2);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // slab 0 landed (own pieces)
  __builtin_amdgcn_s_barrier();
  bf16x8_t ax[4], bx[4], ay[4], by[4], az[4], bz[4];
  {
    const bf16_t* s0 = smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) { ax[i] = *reinterpret_cast<const bf16x8_t*>(s0 + aoff[i][0]); bx[i] = *reinterpret_cast<const bf16x8_t*>(s0 + boff[i][0]); }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  // one phase; DMA = the phase requests slab p + 3 (the last three phases of a tile do not: peeled, so that the body has no branch and
  // the scheduler sees ONE region per half: fillers pinned between the MFMAs with sched_group_barrier)
  auto phase = [&](int p, auto dma_c) {
    constexpr bool DMA = decltype(dma_c)::value;
    const bf16_t* cur = smem + ((MODE >= 1 ? 0 : p) % NSLOT) * SLAB;
    const bf16_t* nxt = smem + ((MODE >= 1 ? 0 : p + 1) % NSLOT) * SLAB;
    if (MODE <= 2) {
      if constexpr (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // slab p+1 landed; slab p+2 may still fly
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PLACE == 2) {
      // 16 groups of [filler | 2 MFMAs]: an LDS-DMA piece (buffer form) in front of every second group, one ds_read_b128 in front of each
      bf16_t* dst = smem + ((p + 3) % NSLOT) * SLAB;
      const int soff = (p + 3) * KS * 2;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if constexpr (DMA && MODE == 0) {
          if ((g & 1) == 0) {
            const int q = g >> 1, i = q >> 1;
            if (q & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t)(dst + 256 * KS + (wave * 4 + i) * 16 * KS), 16, wvo[i], soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lptr_t)(dst + (wave * 4 + i) * 16 * KS), 16, avo[i], soff, 0, 0);
          }
        }
        if (MODE <= 1) {
          if (g < 4) ay[g] = *reinterpret_cast<const bf16x8_t*>(cur + aoff[g][1]);
          else if (g < 8) by[g - 4] = *reinterpret_cast<const bf16x8_t*>(cur + boff[g - 4][1]);
          else if (g < 12) az[g - 8] = *reinterpret_cast<const bf16x8_t*>(nxt + aoff[g - 8][0]);
          else bz[g - 12] = *reinterpret_cast<const bf16x8_t*>(nxt + boff[g - 12][0]);
        }
        const int h = g & 7, i0 = (2 * h) >> 2, j0 = (2 * h) & 3;
        if (g < 8 || MODE >= 2) {
          acc[i0][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx[j0], ax[i0], acc[i0][j0], 0, 0, 0);
          acc[i0][j0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx[j0 + 1], ax[i0], acc[i0][j0 + 1], 0, 0, 0);
        } else {
          acc[i0][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by[j0], ay[i0], acc[i0][j0], 0, 0, 0);
          acc[i0][j0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by[j0 + 1], ay[i0], acc[i0][j0 + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE <= 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { ax[g] = az[g]; bx[g] = bz[g]; }
      }
      return;
    }
    if constexpr (PLACE == 1) {
      // hand placement: k-step 0 = 8 groups of [LDS-DMA piece g | ds_read of fragment g of set Y | 2 MFMAs], k-step 1 = 8 groups of
      // [ds_read of fragment g of set X' | 2 MFMAs]; every group is its own scheduling region
      bf16_t* dst = smem + ((p + 3) % NSLOT) * SLAB;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if constexpr (DMA && MODE == 0) {
          const int i = g >> 1;
          if (g & 1) __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + (p + 3) * KS), (lptr_t)(dst + 256 * KS + (wave * 4 + i) * 16 * KS), 16, 0, 0);
          else __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + (p + 3) * KS), (lptr_t)(dst + (wave * 4 + i) * 16 * KS), 16, 0, 0);
        }
        if (MODE <= 1) { if (g < 4) ay[g] = *reinterpret_cast<const bf16x8_t*>(cur + aoff[g][1]); else by[g - 4] = *reinterpret_cast<const bf16x8_t*>(cur + boff[g - 4][1]); }
        const int i0 = (2 * g) >> 2, j0 = (2 * g) & 3;
        acc[i0][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx[j0], ax[i0], acc[i0][j0], 0, 0, 0);
        acc[i0][j0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx[j0 + 1], ax[i0], acc[i0][j0 + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        // set X is overwritten fragment by fragment: fragment g of X' may only be requested once the MFMAs that read X's fragment g are issued.
        // A fragments (g < 4) are read by row i0 = g of this k-step... so request order: B' first is not possible either; use a third set instead
        if (MODE <= 1) { if (g < 4) az[g] = *reinterpret_cast<const bf16x8_t*>(nxt + aoff[g][0]); else bz[g - 4] = *reinterpret_cast<const bf16x8_t*>(nxt + boff[g - 4][0]); }
        const int i0 = (2 * g) >> 2, j0 = (2 * g) & 3;
        acc[i0][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((MODE <= 1 ? by[j0] : bx[j0]), (MODE <= 1 ? ay[i0] : ax[i0]), acc[i0][j0], 0, 0, 0);
        acc[i0][j0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((MODE <= 1 ? by[j0 + 1] : bx[j0 + 1]), (MODE <= 1 ? ay[i0] : ax[i0]), acc[i0][j0 + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE <= 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { ax[g] = az[g]; bx[g] = bz[g]; }
      }
      return;
    }
    // k-step 0 on set X; set Y <- k-step 1 of this slab; the 8 LDS-DMA requests of slab p+3
    if constexpr (DMA && MODE == 0) {
      bf16_t* dst = smem + ((p + 3) % NSLOT) * SLAB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + (p + 3) * KS), (lptr_t)(dst + (wave * 4 + i) * 16 * KS), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + (p + 3) * KS), (lptr_t)(dst + 256 * KS + (wave * 4 + i) * 16 * KS), 16, 0, 0);
      }
    }
    if (MODE <= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ay[i] = *reinterpret_cast<const bf16x8_t*>(cur + aoff[i][1]); by[i] = *reinterpret_cast<const bf16x8_t*>(cur + boff[i][1]); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx[j], ax[i], acc[i][j], 0, 0, 0);
    if (MODE <= 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (DMA && MODE == 0) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1 on set Y; set X <- k-step 0 of the next slab (landed: certified by this phase's barrier; past the last slab: stale bytes, unused)
    if (MODE <= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ax[i] = *reinterpret_cast<const bf16x8_t*>(nxt + aoff[i][0]); bx[i] = *reinterpret_cast<const bf16x8_t*>(nxt + boff[i][0]); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((MODE <= 1 ? by[j] : bx[j]), (MODE <= 1 ? ay[i] : ax[i]), acc[i][j], 0, 0, 0);
    if (MODE <= 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  int p = 0;
  for (; p + 3 < ns; ++p) phase(p, std::true_type{});
  for (; p < ns; ++p) phase(p, std::false_type{});
  const unsigned long long t2 = __builtin_readcyclecounter();
  // plain fragment-layout store (not timed as part of the loop): acc[i][j][r] = C[m = .. + lane % 32][n = .. + 8 (r / 4) + 4 (lane / 32) + r % 4]
  if (C) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 128 + i * 32 + fr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 128 + j * 32 + 8 * q + 4 * fh;
          if (m < M && n < N) {
            __bf16 v[4] = {(__bf16)acc[i][j][4 * q], (__bf16)acc[i][j][4 * q + 1], (__bf16)acc[i][j][4 * q + 2], (__bf16)acc[i][j][4 * q + 3]};
            *reinterpret_cast<uint2*>(C + (size_t)m * ldc + n) = *reinterpret_cast<const uint2*>(v);
          }
        }
      }
  }
  const unsigned long long t3 = __builtin_readcyclecounter();
  if (tid == 0 && stamps) { stamps[blockIdx.x * 4 + 0] = t0; stamps[blockIdx.x * 4 + 1] = t1; stamps[blockIdx.x * 4 + 2] = t2; stamps[blockIdx.x * 4 + 3] = t3; }
}

// ---- the same loop on 16-deep k-slabs: a ring of 8 x 16 KB, one barrier per MFMA k-step (16 MFMAs), the stream 6 slabs = 1.5 K tiles ahead
// (the 32-deep form above keeps 2 slabs = 1 K tile in flight: on long-M shapes, where the A panel streams from HBM, its counted wait stalls).
// LDS rows are 32 B: one DMA instruction covers 32 rows (lane l -> row l / 2, chunk l % 2), chunk XOR-ed with (row >> 3) & 1.
constexpr int KS16 = 16, SLAB16 = 2 * 256 * KS16, NSLOT16 = 8;
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void kloop4w16(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw, int K, int M, int N, bf16_t* __restrict__ C, int ldc,
               unsigned long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[NSLOT16 * SLAB16];      // 128 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = N / 256;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nbn) * 256, n0 = (tile % nbn) * 256;
  const int ns = K / KS16;
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)min((size_t)M * lda * 2, (size_t)0x7fffffff), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)min((size_t)N * ldw * 2, (size_t)0x7fffffff), 0x00020000);
  int avo[2], wvo[2];      // per slab: 8 pieces of A + 8 of W (1 KB = 32 rows x 32 B); wave w issues pieces 2 w, 2 w + 1 of both
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 32 + (lane >> 1);
    const int sw = ((lane & 1) ^ ((r >> 3) & 1)) * 16;
    avo[i] = (int)(((size_t)min(m0 + r, M - 1) * lda) * 2) + sw;
    wvo[i] = (int)(((size_t)min(n0 + r, N - 1) * ldw) * 2) + sw;
  }
  auto dma = [&](int s) {
    bf16_t* dst = smem + (s % NSLOT16) * SLAB16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lptr_t)(dst + (wave * 2 + i) * 32 * KS16), 16, avo[i], s * KS16 * 2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t)(dst + 256 * KS16 + (wave * 2 + i) * 32 * KS16), 16, wvo[i], s * KS16 * 2, 0, 0);
    }
  };
  const int fr = lane & 31, fh = lane >> 5;
  int aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 128 + i * 32 + fr, rb = wn * 128 + i * 32 + fr;
    aoff[i] = ra * KS16 + ((fh ^ ((ra >> 3) & 1)) << 3);
    boff[i] = 256 * KS16 + rb * KS16 + ((fh ^ ((rb >> 3) & 1)) << 3);
  }
  f32x16_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int s = 0; s < 7; ++s) if (s < ns) dma(s);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // slab 0 landed (own pieces): 6 younger slabs x 4
  __builtin_amdgcn_s_barrier();
  bf16x8_t ax[4], bx[4], ay[4], by[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { ax[i] = *reinterpret_cast<const bf16x8_t*>(smem + aoff[i]); bx[i] = *reinterpret_cast<const bf16x8_t*>(smem + boff[i]); }
  const unsigned long long t1 = __builtin_readcyclecounter();
  auto phase = [&](int p, auto dma_c) {
    constexpr bool DMA = decltype(dma_c)::value;
    const bf16_t* nxt = smem + ((MODE >= 1 ? 0 : p + 1) % NSLOT16) * SLAB16;
    if (MODE <= 2) {
      if constexpr (DMA) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");      // slab p+1 landed; p+2 .. p+6 may still fly
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16_t* dst = smem + ((p + 7) % NSLOT16) * SLAB16;
    const int soff = (p + 7) * KS16 * 2;
#pragma unroll
    for (int g = 0; g < 8; ++g) {      // 8 groups of [filler | 2 MFMAs]: a DMA piece in front of every second group, a fragment read in front of each
      if constexpr (DMA && MODE == 0) {
        if ((g & 1) == 0) {
          const int q = g >> 1, i = q >> 1;
          if (q & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t)(dst + 256 * KS16 + (wave * 2 + i) * 32 * KS16), 16, wvo[i], soff, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lptr_t)(dst + (wave * 2 + i) * 32 * KS16), 16, avo[i], soff, 0, 0);
        }
      }
      if (MODE <= 1) { if (g < 4) ay[g] = *reinterpret_cast<const bf16x8_t*>(nxt + aoff[g]); else by[g - 4] = *reinterpret_cast<const bf16x8_t*>(nxt + boff[g - 4]); }
      const int i0 = (2 * g) >> 2, j0 = (2 * g) & 3;
      acc[i0][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx[j0], ax[i0], acc[i0][j0], 0, 0, 0);
      acc[i0][j0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx[j0 + 1], ax[i0], acc[i0][j0 + 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE <= 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) { ax[g] = ay[g]; bx[g] = by[g]; }
    }
  };
  int p = 0;
  for (; p + 7 < ns; ++p) phase(p, std::true_type{});
  for (; p < ns; ++p) phase(p, std::false_type{});
  const unsigned long long t2 = __builtin_readcyclecounter();
  if (C) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 128 + i * 32 + fr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 128 + j * 32 + 8 * q + 4 * fh;
          if (m < M && n < N) {
            __bf16 v[4] = {(__bf16)acc[i][j][4 * q], (__bf16)acc[i][j][4 * q + 1], (__bf16)acc[i][j][4 * q + 2], (__bf16)acc[i][j][4 * q + 3]};
            *reinterpret_cast<uint2*>(C + (size_t)m * ldc + n) = *reinterpret_cast<const uint2*>(v);
          }
        }
      }
  }
  const unsigned long long t3 = __builtin_readcyclecounter();
  if (tid == 0 && stamps) { stamps[blockIdx.x * 4 + 0] = t0; stamps[blockIdx.x * 4 + 1] = t1; stamps[blockIdx.x * 4 + 2] = t2; stamps[blockIdx.x * 4 + 3] = t3; }
}

static uint16_t f2bf(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }

template <int MODE, int PLACE = 0, int K16 = 0>
static void run(const char* name, const bf16_t* dA, const bf16_t* dW, bf16_t* dC, unsigned long long* dS, int M, int N, int K, const std::vector<uint16_t>& hA,
                const std::vector<uint16_t>& hW, bool check) {
  const int nb = (M / 256) * (N / 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) { if (K16) hipLaunchKernelGGL((kloop4w16<MODE>), dim3(nb), dim3(256), 0, 0, dA, K, dW, K, K, M, N, dC, N, dS); else hipLaunchKernelGGL((kloop4w<MODE, PLACE>), dim3(nb), dim3(256), 0, 0, dA, K, dW, K, K, M, N, dC, N, dS); }
  hipDeviceSynchronize();
  const int reps = 5;
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) { if (K16) hipLaunchKernelGGL((kloop4w16<MODE>), dim3(nb), dim3(256), 0, 0, dA, K, dW, K, K, M, N, dC, N, dS); else hipLaunchKernelGGL((kloop4w<MODE, PLACE>), dim3(nb), dim3(256), 0, 0, dA, K, dW, K, K, M, N, dC, N, dS); }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> st(nb * 4);
  hipMemcpy(st.data(), dS, nb * 32, hipMemcpyDeviceToHost);
  std::vector<double> loop, pro, epi;
  for (int b = 0; b < nb; ++b) { pro.push_back((double)(st[b * 4 + 1] - st[b * 4])); loop.push_back((double)(st[b * 4 + 2] - st[b * 4 + 1]) / (K / 64)); epi.push_back((double)(st[b * 4 + 3] - st[b * 4 + 2])); }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto p10 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 10]; };
  auto p90 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[(v.size() * 9) / 10]; };
  const double tf = 2.0 * M * N * (double)K / (ms * 1e-3) / 1e12;
  printf("| %s | %d x %d x %d | %.3f | %.0f | %.0f (p10 %.0f, p90 %.0f) | %.0f | %.0f |", name, M, N, K, ms, tf, med(loop), p10(loop), p90(loop), med(pro), med(epi));
  if (check) {
    std::vector<uint16_t> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    for (int t = 0; t < 4000; ++t) {
      const int m = (int)((t * 2654435761u) % M), n = (int)((t * 40503u + 17) % N);
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
      const double got = bf2f(hC[(size_t)m * N + n]);
      const double e = fabs(got - ref) / fmax(1.0, fabs(ref));
      worst = fmax(worst, e); bad += e > 1e-2;
    }
    printf(" check: worst rel err %.2e, %d of 4000 beyond 1e-2 |\n", worst, bad);
  } else printf(" — |\n");
}

int main() {
  printf("| loop | M x N x K | kernel ms | TFLOP/s (whole kernel incl. the plain store) | cycles per 64-deep K tile, median over workgroups | prologue cycles | store cycles |\n|---|---|---|---|---|---|---|\n");
  const int shapes[7][3] = {{8192, 8192, 512}, {8192, 8192, 2048}, {65536, 2048, 512}, {65536, 512, 2048},      // 2, 3: FFN1 / FFN2 shapes
                            {65536, 1536, 512}, {65536, 512, 512}, {65536, 512, 1536}};                            // QKV, out-proj dX, QKV dX
  for (int t = 0; t < 7; ++t) {
    const int M = shapes[t][0], N = shapes[t][1], K = shapes[t][2];
    std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hW) v = f2bf(rnd());
    bf16_t *dA, *dW, *dC; unsigned long long* dS;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 2); hipMalloc(&dS, (size_t)(M / 256) * (N / 256) * 32);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    if (t < 2) {
      run<0>("4-wave 32x32x16, full loop (fillers pinned by sched_group_barrier)", dA, dW, dC, dS, M, N, K, hA, hW, true);
      run<0, 1>("4-wave, hand placement (global_load_lds: 1 DMA + 1 ds_read per MFMA pair of k-step 0)", dA, dW, dC, dS, M, N, K, hA, hW, true);
    }
    run<0, 2>("4-wave, hand placement, buffer_load lds (1 DMA per 4 MFMAs, 1 ds_read per 2)", dA, dW, dC, dS, M, N, K, hA, hW, true);
    run<0, 0, 1>("4-wave, 16-deep slabs: ring of 8, stream 1.5 K tiles ahead, barrier per k-step (buffer_load lds)", dA, dW, dC, dS, M, N, K, hA, hW, true);
    run<1, 0, 1>("  16-deep slabs without the LDS-DMA stream", dA, dW, dC, dS, M, N, K, hA, hW, false);
    run<1, 2>("  32-deep slabs without the LDS-DMA stream (resident slab)", dA, dW, dC, dS, M, N, K, hA, hW, false);
    run<2>("  MFMAs + barriers only", dA, dW, dC, dS, M, N, K, hA, hW, false);
    run<3>("  MFMAs only", dA, dW, dC, dS, M, N, K, hA, hW, false);
    hipFree(dA); hipFree(dW); hipFree(dC); hipFree(dS);
  }
  return 0;
}
