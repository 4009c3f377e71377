// lora.hip — the LoRA-parameter side of the step: skinny rank-r gradient GEMMs, group-lasso
// norms (L_structure), fused AdamW, and the small cast/pack helpers.
//
// References (bjzhb666/GS-LoRA): loralib.Linear autograd for lora_A/lora_B (call sites
// vit_pytorch_face/vit_face.py:330,333), engine_cl.get_structure_loss (engine_cl.py:349-432),
// util/cal_norm.get_norm_of_lora (util/cal_norm.py:4-146), torch.optim.AdamW via
// timm.create_optimizer (train/train_own_forget_cl.py:811-813).
#include "gsl_common.h"

using namespace gsl;

// =====================================================================================
// K9  G[n, j] (+)= sum_m Y[m, n] * U[m, j]      (reduction over the M = B*197 token rows)
// HBM-bound: Y is streamed exactly once with 16-byte loads; each thread owns 8 (bf16) or 4 (f32)
// consecutive n and keeps VEC x R accumulators in registers; the r-vector U[m, :] is staged per
// row chunk in LDS and read as a wave-uniform broadcast. Two deterministic stages: per-row-split
// partial slabs, then a fixed-order reduction that also applies the output strides.
// =====================================================================================
constexpr int LG_ROWS = 64;  // rows of U staged per LDS fill
constexpr int LG_FAN = 32;   // splits summed per thread in the first reduction level

template <typename T> struct LgVec;
template <> struct LgVec<bf16_t> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void ld(const bf16_t* p, float v[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
  }
};
template <> struct LgVec<f16_t> {      // fp16 operands (round 5)
  static constexpr int V = 8;
  static __device__ __forceinline__ void ld(const f16_t* p, float v[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    unpack2h(t.x, v[0], v[1]); unpack2h(t.y, v[2], v[3]); unpack2h(t.z, v[4], v[5]); unpack2h(t.w, v[6], v[7]);
  }
};
template <> struct LgVec<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void ld(const float* p, float v[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
};

template <typename T> struct LgRaw;
template <> struct LgRaw<bf16_t> {
  typedef uint4 type;
  static __device__ __forceinline__ void unpack(const uint4 t, float v[8]) {
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
  }
};
template <> struct LgRaw<f16_t> {
  typedef uint4 type;
  static __device__ __forceinline__ void unpack(const uint4 t, float v[8]) {
    unpack2h(t.x, v[0], v[1]); unpack2h(t.y, v[2], v[3]); unpack2h(t.z, v[4], v[5]); unpack2h(t.w, v[6], v[7]);
  }
};
template <> struct LgRaw<float> {
  typedef float4 type;
  static __device__ __forceinline__ void unpack(const float4 t, float v[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
};

// block = 256 threads = CG column groups x RP row phases (RP = 256 / CG in {1,2,4}); the RP phases walk
// interleaved rows of the block's row range and are combined through LDS in a fixed order.
template <typename T, int R>
__global__ __launch_bounds__(256) void lora_grad_partial_kernel(const T* __restrict__ Y, long ldy, const T* __restrict__ U, int ldu,
                                                                float* __restrict__ part, int M, int N, int rows_per_split,
                                                                int CG) {
  fp16_sat_on();
  constexpr int V = LgVec<T>::V;
  __shared__ float us[LG_ROWS][R];
  extern __shared__ __attribute__((aligned(16))) float red[];   // (RP-1) * CG * V * R floats
  const int ncol = N / V;
  const int RP = blockDim.x / CG;
  const int cgl = threadIdx.x % CG, phase = threadIdx.x / CG;
  const int cg = blockIdx.x * CG + cgl;
  const bool active = cg < ncol;
  const int split = blockIdx.y;
  const int r0 = split * rows_per_split, r1 = min(M, r0 + rows_per_split);
  float acc[V][R];
#pragma unroll
  for (int i = 0; i < V; ++i)
#pragma unroll
    for (int j = 0; j < R; ++j) acc[i][j] = 0.f;
  // Raw 16-byte (V*sizeof(T)) loads are double-buffered in registers: the loads of batch b+1 are issued before the FMAs of
  // batch b, so every thread always has LGF loads in flight (a thread's rows are otherwise a chain of exposed latencies).
  constexpr int LGF = 4;
  using raw_t = typename LgRaw<T>::type;
  const T* ycol = Y + (size_t)cg * V;
  for (int rb = r0; rb < r1; rb += LG_ROWS) {
    const int nr = min(LG_ROWS, r1 - rb);
    __syncthreads();
    for (int t = threadIdx.x; t < LG_ROWS * R; t += blockDim.x) {
      const int rr = t / R, j = t % R;
      us[rr][j] = (rr < nr) ? Elem<T>::ld(U + (size_t)(rb + rr) * ldu + j) : 0.f;
    }
    raw_t cur[LGF], nxt[LGF];
    if (active) {
#pragma unroll
      for (int k = 0; k < LGF; ++k) {
        const int r = min(phase + k * RP, nr - 1);
        cur[k] = *reinterpret_cast<const raw_t*>(ycol + (size_t)(rb + r) * ldy);
      }
    }
    __syncthreads();
    if (active) {
      for (int rr = phase; rr < nr; rr += LGF * RP) {
#pragma unroll
        for (int k = 0; k < LGF; ++k) {     // prefetch the next batch (clamped rows: harmless re-reads at the tail)
          const int r = min(rr + (LGF + k) * RP, nr - 1);
          nxt[k] = *reinterpret_cast<const raw_t*>(ycol + (size_t)(rb + r) * ldy);
        }
#pragma unroll
        for (int k = 0; k < LGF; ++k) {
          const int r = rr + k * RP;
          if (r < nr) {
            float y[V];
            LgRaw<T>::unpack(cur[k], y);
#pragma unroll
            for (int j = 0; j < R; ++j) {
              const float u = us[r][j];
#pragma unroll
              for (int i = 0; i < V; ++i) acc[i][j] = fmaf(y[i], u, acc[i][j]);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < LGF; ++k) cur[k] = nxt[k];
      }
    }
  }
  // combine the row phases: phases 1..RP-1 park their accumulators in LDS, phase 0 adds them in order
  if (RP > 1) {
    __syncthreads();
    if (phase > 0) {
      float* dst = red + ((size_t)(phase - 1) * CG + cgl) * (V * R);
#pragma unroll
      for (int i = 0; i < V; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) dst[i * R + j] = acc[i][j];
    }
    __syncthreads();
    if (phase == 0) {
      for (int ph = 1; ph < RP; ++ph) {
        const float* src = red + ((size_t)(ph - 1) * CG + cgl) * (V * R);
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
          for (int j = 0; j < R; ++j) acc[i][j] += src[i * R + j];
      }
    }
  }
  if (active && phase == 0) {
    float* p = part + ((size_t)split * N + (size_t)cg * V) * R;
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
      for (int j = 0; j < R; ++j) p[i * R + j] = acc[i][j];
  }
}

// ---- MFMA form of the partial sums (bf16 operands, N % 256 == 0): G^T-free contraction over the ROW index.
// The contraction runs over m, the slow index of both row-major operands, so the fragments cannot come straight from global
// memory; a [32 rows][256 cols] slab of Y and the matching [32][16] slab of U go through LDS (padded rows) and are gathered as
// MFMA operands with gfx950's LDS transpose read (ds_read_b64_tr_b16: lane l of a 16-lane group receives column l of a 4 x 16
// block, see attention.hip) — A = Y^T fragment (row = column n of Y), B = U fragment (column j), k-slots = the same 8 rows for both.
// One block = 4 waves x 64 columns; global loads of the next slab are in flight during the 4 MFMAs per wave of the current one.
// The VALU kernel above needs 8*R FMAs per row and thread: at r = 16 (ViT-B/16) it is VALU-bound, and at N = 512 it reaches only
// 3 TB/s; this one is bound by the Y stream alone. Writes the same part[split][n][R] layout (the reductions below are shared).
typedef short lg_v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) lg_v4s_t* lg_lds_v4s_p;
typedef __attribute__((ext_vector_type(8))) __bf16 lg_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 lg_f16x8_t;
typedef __attribute__((ext_vector_type(4))) float lg_f32x4_t;
constexpr int LGM_CN = 256, LGM_YLD = LGM_CN + 16, LGM_ULD = 32, LGM_K = 32;

// body of one workgroup: column block bxi (256 columns of Y), row split `split`; R = padded rank of the partial layout (8 or 16)
// F16: the operands are IEEE fp16 (dtype GSL_F16) instead of bf16 — the same bytes through the same LDS path, the other MFMA opcode
template <bool F16>
__device__ __forceinline__ void lgm_block(const bf16_t* __restrict__ Y, long ldy, const bf16_t* __restrict__ U, int ldu,
                                          float* __restrict__ part, int M, int N, int rows_per_split, int R, int bxi, int split) {
  __shared__ __attribute__((aligned(16))) bf16_t ys[2][LGM_K * LGM_YLD];
  __shared__ __attribute__((aligned(16))) bf16_t us[2][LGM_K * LGM_ULD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = bxi * LGM_CN;
  const int r0 = split * rows_per_split, r1 = min(M, r0 + rows_per_split);
  const int lc = tid & 31, lr = tid >> 5;
  uint4 yreg[4], ureg = make_uint4(0, 0, 0, 0);
  auto gload = [&](int rb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rb + lr + 8 * i;
      yreg[i] = (r < r1) ? *reinterpret_cast<const uint4*>(Y + (size_t)r * ldy + n0 + lc * 8) : make_uint4(0, 0, 0, 0);
    }
    if (tid < 64) {
      const int r = rb + (tid >> 1);
      ureg = (r < r1) ? *reinterpret_cast<const uint4*>(U + (size_t)r * ldu + (tid & 1) * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(&ys[buf][(lr + 8 * i) * LGM_YLD + lc * 8]) = yreg[i];
    if (tid < 64) *reinterpret_cast<uint4*>(&us[buf][(tid >> 1) * LGM_ULD + (tid & 1) * 8]) = ureg;
  };
  lg_f32x4_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = lg_f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4, i16 = lane & 15;
  const int trow = 4 * g + (i16 >> 2), tcol = (i16 & 3) * 4;       // this lane's piece of a 4 x 16 block (transpose-read address)
  gload(r0);
  lstore(0);
  __syncthreads();
  int it = 0;
  for (int rb = r0; rb < r1; rb += LGM_K, ++it) {
    const int buf = it & 1;
    const bool more = rb + LGM_K < r1;
    if (more) gload(rb + LGM_K);
    union { lg_v4s_t h[2]; lg_bf16x8_t v; lg_f16x8_t w; } bf;
    bf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lg_lds_v4s_p)(&us[buf][trow * LGM_ULD + tcol]));
    bf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lg_lds_v4s_p)(&us[buf][(trow + 16) * LGM_ULD + tcol]));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      union { lg_v4s_t h[2]; lg_bf16x8_t v; lg_f16x8_t w; } af;
      const bf16_t* p = &ys[buf][trow * LGM_YLD + wave * 64 + t * 16 + tcol];
      af.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lg_lds_v4s_p)(p));
      af.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lg_lds_v4s_p)(p + 16 * LGM_YLD));
      if constexpr (F16) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af.w, bf.w, acc[t], 0, 0, 0);
      else acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf.v, acc[t], 0, 0, 0);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }
  // acc[t][r] = G[n = n0 + wave*64 + t*16 + 4g + r][j = lane & 15]
  if (i16 < R) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wave * 64 + t * 16 + 4 * g + r;
        part[((size_t)split * N + n) * R + i16] = acc[t][r];
      }
  }
}
template <int R, bool F16>
__global__ __launch_bounds__(256) void lora_grad_mfma_kernel(const bf16_t* __restrict__ Y, long ldy, const bf16_t* __restrict__ U, int ldu,
                                                             float* __restrict__ part, int M, int N, int rows_per_split) {
  fp16_sat_on();
  lgm_block<F16>(Y, ldy, U, ldu, part, M, N, rows_per_split, R, blockIdx.x, blockIdx.y);
}

// ---- batched form: every LoRA-gradient reduction of a backward pass in two launches (the launch-bound regime: few-shot batches run 24
// of them, 48 launches of 3 - 8 us each). The descriptors travel BY VALUE in the kernel arguments, so a captured HIP graph carries them.
constexpr int LGB_MAX = 24;
struct LgbEntry {
  const bf16_t* Y; const bf16_t* U; float* G;
  long ldy, gsn, gsj, ws0;
  int ldu, M, N, r, R, acc, bx, nsplit, rps, wg0, rb0;      // wg0 / rb0: first workgroup of this entry in the partial / reduce launch
};
struct LgbArgs { int n; LgbEntry e[LGB_MAX]; };

template <bool F16>
__global__ __launch_bounds__(256) void lora_grad_batch_partial_kernel(const LgbArgs a, float* __restrict__ ws) {
  fp16_sat_on();
  int d = 0;
  for (int k = 1; k < a.n; ++k) d = ((int)blockIdx.x >= a.e[k].wg0) ? k : d;
  const LgbEntry& e = a.e[d];
  const int local = (int)blockIdx.x - e.wg0;
  lgm_block<F16>(e.Y, e.ldy, e.U, e.ldu, ws + e.ws0, e.M, e.N, e.rps, e.R, local % e.bx, local / e.bx);
}
// fixed-order sum of an entry's splits, output strides (+ accumulate) applied on the way out
__global__ __launch_bounds__(256) void lora_grad_batch_reduce_kernel(const LgbArgs a, const float* __restrict__ ws, const float* __restrict__ gscale) {
  fp16_sat_on();
  int d = 0;
  for (int k = 1; k < a.n; ++k) d = ((int)blockIdx.x >= a.e[k].rb0) ? k : d;
  const LgbEntry& e = a.e[d];
  const int idx = ((int)blockIdx.x - e.rb0) * 256 + (int)threadIdx.x;
  if (idx >= e.N * e.R) return;
  const int n = idx / e.R, j = idx - n * e.R;
  if (j >= e.r) return;
  const float* p = ws + e.ws0 + idx;
  const size_t NR = (size_t)e.N * e.R;
  float s = 0.f;
  for (int k = 0; k < e.nsplit; ++k) s += p[(size_t)k * NR];
  if (gscale) s *= gscale[1];      // loss-scaled backward (fp16 operands): divide the power-of-two scale out, exactly
  float* g = e.G + (size_t)n * e.gsn + (size_t)j * e.gsj;
  *g = e.acc ? (*g + s) : s;
}
static inline void lgm_plan(int M, int N, int& bx, int& nsplit, int& rps) {
  bx = N / LGM_CN;
  int target = 1024 / bx;                       // ~4 blocks per CU in total
  if (target < 1) target = 1;
  int steps = (M + LGM_K - 1) / LGM_K;          // 32-row slabs
  nsplit = steps < target ? steps : target;
  if (steps <= 64 && nsplit > LG_FAN) nsplit = LG_FAN;      // few rows (launch-bound regime): one slab of partials, no first reduction level
  rps = ((steps + nsplit - 1) / nsplit) * LGM_K;
  nsplit = (M + rps - 1) / rps;
}
static inline bool lgm_usable(int M, int N, int ldu, int dtype) { return dtype != GSL_F32 && (N % LGM_CN) == 0 && ldu >= 16 && M >= 64; }

// level 1: thread (idx, sb) sums LG_FAN consecutive splits  -> part2[sb][idx]
__global__ __launch_bounds__(256) void lora_grad_reduce1_kernel(const float* __restrict__ part, float* __restrict__ part2,
                                                                int NR, int nsplit) {
  fp16_sat_on();
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NR) return;
  const int s0 = blockIdx.y * LG_FAN, s1 = min(nsplit, s0 + LG_FAN);
  float s = 0.f;
  for (int sp = s0; sp < s1; ++sp) s += part[(size_t)sp * NR + idx];
  part2[(size_t)blockIdx.y * NR + idx] = s;
}
// level 2: fixed-order sum of the level-1 slabs, apply output strides (+ accumulate)
template <int R>
__global__ __launch_bounds__(256) void lora_grad_reduce2_kernel(const float* __restrict__ part2, float* G, long gsn, long gsj,
                                                                int N, int r, int nslab, int accumulate, const float* __restrict__ gscale) {
  fp16_sat_on();
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * R) return;
  const int n = idx / R, j = idx % R;
  if (j >= r) return;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += part2[(size_t)k * N * R + idx];
  if (gscale) s *= gscale[1];
  float* g = G + (size_t)n * gsn + (size_t)j * gsj;
  *g = accumulate ? (*g + s) : s;
}

// single-launch, fixed-order reduction of the MFMA kernel's partials: a block owns 64 outputs, its 4 waves each sum every 4th
// split, the 4 sub-sums are combined through LDS in wave order (deterministic), output strides (+ accumulate) applied on the way out
template <int R>
__global__ __launch_bounds__(256) void lora_grad_reduce_kernel(const float* __restrict__ part, float* G, long gsn, long gsj, int N, int r,
                                                               int nsplit, int accumulate, const float* __restrict__ gscale) {
  fp16_sat_on();
  __shared__ float sm[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + l, NR = N * R;
  float s = 0.f;
  if (idx < NR)
    for (int sp = w; sp < nsplit; sp += 4) s += part[(size_t)sp * NR + idx];
  sm[w][l] = s;
  __syncthreads();
  if (w == 0 && idx < NR) {
    float t = ((sm[0][l] + sm[1][l]) + sm[2][l]) + sm[3][l];
    if (gscale) t *= gscale[1];
    const int n = idx / R, j = idx % R;
    if (j < r) {
      float* g = G + (size_t)n * gsn + (size_t)j * gsj;
      *g = accumulate ? (*g + t) : t;
    }
  }
}

static inline void lg_plan(int M, int N, int V, int R, int& CG, int& bx, int& nsplit, int& rps) {
  const int ncol = N / V;
  CG = ncol >= 256 ? 256 : ((ncol > 64 || R > 8) ? 128 : 64);   // keeps the phase-combine LDS <= 48 KB
  bx = (ncol + CG - 1) / CG;
  int target = 768 / bx;                    // ~3 blocks per CU in total
  if (target < 1) target = 1;
  nsplit = (M + 127) / 128;                 // at least 128 rows per split
  if (nsplit > target) nsplit = target;
  if (nsplit < 1) nsplit = 1;
  rps = (M + nsplit - 1) / nsplit;
  nsplit = (M + rps - 1) / rps;
}

extern "C" long gsl_lora_grad_ws_elems(int M, int N, int r) {
  const int R = (r <= 8) ? 8 : 16;
  long best = 0;
  for (int V = 4; V <= 8; V += 4) {
    int CG, bx, nsplit, rps;
    lg_plan(M, N, V, R, CG, bx, nsplit, rps);
    const long slabs = (long)nsplit + (nsplit + LG_FAN - 1) / LG_FAN;
    if (slabs > best) best = slabs;
  }
  if ((N % LGM_CN) == 0) {
    int bx, nsplit, rps;
    lgm_plan(M, N, bx, nsplit, rps);
    const long slabs = (long)nsplit + (nsplit + LG_FAN - 1) / LG_FAN;
    if (slabs > best) best = slabs;
  }
  return best * (long)N * R;
}

extern "C" int gsl_lora_grad(const void* Y, long ldy, const void* U, int ldu, float* G, long gsn, long gsj, int M, int N, int r,
                             int dtype, int accumulate, float* ws, const float* gscale, gsl_stream_t s) {
  GSL_CHECK_ARG(Y && U && G && ws && M > 0 && N > 0 && ldy >= N, "null/size");
  GSL_CHECK_ARG(r >= 1 && r <= 16 && ldu >= 16 && (ldu % 8) == 0, "r in [1,16], ldu >= 16 (zero-padded)");
  GSL_CHECK_ARG(dtype == GSL_F32 || dtype == GSL_BF16 || dtype == GSL_F16, "dtype");
  const int V = (dtype != GSL_F32) ? 8 : 4;
  GSL_CHECK_ARG((N % V) == 0, "N must be a multiple of the vector width");
  hipStream_t st = as_stream(s);
  const int R = (r <= 8) ? 8 : 16;
  int CG, bx, nsplit, rps;
  {
#ifdef GSL_DEV
    const char* ev = getenv("GSL_LORA_GRAD_MFMA");      // development knob (dev library only): 0 forces the VALU kernel
    const bool want = !ev || atoi(ev) != 0;
#else
    const bool want = true;
#endif
    if (want && lgm_usable(M, N, ldu, dtype) && (reinterpret_cast<uintptr_t>(U) % 16) == 0 && (reinterpret_cast<uintptr_t>(Y) % 16) == 0 &&
        ((size_t)ldu * 2) % 16 == 0 && ((size_t)ldy * 2) % 16 == 0) {
      lgm_plan(M, N, bx, nsplit, rps);
#define GSL_LGM(RR, FF) hipLaunchKernelGGL((lora_grad_mfma_kernel<RR, FF>), dim3(bx, nsplit), dim3(256), 0, st, (const bf16_t*)Y, ldy, (const bf16_t*)U, ldu, ws, M, N, rps)
      if (dtype == GSL_F16) { if (R == 8) GSL_LGM(8, true); else GSL_LGM(16, true); }
      else { if (R == 8) GSL_LGM(8, false); else GSL_LGM(16, false); }
#undef GSL_LGM
      int rc0 = check_launch("gsl_lora_grad(mfma partial)");
      if (rc0) return rc0;
      const int totm = N * R;
      if (nsplit <= 160 && totm >= 8192) {     // few splits, many outputs: one launch (measured 166 -> 162 us at N = 2048; at N = 512 the two-level form wins)
        if (R == 8) hipLaunchKernelGGL(lora_grad_reduce_kernel<8>, dim3((totm + 63) / 64), dim3(256), 0, st, ws, G, gsn, gsj, N, r, nsplit, accumulate, gscale);
        else hipLaunchKernelGGL(lora_grad_reduce_kernel<16>, dim3((totm + 63) / 64), dim3(256), 0, st, ws, G, gsn, gsj, N, r, nsplit, accumulate, gscale);
        return check_launch("gsl_lora_grad(reduce)");
      }
      float* part2m = ws + (size_t)nsplit * N * R;
      int nslabm = (nsplit + LG_FAN - 1) / LG_FAN;
      if (nslabm == 1) { part2m = ws; nslabm = nsplit; }      // one slab: the second level sums the splits itself, in the same order (one launch less)
      else hipLaunchKernelGGL(lora_grad_reduce1_kernel, dim3((totm + 255) / 256, nslabm), dim3(256), 0, st, ws, part2m, totm, nsplit);
      if (R == 8) hipLaunchKernelGGL(lora_grad_reduce2_kernel<8>, dim3((totm + 255) / 256), dim3(256), 0, st, part2m, G, gsn, gsj, N, r, nslabm, accumulate, gscale);
      else hipLaunchKernelGGL(lora_grad_reduce2_kernel<16>, dim3((totm + 255) / 256), dim3(256), 0, st, part2m, G, gsn, gsj, N, r, nslabm, accumulate, gscale);
      return check_launch("gsl_lora_grad(reduce)");
    }
  }
  lg_plan(M, N, V, R, CG, bx, nsplit, rps);
  const int RP = 256 / CG;
  const size_t red_bytes = (size_t)(RP - 1) * CG * V * R * sizeof(float);
  float* part2 = ws + (size_t)nsplit * N * R;
#define LAUNCH(TT, RR)                                                                                                   \
  hipLaunchKernelGGL((lora_grad_partial_kernel<TT, RR>), dim3(bx, nsplit), dim3(256), red_bytes, st, (const TT*)Y, ldy, (const TT*)U, \
                     ldu, ws, M, N, rps, CG)
  if (dtype == GSL_BF16) { if (R == 8) LAUNCH(bf16_t, 8); else LAUNCH(bf16_t, 16); }
  else if (dtype == GSL_F16) { if (R == 8) LAUNCH(f16_t, 8); else LAUNCH(f16_t, 16); }
  else { if (R == 8) LAUNCH(float, 8); else LAUNCH(float, 16); }
#undef LAUNCH
  int rc = check_launch("gsl_lora_grad(partial)");
  if (rc) return rc;
  const int tot = N * R;
  int nslab = (nsplit + LG_FAN - 1) / LG_FAN;
  if (nslab == 1) { part2 = ws; nslab = nsplit; }
  else hipLaunchKernelGGL(lora_grad_reduce1_kernel, dim3((tot + 255) / 256, nslab), dim3(256), 0, st, ws, part2, tot, nsplit);
  if (R == 8) hipLaunchKernelGGL(lora_grad_reduce2_kernel<8>, dim3((tot + 255) / 256), dim3(256), 0, st, part2, G, gsn, gsj, N, r, nslab, accumulate, gscale);
  else hipLaunchKernelGGL(lora_grad_reduce2_kernel<16>, dim3((tot + 255) / 256), dim3(256), 0, st, part2, G, gsn, gsj, N, r, nslab, accumulate, gscale);
  return check_launch("gsl_lora_grad(reduce)");
}

// =====================================================================================
// K12 group-lasso norms over the flat LoRA buffer.  Stage 1: block (t, s) reduces split s of
// tensor t with 16-byte loads + wavefront reductions -> partial[t][s]. Stage 2 (one wave):
// fixed-order sums -> per-tensor sumsq, per-group lasso norm, cal_norm, total loss, mask.
// =====================================================================================
__global__ __launch_bounds__(256) void gnorm_partial_kernel(const float* __restrict__ flat, const int64_t* __restrict__ toff,
                                                            const int64_t* __restrict__ tnumel, float* __restrict__ partial) {
  fp16_sat_on();
  __shared__ float sm[16];
  const int t = blockIdx.x, sp = blockIdx.y;
  const int64_t n = tnumel[t];
  const float* p = flat + toff[t];
  const int64_t chunk = ((n + GSL_NORM_SPLIT - 1) / GSL_NORM_SPLIT + 3) & ~(int64_t)3;
  const int64_t b = sp * chunk, e = min(n, b + chunk);
  float acc = 0.f;
  const bool al = ((reinterpret_cast<uintptr_t>(p + b) & 15) == 0);
  if (al) {
    const int64_t n4 = (e > b) ? (e - b) / 4 : 0;
    const float4* p4 = reinterpret_cast<const float4*>(p + b);
    for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
      const float4 v = p4[i];
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t i = b + n4 * 4 + threadIdx.x; i < e; i += blockDim.x) acc += p[i] * p[i];
  } else {
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) acc += p[i] * p[i];
  }
  const float tot = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[t * GSL_NORM_SPLIT + sp] = tot;
}

__global__ void gnorm_final_kernel(const float* __restrict__ partial, const int32_t* __restrict__ tgroup, int ntensors,
                                   int ngroups, float tau, float* tensor_sumsq, float* group_norm, float* cal_norm,
                                   float* loss, uint8_t* mask) {
  fp16_sat_on();
  // one wave; lane g owns group g (looped), everything in fixed order -> bit-reproducible
  const int lane = threadIdx.x;
  for (int t = lane; t < ntensors; t += 64) {
    float s = 0.f;
    for (int k = 0; k < GSL_NORM_SPLIT; ++k) s += partial[t * GSL_NORM_SPLIT + k];
    tensor_sumsq[t] = s;
  }
  __syncthreads();
  for (int g = lane; g < ngroups; g += 64) {
    float ss = 0.f, cn = 0.f;
    for (int t = 0; t < ntensors; ++t)
      if (tgroup[t] == g) { ss += tensor_sumsq[t]; cn += sqrtf(tensor_sumsq[t]); }
    const float nrm = sqrtf(ss);
    group_norm[g] = nrm;
    if (cal_norm) cal_norm[g] = cn;
    if (mask) mask[g] = nrm > tau ? 1 : 0;
  }
  __syncthreads();
  if (lane == 0 && loss) {
    float l = 0.f;
    for (int g = 0; g < ngroups; ++g) l += group_norm[g];
    loss[0] = l;
  }
}

static int lgb_plan(const gsl_lgrad_desc* descs, int n, LgbArgs* out, long* ws_elems, int* nwg, int* nrb) {
  long ws = 0; int wg = 0, rb = 0;
  // row splits: the launch as a whole should fill the chip about four times over (1024 workgroups), not every entry on its own
  long sum_bx = 0;
  for (int k = 0; k < n; ++k) sum_bx += descs[k].N > 0 ? descs[k].N / LGM_CN : 0;
  const int split_target = (int)max(1L, min(32L, 1024 / max(1L, sum_bx)));
  for (int k = 0; k < n; ++k) {
    const gsl_lgrad_desc& d = descs[k];
    GSL_CHECK_ARG(d.Y && d.U && d.G && d.M > 0 && d.N > 0 && d.ldy >= d.N, "null/size");
    GSL_CHECK_ARG(d.r >= 1 && d.r <= 16 && d.ldu >= 16 && (d.ldu % 8) == 0 && (d.ldy % 8) == 0 && (d.N % LGM_CN) == 0, "r in [1,16], ldu >= 16, N % 256 == 0, 16-byte rows");
    GSL_CHECK_ARG((reinterpret_cast<uintptr_t>(d.Y) % 16) == 0 && (reinterpret_cast<uintptr_t>(d.U) % 16) == 0, "16-byte aligned operands");
    const int bx = d.N / LGM_CN, steps = (d.M + LGM_K - 1) / LGM_K;
    int nsplit = min(steps, split_target);
    const int rps = ((steps + nsplit - 1) / nsplit) * LGM_K;
    nsplit = (d.M + rps - 1) / rps;
    const int R = d.r <= 8 ? 8 : 16;
    if (out) {
      LgbEntry& e = out->e[k];
      e.Y = (const bf16_t*)d.Y; e.U = (const bf16_t*)d.U; e.G = d.G; e.ldy = d.ldy; e.gsn = d.gsn; e.gsj = d.gsj; e.ws0 = ws;
      e.ldu = d.ldu; e.M = d.M; e.N = d.N; e.r = d.r; e.R = R; e.acc = d.accumulate; e.bx = bx; e.nsplit = nsplit; e.rps = rps;
      e.wg0 = wg; e.rb0 = rb;
    }
    ws += (long)nsplit * d.N * R;
    wg += bx * nsplit;
    rb += (d.N * R + 255) / 256;
  }
  if (out) out->n = n;
  if (ws_elems) *ws_elems = ws;
  if (nwg) *nwg = wg;
  if (nrb) *nrb = rb;
  return GSL_OK;
}

extern "C" long gsl_lora_grad_batch_ws_elems(const gsl_lgrad_desc* descs, int n) {
  long total = 0;
  for (int k0 = 0; k0 < n; k0 += LGB_MAX) {
    long ws = 0;
    if (lgb_plan(descs + k0, min(LGB_MAX, n - k0), nullptr, &ws, nullptr, nullptr) != GSL_OK) return -1;
    if (ws > total) total = ws;
  }
  return total;
}

// n LoRA-gradient reductions G_k (+)= Y_k^T U_k (bf16 operands, see gsl_lora_grad) in two launches per 24 descriptors. descs is a HOST
// array: the launch carries the descriptors by value (HIP-graph capture keeps them). The G_k of one call must not overlap.
extern "C" int gsl_lora_grad_batch(const gsl_lgrad_desc* descs, int n, float* ws, int dtype, const float* gscale, gsl_stream_t s) {
  GSL_CHECK_ARG(descs && n > 0 && ws, "null/size");
  GSL_CHECK_ARG(dtype == GSL_BF16 || dtype == GSL_F16, "dtype: bf16 or fp16 operands");
  hipStream_t st = as_stream(s);
  for (int k0 = 0; k0 < n; k0 += LGB_MAX) {
    LgbArgs a;
    int nwg = 0, nrb = 0;
    const int rc = lgb_plan(descs + k0, min(LGB_MAX, n - k0), &a, nullptr, &nwg, &nrb);
    if (rc) return rc;
    if (dtype == GSL_F16) hipLaunchKernelGGL(lora_grad_batch_partial_kernel<true>, dim3(nwg), dim3(256), 0, st, a, ws);
    else hipLaunchKernelGGL(lora_grad_batch_partial_kernel<false>, dim3(nwg), dim3(256), 0, st, a, ws);
    hipLaunchKernelGGL(lora_grad_batch_reduce_kernel, dim3(nrb), dim3(256), 0, st, a, (const float*)ws, gscale);
    const int rc2 = check_launch("gsl_lora_grad_batch");
    if (rc2) return rc2;
  }
  return GSL_OK;
}

extern "C" int gsl_group_norms_fwd(const float* flat, const int64_t* toff, const int64_t* tnumel, const int32_t* tgroup,
                                   int ntensors, int ngroups, float tau, float* partial_ws, float* tensor_sumsq,
                                   float* group_norm, float* cal_norm, float* loss, uint8_t* mask, gsl_stream_t s) {
  GSL_CHECK_ARG(flat && toff && tnumel && tgroup && partial_ws && tensor_sumsq && group_norm, "null");
  GSL_CHECK_ARG(ntensors > 0 && ngroups > 0, "counts");
  hipStream_t st = as_stream(s);
  hipLaunchKernelGGL(gnorm_partial_kernel, dim3(ntensors, GSL_NORM_SPLIT), dim3(256), 0, st, flat, toff, tnumel, partial_ws);
  int rc = check_launch("gsl_group_norms_fwd(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(gnorm_final_kernel, dim3(1), dim3(64), 0, st, partial_ws, tgroup, ntensors, ngroups, tau, tensor_sumsq,
                     group_norm, cal_norm, loss, mask);
  return check_launch("gsl_group_norms_fwd(final)");
}

__global__ __launch_bounds__(256) void gnorm_bwd_kernel(const float* __restrict__ flat, const int64_t* __restrict__ toff,
                                                        const int64_t* __restrict__ tnumel, const int32_t* __restrict__ tgroup,
                                                        const float* __restrict__ group_norm, const float* __restrict__ coef,
                                                        float scale, float* gradflat) {
  fp16_sat_on();
  const int t = blockIdx.x;
  const float nrm = group_norm[tgroup[t]];
  const float k = (nrm > 0.f) ? (coef[0] * scale / nrm) : 0.f;   // subgradient 0 at an all-zero group
  const int64_t n = tnumel[t], o = toff[t];
  for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.y * blockDim.x)
    gradflat[o + i] += k * flat[o + i];
}

extern "C" int gsl_group_norms_bwd(const float* flat, const int64_t* toff, const int64_t* tnumel, const int32_t* tgroup,
                                   int ntensors, const float* group_norm, const float* coef, float scale, float* gradflat,
                                   gsl_stream_t s) {
  GSL_CHECK_ARG(flat && toff && tnumel && tgroup && group_norm && coef && gradflat && ntensors > 0, "null");
  hipLaunchKernelGGL(gnorm_bwd_kernel, dim3(ntensors, GSL_NORM_SPLIT), dim3(256), 0, as_stream(s), flat, toff, tnumel, tgroup,
                     group_norm, coef, scale, gradflat);
  return check_launch("gsl_group_norms_bwd");
}

// =====================================================================================
// K14 fused AdamW (single launch over the flat LoRA bucket; 4 streams x 0.98 MB)
// =====================================================================================
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, const float* __restrict__ guard) {
  fp16_sat_on();
  if (guard && !(*guard < 65504.0f)) return;      // a gradient of this step was clipped by a saturating fp16 store (or is not finite): skip the update
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);          // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * b2 + gi * gi * (1.0f - b2);          // mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// HIP-graph form: the step count and the learning rate live in device memory (a captured graph replays with fresh values)
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, const float* __restrict__ lr_dev, float b1,
                                                        float b2, float eps, float wd, const int64_t* __restrict__ step_dev,
                                                        const float* __restrict__ guard) {
  fp16_sat_on();
  if (guard && !(*guard < 65504.0f)) return;
  const double t = (double)*step_dev;
  const float lr = *lr_dev;
  const float bc1 = (float)(1.0 - pow((double)b1, t)), bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}
extern "C" int gsl_adamw_flat_dev(float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float beta1,
                                  float beta2, float eps, float wd, const int64_t* step_dev, const float* guard, gsl_stream_t s) {
  GSL_CHECK_ARG(p && g && m && v && n > 0 && lr_dev && step_dev, "null/size");
  const int grid = (int)min((n + 255) / 256, (long)(256 * 8));
  hipLaunchKernelGGL(adamw_dev_kernel, dim3(grid), dim3(256), 0, as_stream(s), p, g, m, v, n, lr_dev, beta1, beta2, eps, wd, step_dev, guard);
  return check_launch("gsl_adamw_flat_dev");
}

extern "C" int gsl_adamw_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                              float eps, float wd, int step, const float* guard, gsl_stream_t s) {
  GSL_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "null/size/step");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int grid = (int)min((n + 255) / 256, (long)(256 * 8));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, as_stream(s), p, g, m, v, n, lr, beta1, beta2, eps, wd, (float)bc1,
                     (float)sqrt(bc2), guard);
  return check_launch("gsl_adamw_flat");
}

// =====================================================================================
// casts / packing
// =====================================================================================
template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, long n) {
  fp16_sat_on();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) Elem<T>::st(out + i, in[i]);
}
extern "C" int gsl_cast(const float* in, void* out, long n, int dtype, gsl_stream_t s) {
  GSL_CHECK_ARG(in && out && n > 0, "null/size");
  const int grid = (int)min((n + 255) / 256, (long)(256 * 16));
  if (dtype == GSL_BF16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(s), in, (bf16_t*)out, n);
  else if (dtype == GSL_F16) hipLaunchKernelGGL(cast_kernel<f16_t>, dim3(grid), dim3(256), 0, as_stream(s), in, (f16_t*)out, n);
  else if (dtype == GSL_F32) hipLaunchKernelGGL(cast_kernel<float>, dim3(grid), dim3(256), 0, as_stream(s), in, (float*)out, n);
  else return fail(GSL_ERR_ARG, "gsl_cast: bad dtype%s %ld", "", dtype);
  return check_launch("gsl_cast");
}

template <typename T>
__global__ void transpose_cast_kernel(const float* __restrict__ in, T* __restrict__ out, int R, int C) {
  fp16_sat_on();
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (c < C && r < R) Elem<T>::st(out + (size_t)c * R + r, tile[threadIdx.x][i]);
  }
}
extern "C" int gsl_transpose_cast(const float* in, void* out, int R, int C, int dtype, gsl_stream_t s) {
  GSL_CHECK_ARG(in && out && R > 0 && C > 0, "null/size");
  dim3 grid((C + 31) / 32, (R + 31) / 32), blk(32, 8);
  if (dtype == GSL_BF16) hipLaunchKernelGGL(transpose_cast_kernel<bf16_t>, grid, blk, 0, as_stream(s), in, (bf16_t*)out, R, C);
  else if (dtype == GSL_F16) hipLaunchKernelGGL(transpose_cast_kernel<f16_t>, grid, blk, 0, as_stream(s), in, (f16_t*)out, R, C);
  else if (dtype == GSL_F32) hipLaunchKernelGGL(transpose_cast_kernel<float>, grid, blk, 0, as_stream(s), in, (float*)out, R, C);
  else return fail(GSL_ERR_ARG, "gsl_transpose_cast: bad dtype%s %ld", "", dtype);
  return check_launch("gsl_transpose_cast");
}

template <typename T>
__global__ void pack_pad_kernel(const float* __restrict__ in, long si, long sj, int rows, int cols, float scale, T* __restrict__ out,
                                int rows_out, int ld_out) {
  fp16_sat_on();
  const long tot = (long)rows_out * ld_out;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (long)gridDim.x * blockDim.x) {
    const int i = (int)(idx / ld_out), j = (int)(idx % ld_out);
    const float v = (i < rows && j < cols) ? scale * in[(size_t)i * si + (size_t)j * sj] : 0.f;
    Elem<T>::st(out + idx, v);
  }
}
extern "C" int gsl_pack_pad(const float* in, long si, long sj, int rows, int cols, float scale, void* out, int rows_out,
                            int ld_out, int dtype, gsl_stream_t s) {
  GSL_CHECK_ARG(in && out && rows > 0 && cols > 0 && rows_out >= rows && ld_out >= cols, "null/size");
  const long tot = (long)rows_out * ld_out;
  const int grid = (int)min((tot + 255) / 256, (long)(256 * 8));
  if (dtype == GSL_BF16) hipLaunchKernelGGL(pack_pad_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(s), in, si, sj, rows, cols, scale, (bf16_t*)out, rows_out, ld_out);
  else if (dtype == GSL_F16) hipLaunchKernelGGL(pack_pad_kernel<f16_t>, dim3(grid), dim3(256), 0, as_stream(s), in, si, sj, rows, cols, scale, (f16_t*)out, rows_out, ld_out);
  else if (dtype == GSL_F32) hipLaunchKernelGGL(pack_pad_kernel<float>, dim3(grid), dim3(256), 0, as_stream(s), in, si, sj, rows, cols, scale, (float*)out, rows_out, ld_out);
  else return fail(GSL_ERR_ARG, "gsl_pack_pad: bad dtype%s %ld", "", dtype);
  return check_launch("gsl_pack_pad");
}

// all LoRA operand packs of a step in ONE launch: blockIdx.y selects the descriptor (device-resident table built once by the host)
template <typename T>
__global__ void pack_pad_batch_kernel(const gsl_pack_desc* __restrict__ descs) {
  fp16_sat_on();
  const gsl_pack_desc d = descs[blockIdx.y];
  const long tot = (long)d.rows_out * d.ld_out;
  T* out = reinterpret_cast<T*>(d.out);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (long)gridDim.x * blockDim.x) {
    const int i = (int)(idx / d.ld_out), j = (int)(idx % d.ld_out);
    const float v = (i < d.rows && j < d.cols) ? d.scale * d.in[(size_t)i * d.si + (size_t)j * d.sj] : 0.f;
    Elem<T>::st(out + idx, v);
  }
}
extern "C" int gsl_pack_pad_batch(const gsl_pack_desc* descs_dev, int n, long max_elems, int dtype, gsl_stream_t s) {
  GSL_CHECK_ARG(descs_dev && n > 0 && max_elems > 0, "null/size");
  const dim3 grid((unsigned)min((max_elems + 255) / 256, (long)64), (unsigned)n);
  if (dtype == GSL_BF16) hipLaunchKernelGGL(pack_pad_batch_kernel<bf16_t>, grid, dim3(256), 0, as_stream(s), descs_dev);
  else if (dtype == GSL_F16) hipLaunchKernelGGL(pack_pad_batch_kernel<f16_t>, grid, dim3(256), 0, as_stream(s), descs_dev);
  else if (dtype == GSL_F32) hipLaunchKernelGGL(pack_pad_batch_kernel<float>, grid, dim3(256), 0, as_stream(s), descs_dev);
  else return fail(GSL_ERR_ARG, "gsl_pack_pad_batch: bad dtype%s %ld", "", dtype);
  return check_launch("gsl_pack_pad_batch");
}

__global__ void dropout_mask_kernel(uint8_t* keep, long n, DropCfg d) {
  fp16_sat_on();
  resolve_drop(d);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    keep[i] = drop_mul(d, (uint64_t)i) != 0.f ? 1 : 0;
}
extern "C" int gsl_dropout_mask(uint8_t* keep, long n, float p_drop, uint64_t seed, uint32_t site, gsl_stream_t s) {
  GSL_CHECK_ARG(keep && n > 0 && p_drop >= 0.f && p_drop < 1.f, "null/size/p");
  const int grid = (int)min((n + 255) / 256, (long)(256 * 8));
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid), dim3(256), 0, as_stream(s), keep, n, make_drop(p_drop, seed, site));
  return check_launch("gsl_dropout_mask");
}

// =====================================================================================
// library-wide
// =====================================================================================
namespace gsl { thread_local char g_err[512] = {0}; }
extern "C" int gsl_version(void) { return 100; }
extern "C" const char* gsl_last_error(void) { return gsl::g_err; }
