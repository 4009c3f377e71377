// norm.hip — LayerNorm forward / backward-dx (nn.LayerNorm eps 1e-5, affine; reference
// vit_pytorch_face/vit_face.py:316-323 PreNorm and :498-500 mlp_head). gamma/beta are frozen in
// GS-LoRA, so backward produces dx only.
//
// HBM-bound: one wave64 per row, the whole row lives in registers (D = 64*NPL), mean and the
// centred variance are wavefront reductions (two-pass, as torch does), 16-byte loads when the
// per-lane count is a multiple of 4. Backward fuses the residual-gradient add and emits the
// operand-dtype copy (with the consumer's dropout mask applied) that the next dX GEMM reads.
#include "gsl_common.h"

using namespace gsl;

template <int NPL>
struct RowIO {
  static constexpr int VEC = (NPL % 4 == 0) ? 4 : 1;
  static constexpr int NV = NPL / VEC;
  // element index of (chunk c, sub i) for this lane
  static __device__ __forceinline__ int idx(int lane, int c, int i) { return (c * 64 + lane) * VEC + i; }
  template <typename T>
  static __device__ __forceinline__ void load(const T* row, int lane, float v[NPL]) {
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      if constexpr (VEC == 4) Elem<T>::ld4(row + (c * 64 + lane) * 4, &v[c * 4]);
      else v[c] = Elem<T>::ld(row + c * 64 + lane);
    }
  }
  template <typename T>
  static __device__ __forceinline__ void store(T* row, int lane, const float v[NPL]) {
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      if constexpr (VEC == 4) Elem<T>::st4(row + (c * 64 + lane) * 4, &v[c * 4]);
      else Elem<T>::st(row + c * 64 + lane, v[c]);
    }
  }
};

// X: element type of the residual stream (f32; bf16 when the speed mode carries the forward stream in bf16)
template <int NPL, typename T, typename X>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const X* __restrict__ x, long xs, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M) {
  fp16_sat_on();
  constexpr int D = NPL * 64;
  using IO = RowIO<NPL>;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  float g[NPL], b[NPL];
  IO::load(gamma, lane, g);
  IO::load(beta, lane, b);
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += gridDim.x * wpb) {
    float v[NPL];
    IO::load(x + (size_t)row * xs, lane, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) s += v[i];
    const float mu = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { v[i] -= mu; q += v[i] * v[i]; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = v[i] * rs * g[i] + b[i];
    if (y) IO::store(y + (size_t)row * D, lane, v);      // (y == nullptr: row statistics only — timing experiments of a consumer-side LayerNorm)
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// S: element type of the residual-GRADIENT stream (dres in, dx out): f32, or bf16 in speed mode (the stream is re-read and re-written by
// every LayerNorm backward of the chain: 2 x 413 MB per call at M = 201 728 in f32)
// X: element type of the saved forward residual stream x. cls_T > 0: dres holds only the rows of the cls tokens ([M / cls_T] rows, ios
// apart); row m receives dres[m / cls_T] when m % cls_T == 0 and nothing otherwise (the stream gradient that leaves the cls-row-only
// backward of the last block is exactly zero on every other row: no zero-filled dense tensor is written or read).
template <int NPL, typename T, typename S, typename X>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const X* __restrict__ x, long xs,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const S* dres,
                                                     S* dx, long ios, T* __restrict__ dxb, int M, DropCfg drop,
                                                     long drop_row_stride, int cls_T, float* gmax) {
  fp16_sat_on();
  resolve_drop(drop);
  constexpr int D = NPL * 64;
  using IO = RowIO<NPL>;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  float g[NPL];
  IO::load(gamma, lane, g);
  float am = 0.f;      // overflow guard: largest |dy| read and |dx| stored by this wave (gmax != nullptr)
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += gridDim.x * wpb) {
    float xv[NPL], gy[NPL];
    IO::load(x + (size_t)row * xs, lane, xv);
    IO::load(dy + (size_t)row * D, lane, gy);
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    if (gmax) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) am = fmaxf(am, fabsf(gy[i]));
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      xv[i] = (xv[i] - mu) * rs;   // x-hat
      gy[i] *= g[i];               // dL/dx-hat
      s1 += gy[i];
      s2 += gy[i] * xv[i];
    }
    const float c1 = wave_sum(s1) * (1.0f / D), c2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < NPL; ++i) gy[i] = rs * (gy[i] - c1 - xv[i] * c2);
    if (dres && (cls_T == 0 || (row % cls_T) == 0)) {
      float r[NPL];
      IO::load(dres + (size_t)(cls_T ? row / cls_T : row) * ios, lane, r);
#pragma unroll
      for (int i = 0; i < NPL; ++i) gy[i] += r[i];
    }
    if (gmax) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) am = fmaxf(am, fabsf(gy[i]));
    }
    IO::store(dx + (size_t)row * (cls_T ? (long)D : ios), lane, gy);
    if (dxb) {
      if (drop.thr) {
#pragma unroll
        for (int c = 0; c < IO::NV; ++c) {
          if constexpr (IO::VEC == 4) {
            float dm[4];
            drop_mul4(drop, (uint64_t)row * drop_row_stride + IO::idx(lane, c, 0), dm);
#pragma unroll
            for (int i = 0; i < 4; ++i) gy[c * 4 + i] *= dm[i];
          } else {
            gy[c] *= drop_mul(drop, (uint64_t)row * drop_row_stride + IO::idx(lane, c, 0));
          }
        }
      }
      IO::store(dxb + (size_t)row * D, lane, gy);
    }
  }
  if (gmax) {      // non-negative floats order like their bit patterns; the plain read keeps the atomics to the few waves that raise the maximum
    am = wave_max(am);
    if (lane == 0 && !(am <= *gmax)) atomicMax(reinterpret_cast<unsigned int*>(gmax), __float_as_uint(am));
  }
}

template <int NPL>
static int ln_fwd_launch(const void* x, long xs, const float* gamma, const float* beta, float eps, void* y, float* mean,
                         float* rstd, int M, int dtype, int xdtype, hipStream_t st) {
  const int grid = min((M + 3) / 4, 256 * 8);
#define GSL_LNF(T, X)                                                                                                              \
  hipLaunchKernelGGL((ln_fwd_kernel<NPL, T, X>), dim3(grid), dim3(256), 0, st, (const X*)x, xs, gamma, beta, eps, (T*)y, mean, rstd, M)
  if (dtype == GSL_F16 && xdtype == GSL_F16) GSL_LNF(f16_t, f16_t);      // fp16 operands (round 5): y is an fp16 MFMA operand
  else if (dtype == GSL_F16) GSL_LNF(f16_t, float);
  else if (dtype == GSL_BF16 && xdtype == GSL_F16) GSL_LNF(bf16_t, f16_t);
  else if (dtype == GSL_BF16 && xdtype == GSL_BF16) GSL_LNF(bf16_t, bf16_t);
  else if (dtype == GSL_BF16) GSL_LNF(bf16_t, float);
  else GSL_LNF(float, float);
#undef GSL_LNF
  return check_launch("gsl_layernorm_fwd");
}

template <int NPL>
static int ln_bwd_launch(const void* dy, const void* x, long xs, const float* gamma, const float* mean, const float* rstd,
                         const void* dres, void* dx, long ios, void* dxb, int M, int dtype, int sdtype, int xdtype, DropCfg drop, long drs,
                         int cls_T, float* gmax, hipStream_t st) {
  const int grid = min((M + 3) / 4, 256 * 8);
#define GSL_LNB(T, S, X)                                                                                                            \
  hipLaunchKernelGGL((ln_bwd_kernel<NPL, T, S, X>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const X*)x, xs, gamma, mean, rstd, \
                     (const S*)dres, (S*)dx, ios, (T*)dxb, M, drop, drs, cls_T, gmax)
  if (dtype == GSL_F16 && sdtype == GSL_F16 && xdtype == GSL_F16) GSL_LNB(f16_t, f16_t, f16_t);      // fp16 operands: loss-scaled gradients
  else if (dtype == GSL_F16 && sdtype == GSL_F16) GSL_LNB(f16_t, f16_t, float);
  else if (dtype == GSL_F16 && xdtype == GSL_F16) GSL_LNB(f16_t, float, f16_t);
  else if (dtype == GSL_F16) GSL_LNB(f16_t, float, float);
  else if (dtype == GSL_BF16 && sdtype == GSL_BF16 && xdtype == GSL_F16) GSL_LNB(bf16_t, bf16_t, f16_t);
  else if (dtype == GSL_BF16 && xdtype == GSL_F16) GSL_LNB(bf16_t, float, f16_t);
  else if (dtype == GSL_BF16 && sdtype == GSL_BF16 && xdtype == GSL_BF16) GSL_LNB(bf16_t, bf16_t, bf16_t);
  else if (dtype == GSL_BF16 && sdtype == GSL_BF16) GSL_LNB(bf16_t, bf16_t, float);
  else if (dtype == GSL_BF16 && xdtype == GSL_BF16) GSL_LNB(bf16_t, float, bf16_t);
  else if (dtype == GSL_BF16) GSL_LNB(bf16_t, float, float);
  else GSL_LNB(float, float, float);
#undef GSL_LNB
  return check_launch("gsl_layernorm_bwd");
}

// =====================================================================================
// LayerNorm forward + LoRA down-projection in one pass (bf16 mode): xn = LN(x) and u = alpha * xn P^T for the rank-r adapter that reads
// xn (FFN1's lora_A: the second K segment [xn | u] of the fused FFN1 GEMM, vit_face.py:330 + loralib Linear.forward). The separate
// skinny GEMM re-read xn (206 MB at M = 201 728) for 16 output columns. Here a wave owns 16 rows at a time and holds them in the MFMA
// operand layout (lane l: row l % 16, the 8-element chunks 32 s + 8 (l / 16) of every 32-wide k-step s: one 16-byte load per step), so
// the normalised row, rounded to bf16 for the store, IS the B operand of v_mfma_f32_16x16x32_bf16 against P's fragments (LDS):
// u^T[j][row] accumulates in k order like the GEMM it replaces. Row statistics: 128 (192) elements per lane + two xor-shuffles over the
// four lanes of a row; two-pass (mean, centred variance) as the plain kernel. gamma / beta / P live in LDS.
// u is written as [M, 64] bf16 with columns >= 16 zero (the K-segment width of the consumer).
// =====================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 ln_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float ln_f32x4_t;

template <int KS>
__global__ __launch_bounds__(256, (KS <= 16 ? 4 : 3)) void ln_fwd_lora_kernel(const bf16_t* __restrict__ x, long xs, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, bf16_t* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                          const bf16_t* __restrict__ P, int ldp, float alpha, bf16_t* __restrict__ u) {
  fp16_sat_on();
  constexpr int D = KS * 32;
  constexpr int PLD = D + 8;                  // 16 more bytes per row: the 16 rows of a fragment read start 4 banks apart
  __shared__ float sg[D], sb[D];
  __shared__ __attribute__((aligned(16))) bf16_t sP[16 * PLD];
  for (int i = threadIdx.x; i < D; i += blockDim.x) { sg[i] = gamma[i]; sb[i] = beta[i]; }
  for (int i = threadIdx.x; i < 16 * (D / 8); i += blockDim.x) {
    const int j = i / (D / 8), c = i - j * (D / 8);
    *reinterpret_cast<uint4*>(&sP[j * PLD + c * 8]) = *reinterpret_cast<const uint4*>(P + (size_t)j * ldp + c * 8);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, kc = lane >> 4;
  const int wpb = blockDim.x >> 6, ngrp = (M + 15) >> 4;
  for (int grp = blockIdx.x * wpb + (threadIdx.x >> 6); grp < ngrp; grp += gridDim.x * wpb) {
    const int row = grp * 16 + r;
    const bool valid = row < M;
    const bf16_t* px = x + (size_t)(valid ? row : M - 1) * xs + kc * 8;
    uint4 v[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) v[s] = *reinterpret_cast<const uint4*>(px + s * 32);
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint32_t w[4] = {v[s].x, v[s].y, v[s].z, v[s].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) sum += __uint_as_float(w[i] << 16) + __uint_as_float(w[i] & 0xffff0000u);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mu = sum * (1.0f / D);
    // the packed row stays the only copy: without the opaque touch the compiler keeps the 128 converted floats of one pass for the next (spills)
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(v[s].x), "+v"(v[s].y), "+v"(v[s].z), "+v"(v[s].w));
    float q = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint32_t w[4] = {v[s].x, v[s].y, v[s].z, v[s].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = __uint_as_float(w[i] << 16) - mu, b = __uint_as_float(w[i] & 0xffff0000u) - mu;
        q += a * a + b * b;
      }
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rs = rsqrtf(q * (1.0f / D) + eps);
    const float mu3 = mu;
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(v[s].x), "+v"(v[s].y), "+v"(v[s].z), "+v"(v[s].w));
    ln_f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    bf16_t* py = y + (size_t)row * D + kc * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k0 = s * 32 + kc * 8;
      const float4 g0 = *reinterpret_cast<const float4*>(&sg[k0]), g1 = *reinterpret_cast<const float4*>(&sg[k0 + 4]);
      const float4 b0 = *reinterpret_cast<const float4*>(&sb[k0]), b1 = *reinterpret_cast<const float4*>(&sb[k0 + 4]);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const uint32_t w[4] = {v[s].x, v[s].y, v[s].z, v[s].w};
      uint32_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = (__uint_as_float(w[i] << 16) - mu3) * rs * gg[2 * i] + bb[2 * i];
        const float b = (__uint_as_float(w[i] & 0xffff0000u) - mu3) * rs * gg[2 * i + 1] + bb[2 * i + 1];
        o[i] = pack2bf(a, b);
      }
      union { uint4 q4; ln_bf16x8_t h; } xf, pf;
      xf.q4 = make_uint4(o[0], o[1], o[2], o[3]);
      if (valid) *reinterpret_cast<uint4*>(py + s * 32) = xf.q4;
      pf.q4 = *reinterpret_cast<const uint4*>(&sP[r * PLD + k0]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf.h, xf.h, acc, 0, 0, 0);      // acc[i] = u[row][4 kc + i]
      __builtin_amdgcn_sched_barrier(0);      // keep the LDS reads of step s + 1 behind this step (hoisted, they cost 300 more VGPRs)
    }
    if (valid) {
      bf16_t* pu = u + (size_t)row * 64;
      *reinterpret_cast<uint2*>(pu + kc * 4) = make_uint2(pack2bf(acc[0] * alpha, acc[1] * alpha), pack2bf(acc[2] * alpha, acc[3] * alpha));
      *reinterpret_cast<uint4*>(pu + 16 + kc * 8) = make_uint4(0u, 0u, 0u, 0u);
      if (kc < 2) *reinterpret_cast<uint4*>(pu + 48 + kc * 8) = make_uint4(0u, 0u, 0u, 0u);
      if (kc == 0) { mean[row] = mu; rstd[row] = rs; }
    }
  }
}

#define GSL_DISPATCH_D(D, CALL)                                                             \
  switch (D) {                                                                              \
    case 64: return CALL(1);                                                                \
    case 128: return CALL(2);                                                               \
    case 256: return CALL(4);                                                               \
    case 512: return CALL(8);                                                               \
    case 768: return CALL(12);                                                              \
    case 1024: return CALL(16);                                                             \
    default: return fail(GSL_ERR_UNSUPPORTED, "%s: unsupported LayerNorm width %ld", __func__, (long)(D)); \
  }

extern "C" int gsl_layernorm_fwd(const void* x, long x_row_stride, const float* gamma, const float* beta, float eps,
                                 void* y, float* mean, float* rstd, int M, int D, int dtype, int x_dtype, gsl_stream_t s) {
  GSL_CHECK_ARG(x && gamma && beta && mean && rstd && M > 0, "null/size");      // y == nullptr: statistics only
  GSL_CHECK_ARG(dtype == GSL_F32 || dtype == GSL_BF16 || dtype == GSL_F16, "dtype");
  GSL_CHECK_ARG(x_dtype == GSL_F32 || ((x_dtype == GSL_BF16 || x_dtype == GSL_F16) && dtype == GSL_BF16) || (x_dtype == GSL_F16 && dtype == GSL_F16),
                "x dtype (a 16-bit stream only in a 16-bit mode; bf16 stream only with bf16 operands)");
  GSL_CHECK_ARG((x_row_stride % 4) == 0, "row stride alignment");
#define CALL(N) ln_fwd_launch<N>(x, x_row_stride, gamma, beta, eps, y, mean, rstd, M, dtype, x_dtype, as_stream(s))
  GSL_DISPATCH_D(D, CALL)
#undef CALL
}

extern "C" int gsl_layernorm_bwd(const void* dy, const void* x, long x_row_stride, const float* gamma, const float* mean,
                                 const float* rstd, const void* dres, void* dx, long io_row_stride, void* dxb, int M, int D,
                                 int dtype, int stream_dtype, int x_dtype, float p_drop, uint64_t seed, uint32_t site,
                                 long drop_row_stride, int dres_cls_T, float* gmax, gsl_stream_t s) {
  GSL_CHECK_ARG(dy && x && gamma && mean && rstd && dx && M > 0, "null/size");
  GSL_CHECK_ARG(dtype == GSL_F32 || dtype == GSL_BF16 || dtype == GSL_F16, "dtype");
  GSL_CHECK_ARG(stream_dtype == GSL_F32 || (stream_dtype == dtype && dtype != GSL_F32), "stream dtype (f32, or the operand format of a 16-bit mode)");
  GSL_CHECK_ARG(x_dtype == GSL_F32 || ((x_dtype == GSL_BF16 || x_dtype == GSL_F16) && dtype == GSL_BF16) || (x_dtype == GSL_F16 && dtype == GSL_F16),
                "x dtype (a 16-bit stream only in a 16-bit mode; bf16 stream only with bf16 operands)");
  GSL_CHECK_ARG((x_row_stride % 4) == 0, "row stride alignment");
  GSL_CHECK_ARG(dres_cls_T >= 0 && (dres_cls_T == 0 || (dres && dres != dx)), "dres_cls_T: compact dres, out of place");
  const DropCfg drop = make_drop(p_drop, seed, site);
  const long ios = io_row_stride > 0 ? io_row_stride : D, drs = drop_row_stride > 0 ? drop_row_stride : D;
  GSL_CHECK_ARG((ios % 4) == 0, "io row stride alignment");
#define CALL(N) ln_bwd_launch<N>(dy, x, x_row_stride, gamma, mean, rstd, dres, dx, ios, dxb, M, dtype, stream_dtype, x_dtype, drop, drs, dres_cls_T, gmax, as_stream(s))
  GSL_DISPATCH_D(D, CALL)
#undef CALL
}

// LayerNorm forward that also emits u = alpha * LN(x) P^T (bf16 mode; P [>= 16, D] bf16 rows = lora_A rows zero-padded to 16; u [M, 64] bf16,
// columns >= 16 written as zero): see ln_fwd_lora_kernel. D in {512, 768}.
extern "C" int gsl_layernorm_fwd_lora(const void* x, long x_row_stride, const float* gamma, const float* beta, float eps, void* y,
                                      float* mean, float* rstd, int M, int D, const void* P, int ldp, float alpha, void* u,
                                      gsl_stream_t s) {
  GSL_CHECK_ARG(x && gamma && beta && y && mean && rstd && P && u && M > 0, "null/size");
  GSL_CHECK_ARG(D == 512 || D == 768, "width (512 or 768)");
  GSL_CHECK_ARG((x_row_stride % 8) == 0 && (ldp % 8) == 0 && ldp >= D, "row stride alignment");
  const int grid = min((M + 63) / 64, 256 * 4);
  if (D == 512)
    hipLaunchKernelGGL((ln_fwd_lora_kernel<16>), dim3(grid), dim3(256), 0, as_stream(s), (const bf16_t*)x, x_row_stride, gamma, beta, eps,
                       (bf16_t*)y, mean, rstd, M, (const bf16_t*)P, ldp, alpha, (bf16_t*)u);
  else
    hipLaunchKernelGGL((ln_fwd_lora_kernel<24>), dim3(grid), dim3(256), 0, as_stream(s), (const bf16_t*)x, x_row_stride, gamma, beta, eps,
                       (bf16_t*)y, mean, rstd, M, (const bf16_t*)P, ldp, alpha, (bf16_t*)u);
  return check_launch("gsl_layernorm_fwd_lora");
}
