// gsl_common.h — shared device helpers for libgslora_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gslora_hip.h"

namespace gsl {

// ------------------------------------------------------------------ errors
extern thread_local char g_err[512];
inline int fail(int code, const char* fmt, const char* a = "", long b = 0, long c = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c);
  return code;
}
#define GSL_CHECK_ARG(cond, msg)                                                        \
  do {                                                                                  \
    if (!(cond)) return gsl::fail(GSL_ERR_ARG, "%s: argument check failed: " msg " (%ld,%ld)", __func__, 0, 0); \
  } while (0)
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return GSL_ERR_LAUNCH;
  }
  return GSL_OK;
}
#define GSL_LAUNCH_CHECK() return gsl::check_launch(__func__)

// ------------------------------------------------------------------ bf16
typedef uint16_t bf16_t;
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16 round-to-nearest-even in hardware: v_cvt_pk_bf16_f32 (gfx950). A hand-written integer RNE with a
// NaN test costs ~7 VALU ops and a divergent exec-mask branch PER ELEMENT (88 s_and_saveexec in the attention loop).
typedef __attribute__((ext_vector_type(2))) float gsl_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 gsl_bf16x2;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const gsl_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, gsl_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// Everything below up to the end of the operand-format section lives in an INLINE namespace named after the translation unit's 16-bit operand
// format: gemm.hip / attention.hip are compiled twice (bf16 and, with -DGSL_OP_F16, fp16), and Elem<uint16_t>, pack2o, unpack2o, o2f, f2o then
// have two different definitions under one name — distinct mangled names keep that legal under -fgpu-rdc / LTO or if one stops being inlined
// (ADVICE r05). Source code is unaffected (inline namespace members are found in gsl::).
#ifdef GSL_OP_F16
inline namespace op_f16 {
#else
inline namespace op_bf16 {
#endif
template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void ld4(const float* p, float v[4]) {
    float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void st4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
// ---- fp16 (IEEE binary16): element type of the FORWARD residual stream in speed mode (round 4). The stream is only ever read by
// LayerNorm kernels and residual-add epilogues — never a matrix-core operand — so it does not need bf16's exponent range, and fp16's 11-bit
// significand rounds each of the 13 stream tensors of a forward 8x finer than bf16's 8 bits at the same 2 bytes per element
// (profiles/r04_b_precision_ablation.md: the bf16 stream was 3/4 of the speed mode's logit error). Values are clamped to +-65504 on store.
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(2))) _Float16 gsl_f16x2;
// f32 -> fp16 SATURATES in hardware: every kernel that stores fp16 starts with fp16_sat_on(), which sets MODE.FP16_OVFL (bit 23) for the
// wave — a finite result beyond +-65504 becomes +-65504 instead of Inf, while Inf and NaN inputs stay Inf / NaN (a diverged step must
// stay visible through every 16-bit tensor, as in the bf16 / f32 forms and in the reference). Probed on MI355X (tools/probes/fp16_ovfl.hip):
// 1e6 -> 0x7bff, 65520 -> 0x7bff, Inf -> 0x7c00, NaN -> 0x7e00; without the bit 65520 and 1e6 -> Inf. The convert is then one
// v_cvt_pk_f16_f32 per pair — the cost of the bf16 pack; a software clamp (v_med3 + a NaN select, round 4 / ADVICE r04) cost the fused
// FFN1 epilogue +9 % and the step +6 % (profiles/r05_notes.md). clamp_h() is the software form for code that cannot rely on the mode bit.
__device__ __forceinline__ void fp16_sat_on() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
__device__ __forceinline__ float clamp_h(float v) {
  const float c = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
  return (v != v || __builtin_isinf(v)) ? v : c;
}
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {      // requires fp16_sat_on() at kernel entry
  const gsl_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, gsl_f16x2));      // round to nearest even
}
__device__ __forceinline__ void unpack2h(uint32_t u, float& lo, float& hi) {
  const gsl_f32x2 v = __builtin_convertvector(__builtin_bit_cast(gsl_f16x2, u), gsl_f32x2);
  lo = v[0]; hi = v[1];
}
// two stream elements <-> one dword, the element type chosen at run time (wave-uniform): 0 = bf16, 1 = fp16
__device__ __forceinline__ uint32_t pack2s(float lo, float hi, int f16) { return f16 ? pack2h(lo, hi) : pack2bf(lo, hi); }
__device__ __forceinline__ void unpack2s(uint32_t u, int f16, float& lo, float& hi) {
  if (f16) unpack2h(u, lo, hi);
  else { lo = __uint_as_float(u << 16); hi = __uint_as_float(u & 0xffff0000u); }
}
template <> struct Elem<f16_t> {
  static __device__ __forceinline__ float ld(const f16_t* p) { return (float)*p; }
  static __device__ __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }      // (fp16_sat_on() at kernel entry)
  static __device__ __forceinline__ void ld4(const f16_t* p, float v[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    unpack2h(t.x, v[0], v[1]); unpack2h(t.y, v[2], v[3]);
  }
  static __device__ __forceinline__ void st4(f16_t* p, const float v[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2h(v[0], v[1]), pack2h(v[2], v[3]));
  }
};

// ------------------------------------------------------------------ the 16-bit matrix-core operand format of a translation unit
// The speed mode exists in two operand formats with the SAME kernels, bytes and MFMA rate: bf16 (dtype GSL_BF16: 8-bit significand,
// f32's exponent range) and IEEE fp16 (dtype GSL_F16, round 5: 11-bit significand; activations clamp at +-65504 on store, the
// backward runs on gradients multiplied by a power-of-two loss scale that the LoRA-gradient reductions divide out again).
// gemm.hip and attention.hip are compiled TWICE (gslora_hip/build.py): once plain (bf16 + the f32 parity kernels, the exported entry
// points) and once with -DGSL_OP_F16 — then every kernel of the file lives in namespace gsl_h16, the operand helpers below mean fp16,
// only the 16-bit paths are compiled and the entry points are the hidden symbols h16_<name> that the exported ones forward to when
// dtype == GSL_F16. Storage type of an operand element is uint16_t in both formats (op16_t): every conversion goes through these
// helpers, nothing converts numerically by accident.
typedef uint16_t op16_t;
// first statement of every kernel of gemm.hip / attention.hip (both compiles: the bf16 build stores fp16 too — the forward residual stream)
#define GSL_OP16_KERNEL_ENTRY() gsl::fp16_sat_on()
#ifdef GSL_OP_F16
#define GSL_OP16 GSL_F16
#define GSL_OPNS_BEGIN namespace gsl_h16 {
#define GSL_OPNS_END }
#define GSL_ENTRY(name) h16_##name
#define GSL_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define GSL_MFMA32 __builtin_amdgcn_mfma_f32_32x32x16_f16
typedef _Float16 gsl_op16_elem;
__device__ __forceinline__ uint32_t pack2o(float lo, float hi) { return pack2h(lo, hi); }
__device__ __forceinline__ void unpack2o(uint32_t u, float& lo, float& hi) { unpack2h(u, lo, hi); }
__device__ __forceinline__ float o2f(op16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ op16_t f2o(float f) { return __builtin_bit_cast(op16_t, (_Float16)f); }      // (fp16_sat_on() at kernel entry)
#else
#define GSL_OP16 GSL_BF16
#define GSL_OPNS_BEGIN
#define GSL_OPNS_END
#define GSL_ENTRY(name) name
#define GSL_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define GSL_MFMA32 __builtin_amdgcn_mfma_f32_32x32x16_bf16
typedef __bf16 gsl_op16_elem;
__device__ __forceinline__ uint32_t pack2o(float lo, float hi) { return pack2bf(lo, hi); }
__device__ __forceinline__ void unpack2o(uint32_t u, float& lo, float& hi) { lo = __uint_as_float(u << 16); hi = __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float o2f(op16_t v) { return bf2f(v); }
__device__ __forceinline__ op16_t f2o(float f) { return f2bf(f); }
#endif
typedef __attribute__((ext_vector_type(8))) gsl_op16_elem op16x8_t;      // one MFMA A / B operand: 8 elements per lane
// element access of an operand tensor (uint16_t storage) in this translation unit's format
template <> struct Elem<op16_t> {
  static __device__ __forceinline__ float ld(const op16_t* p) { return o2f(*p); }
  static __device__ __forceinline__ void st(op16_t* p, float v) { *p = f2o(v); }
  static __device__ __forceinline__ void ld4(const op16_t* p, float v[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    unpack2o(t.x, v[0], v[1]); unpack2o(t.y, v[2], v[3]);
  }
  static __device__ __forceinline__ void st4(op16_t* p, const float v[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2o(v[0], v[1]), pack2o(v[2], v[3]));
  }
};
}  // inline namespace op_f16 / op_bf16

// ------------------------------------------------------------------ wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block reduce (sum) for blockDim.x <= 1024, result valid in all threads; `sm` >= 16 floats
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sm[i];   // fixed order -> deterministic
  return t;
}
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float t = -3.0e38f;
  for (int i = 0; i < nw; ++i) t = fmaxf(t, sm[i]);
  return t;
}

// ------------------------------------------------------------------ dropout (counter-based hash)
// keep(i) for element index i of dropout site `site` under `seed`: a 2-round multiply-xorshift
// hash of the 64-bit pair counter idx/2 gives two 16-bit samples (elements 2k, 2k+1), each compared
// with round(p*2^16). The same function regenerates the mask in backward — no mask tensor is stored.
struct DropCfg {
  uint32_t thr;   // drop if 16-bit sample < thr ; thr = 0 disables (thr = round(p * 65536))
  uint32_t key;   // seed/site mix (host-computed when the seed is passed by value)
  float scale;    // 1/(1-p)
  uint32_t site;
  const uint64_t* seed_ptr;   // != nullptr: the seed lives in device memory (HIP-graph replays advance it), key is derived in-kernel
};
#define GSL_SEED_ON_DEVICE 0x80000000u   // flag bit of every `site` argument of the C ABI: `seed` is then a device pointer to a uint64
__host__ __device__ inline uint32_t drop_mix(uint64_t seed, uint32_t site) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + (uint64_t)site * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
  z ^= z >> 29; z *= 0xD6E8FEB86659FD93ull; z ^= z >> 32;
  return (uint32_t)z;
}
inline DropCfg make_drop(float p, uint64_t seed, uint32_t site) {
  DropCfg d;
  d.thr = (p > 0.f) ? (uint32_t)(p * 65536.0f + 0.5f) : 0u;
  d.site = site & ~GSL_SEED_ON_DEVICE;
  d.seed_ptr = (site & GSL_SEED_ON_DEVICE) ? reinterpret_cast<const uint64_t*>(seed) : nullptr;
  d.key = d.seed_ptr ? 0u : drop_mix(seed, d.site);
  d.scale = (p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
  return d;
}
// first statement of every kernel that applies dropout: one uniform 8-byte load when the seed is device resident
__device__ __forceinline__ void resolve_drop(DropCfg& d) {
  if (d.seed_ptr && d.thr) { d.key = drop_mix(*d.seed_ptr, d.site); }
  d.seed_ptr = nullptr;
}
// one 32-bit hash serves the element pair (2k, 2k+1): low / high 16 bits. First stage: a Weyl sequence in the pair counter,
// w(k) = (k + key) * phi mod 2^32 (key = seed/site mix), second stage: xor-shift + multiply. The first stage is LINEAR in k, so a
// kernel that walks a regular lattice of indices (the GEMM epilogues: rows 16 apart, columns 16 apart) advances w with one add per
// hash instead of a 64-bit index + a quarter-rate 32-bit multiply (drop_w0 / drop_finish below) — the epilogue that uses it is
// VALU-bound (profiles/r01_gemm_ab.md), every instruction per element counts. Only the low 32 bits of the pair counter enter (the
// stream repeats every 2^33 elements; the tensors here have < 2^29). Checked on 16 M consecutive indices, four keys, p = 0.1 / 0.5:
// rate, lag-1..8192 autocorrelation (< 5e-3), drop-gap variance (90.1 vs 90.0 geometric), per-1024 counts and cross-key joint rate
// match an ideal Bernoulli stream; dropping the second multiply does NOT (gap variance 50: k*phi alone is a low-discrepancy sequence).
constexpr uint32_t DROP_PHI = 0x9E3779B1u;
__device__ __forceinline__ uint32_t drop_finish(uint32_t w) {     // second stage on a first-stage value
  w ^= w >> 15; w *= 0x85EBCA77u;
  return w;
}
__device__ __forceinline__ uint32_t drop_w0(uint32_t key, uint64_t pair) { return ((uint32_t)pair + key) * DROP_PHI; }
__device__ __forceinline__ uint32_t drop_hash(uint32_t key, uint64_t pair) { return drop_finish(drop_w0(key, pair)); }

// multiplier to apply to an element: 0 or 1/(1-p)
__device__ __forceinline__ float drop_mul(const DropCfg& d, uint64_t idx) {
  if (d.thr == 0u) return 1.0f;
  const uint32_t h = drop_hash(d.key, idx >> 1);
  const uint32_t s = (idx & 1) ? (h >> 16) : (h & 0xffffu);
  return (s < d.thr) ? 0.0f : d.scale;
}
// four consecutive elements starting at a multiple of 4: two hashes; w0 = drop_w0(key, idx >> 1)
__device__ __forceinline__ void drop_mul4_w(const DropCfg& d, uint32_t w0, float m[4]) {
  if (d.thr == 0u) { m[0] = m[1] = m[2] = m[3] = 1.0f; return; }
  const uint32_t h0 = drop_finish(w0), h1 = drop_finish(w0 + DROP_PHI);
  m[0] = ((h0 & 0xffffu) < d.thr) ? 0.0f : d.scale;
  m[1] = ((h0 >> 16) < d.thr) ? 0.0f : d.scale;
  m[2] = ((h1 & 0xffffu) < d.thr) ? 0.0f : d.scale;
  m[3] = ((h1 >> 16) < d.thr) ? 0.0f : d.scale;
}
// the same without the "no dropout" test: for callers that decided it once, outside their loop
__device__ __forceinline__ void drop_mul4_on(const DropCfg& d, uint32_t w0, float m[4]) {
  const uint32_t h0 = drop_finish(w0), h1 = drop_finish(w0 + DROP_PHI);
  m[0] = ((h0 & 0xffffu) < d.thr) ? 0.0f : d.scale;
  m[1] = ((h0 >> 16) < d.thr) ? 0.0f : d.scale;
  m[2] = ((h1 & 0xffffu) < d.thr) ? 0.0f : d.scale;
  m[3] = ((h1 >> 16) < d.thr) ? 0.0f : d.scale;
}
__device__ __forceinline__ void drop_mul4(const DropCfg& d, uint64_t idx, float m[4]) {
  drop_mul4_w(d, d.thr ? drop_w0(d.key, idx >> 1) : 0u, m);
}

// exact-erf GELU and its derivative (nn.GELU default, vit_face.py:331)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// speed-mode GELU + derivative: Abramowitz-Stegun 7.1.26 erf (|err| < 1.5e-7), one v_exp shared by the
// cdf and the pdf (exp(-z^2) with z = |a|/sqrt2 IS exp(-a^2/2)), one v_rcp. ~20 VALU ops instead of ~120.
__device__ __forceinline__ void gelu_pair_fast(float a, float& g, float& gp) {
  // constants folded so that every step is one VALU op (the epilogue that calls this is VALU-bound): 1 + p z = fma(|a|, p/sqrt2, 1);
  // the final 1/2 of the cdf sits in the polynomial coefficients: h = (t poly(t) / 2) e = 1 - Phi(|a|), Phi(a) = 1/2 + copysign(1/2 - h, a)
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(a), 0.3275911f * 0.70710678118654752f, 1.0f));   // v_rcp_f32 (1 ulp)
  const float e = __builtin_amdgcn_exp2f((a * a) * -0.72134752044448170f);   // exp(-a^2/2) = 2^(-a^2 * log2(e)/2): one v_exp_f32
  const float ph = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f), 0.5f * 1.421413741f), 0.5f * -0.284496736f),
                            0.5f * 0.254829592f);
  const float cdf = 0.5f + copysignf(fmaf(-ph, e, 0.5f), a);
  g = a * cdf;
  gp = fmaf(a * 0.39894228040143268f, e, cdf);
}

inline hipStream_t as_stream(gsl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace gsl
