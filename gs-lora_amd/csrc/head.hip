// head.hip — the two ends of the network and the scalar losses:
//   K1  patch gather (einops rearrange, vit_pytorch_face/vit_face.py:530)
//   K10 cls-pool + LayerNorm + CosFace margin head (vit_face.py:540-546, 171-208) fwd / bwd
//   K11 mean cross-entropy + top-1 (engine_cl.py:65-78, util/utils.py:354-368) fwd / bwd
//   K13 prototype KL (engine_cl.py:571-603) fwd / bwd
// All are tiny next to the GEMMs; they exist so that a step needs no host sync and no [B,C]-sized
// PyTorch elementwise chain. Upstream gradient scalars arrive as DEVICE pointers (coef).
#include "gsl_common.h"

using namespace gsl;

// ------------------------------------------------------------------ K1 patchify
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int C, int H, int W, int p) {
  fp16_sat_on();
  const int hp = H / p, wp = W / p, Tn = 1 + hp * wp, Kp = p * p * C;
  const long total = (long)B * Tn * p * p;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int p2 = (int)(idx % p);
    const int p1 = (int)((idx / p) % p);
    const int t = (int)((idx / (p * p)) % Tn);
    const int b = (int)(idx / ((long)p * p * Tn));
    T* o = out + ((size_t)b * Tn + t) * Kp + (size_t)(p1 * p + p2) * C;
    if (t == 0) {
      for (int c = 0; c < C; ++c) Elem<T>::st(o + c, 0.f);
    } else {
      const int h = (t - 1) / wp, w = (t - 1) % wp;
      const float* src = img + ((size_t)b * C * H + (size_t)(h * p + p1)) * W + (size_t)(w * p + p2);
      for (int c = 0; c < C; ++c) Elem<T>::st(o + c, src[(size_t)c * H * W]);
    }
  }
}

extern "C" int gsl_patchify(const float* img, void* out, int B, int C, int H, int W, int p, int dtype, gsl_stream_t s) {
  GSL_CHECK_ARG(img && out && B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, "shape");
  const long total = (long)B * (1 + (H / p) * (W / p)) * p * p;
  const int grid = (int)min((total + 255) / 256, (long)(256 * 16));
  if (dtype == GSL_BF16) hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(s), img, (bf16_t*)out, B, C, H, W, p);
  else if (dtype == GSL_F16) hipLaunchKernelGGL(patchify_kernel<f16_t>, dim3(grid), dim3(256), 0, as_stream(s), img, (f16_t*)out, B, C, H, W, p);
  else if (dtype == GSL_F32) hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid), dim3(256), 0, as_stream(s), img, (float*)out, B, C, H, W, p);
  else return fail(GSL_ERR_ARG, "gsl_patchify: bad dtype%s %ld", "", dtype);
  return check_launch("gsl_patchify");
}

// ------------------------------------------------------------------ K10 head
__global__ void cosface_prep_kernel(const float* __restrict__ W, float* __restrict__ Wn, int C, int D) {
  fp16_sat_on();
  // one wave per class row: Wn = W / max(||W||, 1e-12)   (F.normalize, vit_face.py:181)
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = W[(size_t)c * D + d]; ss += v * v; }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
  for (int d = lane; d < D; d += 64) Wn[(size_t)c * D + d] = W[(size_t)c * D + d] * inv;
}
extern "C" int gsl_cosface_prep(const float* W, float* Wn, int C, int D, gsl_stream_t s) {
  GSL_CHECK_ARG(W && Wn && C > 0 && D > 0, "null/size");
  hipLaunchKernelGGL(cosface_prep_kernel, dim3((C + 3) / 4), dim3(256), 0, as_stream(s), W, Wn, C, D);
  return check_launch("gsl_cosface_prep");
}

constexpr int HEAD_MAXD = 1024;

// X: element type of the residual stream x (f32, or bf16 in speed mode)
template <typename X>
__global__ __launch_bounds__(1024) void head_fwd_kernel(const X* __restrict__ x, int T, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, const float* __restrict__ Wn,
                                                       const int64_t* __restrict__ label, float* __restrict__ emb,
                                                       float* __restrict__ mean, float* __restrict__ rstd,
                                                       float* __restrict__ logits, int D, int C, float cs, float cm,
                                                       const float* __restrict__ hbias, int linear, int pool_mean) {
  fp16_sat_on();
  __shared__ float e[HEAD_MAXD];
  __shared__ float sm[16];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, nwv = blockDim.x >> 6;      // 256 threads, or 1024 when few images share the chip
  const X* xr = x + (size_t)b * T * D;
  float s = 0.f;
  for (int d = tid; d < D; d += nt) {
    float pv = Elem<X>::ld(xr + d);                 // pool = 'cls': token 0 (vit_face.py:540)
    if (pool_mean) {                                // pool = 'mean': x.mean(dim=1) over all T tokens, summed in token order
      for (int t = 1; t < T; ++t) pv += Elem<X>::ld(xr + (size_t)t * D + d);
      pv = pv / (float)T;
    }
    e[d] = pv; s += pv;
  }
  const float mu = block_sum(s, sm) / D;
  float q = 0.f;
  for (int d = tid; d < D; d += nt) { const float c = e[d] - mu; q += c * c; }
  const float rs = rsqrtf(block_sum(q, sm) / D + eps);
  float nn = 0.f;
  for (int d = tid; d < D; d += nt) {
    const float v = (e[d] - mu) * rs * gamma[d] + beta[d];
    e[d] = v;
    emb[(size_t)b * D + d] = v;
    nn += v * v;
  }
  if (tid == 0) { mean[b] = mu; rstd[b] = rs; }
  const float inv = 1.0f / fmaxf(sqrtf(block_sum(nn, sm)), 1e-12f);   // block_sum syncs -> e[] complete
  if (!logits) return;
  const int lane = tid & 63, wave = tid >> 6;
  const long lab = label ? (long)label[b] : -1;
  // a wave owns classes wave, wave + 4, ...; five of them per round so that the loads of five W rows are in flight together (one
  // class per round made the kernel a chain of 25 dependent global-load latencies: 60 us for 100 classes at any batch size)
  constexpr int CB = 5;
  for (int c0 = wave; c0 < C; c0 += nwv * CB) {
    float dot[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) dot[k] = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float ev = e[d];
#pragma unroll
      for (int k = 0; k < CB; ++k) {
        const int c = min(c0 + nwv * k, C - 1);
        dot[k] += ev * Wn[(size_t)c * D + d];
      }
    }
#pragma unroll
    for (int k = 0; k < CB; ++k) {
      const int c = c0 + nwv * k;
      float dk = wave_sum(dot[k]);
      if (lane == 0 && c < C) {
        if (linear) logits[(size_t)b * C + c] = dk + (hbias ? hbias[c] : 0.f);     // plain nn.Linear head (modified_VIT.py:34-36)
        else { dk *= inv; logits[(size_t)b * C + c] = cs * ((c == lab) ? (dk - cm) : dk); }
      }
    }
  }
}

extern "C" int gsl_head_fwd(const void* x, int x_dtype, int T, const float* gamma, const float* beta, float eps, const float* Wn,
                            const int64_t* label, float* emb, float* mean, float* rstd, float* logits, int B, int D, int C,
                            float cos_s, float cos_m, const float* head_bias, int linear_head, int pool_mean, gsl_stream_t s) {
  GSL_CHECK_ARG(x_dtype == GSL_F32 || x_dtype == GSL_BF16 || x_dtype == GSL_F16, "x dtype");
  GSL_CHECK_ARG(x && gamma && beta && emb && mean && rstd && B > 0 && T > 0, "null/size");
  GSL_CHECK_ARG(D > 0 && D <= HEAD_MAXD && (D % 4) == 0, "D <= 1024, D%4==0");
  GSL_CHECK_ARG(!logits || (Wn && C > 0), "Wn required for logits");
  const int nthr = B <= 128 ? 1024 : 256;      // one workgroup per image: with few images give each one 16 waves (100 class rows in two rounds)
  if (x_dtype == GSL_F16)
    hipLaunchKernelGGL(head_fwd_kernel<f16_t>, dim3(B), dim3(nthr), 0, as_stream(s), (const f16_t*)x, T, gamma, beta, eps, Wn, label, emb,
                       mean, rstd, logits, D, C, cos_s, cos_m, head_bias, linear_head, pool_mean);
  else if (x_dtype == GSL_BF16)
    hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3(B), dim3(nthr), 0, as_stream(s), (const bf16_t*)x, T, gamma, beta, eps, Wn, label, emb,
                       mean, rstd, logits, D, C, cos_s, cos_m, head_bias, linear_head, pool_mean);
  else
    hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(B), dim3(nthr), 0, as_stream(s), (const float*)x, T, gamma, beta, eps, Wn, label, emb,
                       mean, rstd, logits, D, C, cos_s, cos_m, head_bias, linear_head, pool_mean);
  return check_launch("gsl_head_fwd");
}

// compact != 0 (pool = 'cls' only): dx / dxb are [B, D] — the gradient of the cls rows alone; the stream gradient of every other token is
// exactly zero and is neither written here nor read by the consumers (the cls-row-only backward of the last block, gsl_layernorm_bwd's
// dres_cls_T). The dropout counter of element (b, d) stays that of the dense tensor, (b*Tn)*D + d: same masks in both forms.
template <typename T, typename S, typename X>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ demb_in,
                                                       const X* __restrict__ x, int Tn, const float* __restrict__ gamma,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ emb, const float* __restrict__ Wn,
                                                       S* __restrict__ dx, T* __restrict__ dxb, int D, int C, float cs,
                                                       DropCfg drop, int linear, int pool_mean, int compact,
                                                       float* __restrict__ amax_out, const float* __restrict__ amax_in, int n_amax,
                                                       float* __restrict__ gscale_out, int target_exp) {
  fp16_sat_on();
  resolve_drop(drop);
  // fp16 operands (round 5): the backward runs on gradients multiplied by a power of two S chosen from the largest stream gradient
  // this kernel produces, S * max|g| in [2^(target_exp-1), 2^target_exp). Pass 1 (amax_out) writes max|g| of every image and stores
  // nothing else; pass 2 (amax_in) reduces them — every block the same way — scales its stores and block 0 publishes {S, 1/S}
  // (gscale_out) for the LoRA-gradient reductions, which divide S out again. Power of two: exact in every format.
  float gs = 1.0f;
  if (amax_in) {
    float am = 0.f;
    for (int i = threadIdx.x; i < n_amax; i += 256) am = fmaxf(am, amax_in[i]);
    __shared__ float sma[16];
    am = block_max(am, sma);
    // the exponent in use: gscale[3], settled by pass 1 (overflow guard, see below); a caller without the guard's state passes target_exp alone
    const int tex = gscale_out ? (int)gscale_out[3] : target_exp;
    if (am > 0.f && am < 3.0e38f) {
      int ex;
      (void)frexpf(am, &ex);                 // am = m 2^ex, m in [0.5, 1)
      gs = ldexpf(1.0f, min(max(tex - ex, -60), 60));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && gscale_out) { gscale_out[0] = gs; gscale_out[1] = 1.0f / gs; }
  }
  // Overflow guard (round 6): pass 1's block 0 settles the exponent E of THIS backward from what the PREVIOUS one saw — gscale[2] = the largest
  // |scaled gradient| its LayerNorm backwards read or stored: saturated (>= 65504) or non-finite -> E - 2 (floor 4); below 2^9 and E under the
  // target -> E + 1; E outside [4, 15] (a freshly zeroed buffer) -> the target — and clears gscale[2] for this backward. Pass 2 (a later launch)
  // reads E from gscale[3]. No host sync; replays of a captured graph carry the state in the buffer.
  if (amax_out && gscale_out && blockIdx.x == 0 && threadIdx.x == 0) {
    const float seen = gscale_out[2];
    int E = (int)gscale_out[3];
    if (!(gscale_out[3] >= 4.0f && gscale_out[3] <= 15.0f)) E = target_exp;
    else if (!(seen < 65504.0f)) E = max(E - 2, 4);
    else if (seen < 512.0f && seen > 0.f && E < target_exp) E = E + 1;
    gscale_out[3] = (float)E;
    gscale_out[2] = 0.0f;
  }
  __shared__ float de[HEAD_MAXD];   // d emb
  __shared__ float dl[1024];        // s * dlogits row (C <= 1024)
  __shared__ float xp[HEAD_MAXD];   // pooled row (pool = 'mean')
  __shared__ float sm[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  // pool = 'cls': zero the non-cls token rows of this image (their stream gradient is exactly 0)
  if (!pool_mean && !compact && !amax_out) {
    const long n4 = (long)(Tn - 1) * D / 4;
    {
      S* z = dx + ((size_t)b * Tn + 1) * D;
      const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
      for (long i = tid; i < n4; i += 256) Elem<S>::st4(z + i * 4, zero4);
    }
    if (dxb) {
      T* zb = dxb + ((size_t)b * Tn + 1) * D;
      const float zero[4] = {0.f, 0.f, 0.f, 0.f};
      for (long i = tid; i < n4; i += 256) Elem<T>::st4(zb + i * 4, zero);
    }
  }
  const float* er = emb + (size_t)b * D;
  float nn = 0.f;
  for (int d = tid; d < D; d += 256) nn += er[d] * er[d];
  const float nrm = fmaxf(sqrtf(block_sum(nn, sm)), 1e-12f);
  if (dlogits) {
    for (int c = tid; c < C; c += 256) dl[c] = cs * dlogits[(size_t)b * C + c];
  }
  __syncthreads();
  float dotp = 0.f;
  for (int d = tid; d < D; d += 256) {
    float g = 0.f;
    if (dlogits) {
      for (int c = 0; c < C; ++c) g += dl[c] * Wn[(size_t)c * D + d];
    }
    de[d] = g;                       // d e-hat
    dotp += g * er[d];
  }
  const float proj = block_sum(dotp, sm) / (nrm * nrm);   // (e-hat . d e-hat) / ||e||
  const float mu = mean[b], rs = rstd[b];
  const X* xr = x + (size_t)b * Tn * D;
  for (int d = tid; d < D; d += 256) {          // the pooled row the forward normalised (same summation order)
    float pv = Elem<X>::ld(xr + d);
    if (pool_mean) {
      for (int t = 1; t < Tn; ++t) pv += Elem<X>::ld(xr + (size_t)t * D + d);
      pv = pv / (float)Tn;
    }
    xp[d] = pv;
  }
  float s1 = 0.f, s2 = 0.f;
  for (int d = tid; d < D; d += 256) {
    float g = linear ? de[d] : (de[d] - er[d] * proj) / nrm;   // d emb from the Linear / CosFace head
    if (demb_in) g += demb_in[(size_t)b * D + d];
    g *= gamma[d];
    de[d] = g;
    const float xh = (xp[d] - mu) * rs;
    s1 += g;
    s2 += g * xh;
  }
  const float c1 = block_sum(s1, sm) / D;
  const float c2 = block_sum(s2, sm) / D;
  if (amax_out) {      // pass 1 of the loss-scaled form: this image's largest |stream gradient|, nothing else
    float am = 0.f;
    for (int d = tid; d < D; d += 256) {
      const float xh = (xp[d] - mu) * rs;
      am = fmaxf(am, fabsf(rs * (de[d] - c1 - xh * c2)));
    }
    am = block_max(am, sm);
    if (tid == 0) amax_out[b] = pool_mean ? am / (float)Tn : am;
    return;
  }
  for (int d = tid; d < D; d += 256) {
    const float xh = (xp[d] - mu) * rs;
    const float g = gs * (rs * (de[d] - c1 - xh * c2));
    if (!pool_mean) {
      const size_t o = (size_t)b * Tn * D + d, oc = compact ? (size_t)b * D + d : o;
      Elem<S>::st(dx + oc, g);
      if (dxb) Elem<T>::st(dxb + oc, g * drop_mul(drop, (uint64_t)o));
    } else {                                     // every token receives d pooled / T
      const float gt = g / (float)Tn;
      for (int t = 0; t < Tn; ++t) {
        const size_t o = ((size_t)b * Tn + t) * D + d;
        Elem<S>::st(dx + o, gt);
        if (dxb) Elem<T>::st(dxb + o, gt * drop_mul(drop, (uint64_t)o));
      }
    }
  }
}

// Loss scale of the fp16 backward: S * max|stream gradient at the head| lands in [2^10, 2^11). Measured on the full ViT-P8S8 (CPU emulation,
// tools/emu_operand_precision.py): the largest gradient operand anywhere in the backward is 1.2x the head's, so the chain peaks near 2.5e3
// (26x below fp16's 65504; stores saturate, they never produce Inf), and the LoRA-gradient error is flat for S between 2^6 and 2^20.
constexpr int GSL_GRAD_TARGET_EXP = 11;
extern "C" int gsl_head_bwd(const float* dlogits, const float* demb, const void* x, int x_dtype, int T, const float* gamma,
                            const float* mean, const float* rstd, const float* emb, const float* Wn, void* dx, void* dxb,
                            int B, int D, int C, float cos_s, int dtype, int stream_dtype, float p_drop, uint64_t seed, uint32_t site,
                            int linear_head, int pool_mean, int compact, float* gscale, float* amax_ws, int target_exp, gsl_stream_t s) {
  GSL_CHECK_ARG(x && gamma && mean && rstd && emb && dx && B > 0 && T >= 1, "null/size");
  GSL_CHECK_ARG(D > 0 && D <= HEAD_MAXD && (D % 4) == 0 && C <= 1024, "D <= 1024, D%4==0, C <= 1024");
  GSL_CHECK_ARG(!dlogits || Wn, "Wn required with dlogits");
  GSL_CHECK_ARG(!(compact && pool_mean), "compact cls-row gradients need pool = 'cls'");
  GSL_CHECK_ARG(!gscale || amax_ws, "gscale (loss-scaled gradients) needs amax_ws [B]");
  GSL_CHECK_ARG(target_exp == 0 || (target_exp >= 4 && target_exp <= 15), "target_exp: 0 (default 11) or 4 .. 15");
  const int texp = target_exp ? target_exp : GSL_GRAD_TARGET_EXP;
  const DropCfg drop = make_drop(p_drop, seed, site);
  GSL_CHECK_ARG(stream_dtype == GSL_F32 || (stream_dtype == dtype && dtype != GSL_F32), "stream dtype (f32, or the operand format of a 16-bit mode)");
  GSL_CHECK_ARG(x_dtype == GSL_F32 || ((x_dtype == GSL_BF16 || x_dtype == GSL_F16) && dtype == GSL_BF16) || (x_dtype == GSL_F16 && dtype == GSL_F16),
                "x dtype (a 16-bit stream only in a 16-bit mode; bf16 stream only with bf16 operands)");
#define GSL_HB(T_, S_, X_)                                                                                                          \
  do {                                                                                                                              \
    if (gscale)                                                                                                                     \
      hipLaunchKernelGGL((head_bwd_kernel<T_, S_, X_>), dim3(B), dim3(256), 0, as_stream(s), dlogits, demb, (const X_*)x, T, gamma, mean, \
                         rstd, emb, Wn, (S_*)dx, (T_*)dxb, D, C, cos_s, drop, linear_head, pool_mean, compact, amax_ws, (const float*)nullptr, 0, \
                         gscale, texp);                                                                                             \
    hipLaunchKernelGGL((head_bwd_kernel<T_, S_, X_>), dim3(B), dim3(256), 0, as_stream(s), dlogits, demb, (const X_*)x, T, gamma, mean, \
                       rstd, emb, Wn, (S_*)dx, (T_*)dxb, D, C, cos_s, drop, linear_head, pool_mean, compact, (float*)nullptr,       \
                       (const float*)(gscale ? amax_ws : nullptr), B, gscale, texp);                                 \
  } while (0)
  if (dtype == GSL_F16 && stream_dtype == GSL_F16 && x_dtype == GSL_F16) GSL_HB(f16_t, f16_t, f16_t);
  else if (dtype == GSL_F16 && stream_dtype == GSL_F16) GSL_HB(f16_t, f16_t, float);
  else if (dtype == GSL_F16 && x_dtype == GSL_F16) GSL_HB(f16_t, float, f16_t);
  else if (dtype == GSL_F16) GSL_HB(f16_t, float, float);
  else if (dtype == GSL_BF16 && stream_dtype == GSL_BF16 && x_dtype == GSL_F16) GSL_HB(bf16_t, bf16_t, f16_t);
  else if (dtype == GSL_BF16 && x_dtype == GSL_F16) GSL_HB(bf16_t, float, f16_t);
  else if (dtype == GSL_BF16 && stream_dtype == GSL_BF16 && x_dtype == GSL_BF16) GSL_HB(bf16_t, bf16_t, bf16_t);
  else if (dtype == GSL_BF16 && stream_dtype == GSL_BF16) GSL_HB(bf16_t, bf16_t, float);
  else if (dtype == GSL_BF16 && x_dtype == GSL_BF16) GSL_HB(bf16_t, float, bf16_t);
  else if (dtype == GSL_BF16) GSL_HB(bf16_t, float, float);
  else if (dtype == GSL_F32) GSL_HB(float, float, float);
  else return fail(GSL_ERR_ARG, "gsl_head_bwd: bad dtype%s %ld", "", dtype);
#undef GSL_HB
  return check_launch("gsl_head_bwd");
}

// ------------------------------------------------------------------ K11 cross entropy
// wave-per-row log-softmax; one block so the batch sum is a fixed-order (deterministic) reduction.
__device__ __forceinline__ void row_softmax_stats(const float* row, int C, int lane, float& mx, float& lse, int& amax) {
  float m = -3.0e38f; int mi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) { const float v = row[c]; if (v > m) { m = v; mi = c; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64); const int oi = __shfl_xor(mi, o, 64);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(row[c] - m);
  se = wave_sum(se);
  mx = m; lse = m + logf(se); amax = mi;
}

__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                      float* __restrict__ rows, int B, int C) {
  fp16_sat_on();
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= B) return;
  float mx, lse; int am;
  row_softmax_stats(logits + (size_t)r * C, C, lane, mx, lse, am);
  // a label outside [0, C) (the reference's CrossEntropyLoss raises): NaN loss, no out-of-bounds read; the deferred meter read stops the run
  const long yl = (long)labels[r];
  const bool ok = yl >= 0 && yl < C;
  const int y = ok ? (int)yl : -1;
  if (lane == 0) { rows[2 * r] = ok ? lse - logits[(size_t)r * C + y] : __int_as_float(0x7fc00000); rows[2 * r + 1] = (am == y) ? 1.f : 0.f; }
}
// deterministic: lane-strided partial sums in a fixed order, fixed-order cross-wave combine
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ rows, float* __restrict__ out, int B, int ncol) {
  fp16_sat_on();
  __shared__ float sm[16];
  for (int c = 0; c < ncol; ++c) {
    float a = 0.f;
    for (int r = threadIdx.x; r < B; r += blockDim.x) a += rows[(size_t)r * ncol + c];
    const float t = block_sum(a, sm);
    if (threadIdx.x == 0) out[c] = t;
  }
}
extern "C" int gsl_ce_fwd(const float* logits, const int64_t* labels, float* out2, float* row_ws, int B, int C, gsl_stream_t s) {
  GSL_CHECK_ARG(logits && labels && out2 && row_ws && B > 0 && C > 0, "null/size");
  hipLaunchKernelGGL(ce_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(s), logits, labels, row_ws, B, C);
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, as_stream(s), row_ws, out2, B, 2);
  return check_launch("gsl_ce_fwd");
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     const float* __restrict__ coef, float scale, float* dlogits, int B, int C,
                                                     int accumulate) {
  fp16_sat_on();
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= B) return;
  float mx, lse; int am;
  row_softmax_stats(logits + (size_t)r * C, C, lane, mx, lse, am);
  const float k = coef[0] * scale;
  const long yl = (long)labels[r];
  const bool ok = yl >= 0 && yl < C;      // out-of-range label: NaN gradient row (see ce_rows_kernel)
  const int y = ok ? (int)yl : -1;
  for (int c = lane; c < C; c += 64) {
    const float g = ok ? k * (expf(logits[(size_t)r * C + c] - lse) - (c == y ? 1.f : 0.f)) : __int_as_float(0x7fc00000);
    float* d = dlogits + (size_t)r * C + c;
    *d = accumulate ? (*d + g) : g;
  }
}
extern "C" int gsl_ce_bwd(const float* logits, const int64_t* labels, const float* coef, float scale, float* dlogits, int B,
                          int C, int accumulate, gsl_stream_t s) {
  GSL_CHECK_ARG(logits && labels && coef && dlogits && B > 0 && C > 0, "null/size");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(s), logits, labels, coef, scale, dlogits, B, C, accumulate);
  return check_launch("gsl_ce_bwd");
}

// ------------------------------------------------------------------ K13 prototype KL
__device__ __forceinline__ float row_lse(const float* row, int D, int lane) {
  float m = -3.0e38f;
  for (int d = lane; d < D; d += 64) m = fmaxf(m, row[d]);
  m = wave_max(m);
  float se = 0.f;
  for (int d = lane; d < D; d += 64) se += expf(row[d] - m);
  return m + logf(wave_sum(se));
}

__global__ __launch_bounds__(256) void proto_kl_rows_kernel(const float* __restrict__ emb, const int64_t* __restrict__ labels,
                                                            const float* __restrict__ proto, float* __restrict__ rows, int B, int D, int C) {
  fp16_sat_on();
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= B) return;
  const float* a = emb + (size_t)r * D;
  const long y = (long)labels[r];
  // a label outside the prototype table (the reference raises KeyError, engine_cl.py:587-589): no out-of-bounds read, the loss turns
  // NaN — a class missing INSIDE the table holds NaN rows (losses.prototype_table), with the same effect
  if (y < 0 || y >= C) { if (lane == 0) rows[r] = __int_as_float(0x7fc00000); return; }
  const float* t = proto + (size_t)y * D;
  const float la = row_lse(a, D, lane), lt = row_lse(t, D, lane);
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float ltd = t[d] - lt;
    acc += expf(ltd) * (ltd - (a[d] - la));
  }
  acc = wave_sum(acc);
  if (lane == 0) rows[r] = acc;
}
extern "C" int gsl_proto_kl_fwd(const float* emb, const int64_t* labels, const float* proto, float* out1, float* row_ws, int B,
                                int D, int C, gsl_stream_t s) {
  GSL_CHECK_ARG(emb && labels && proto && out1 && row_ws && B > 0 && D > 0 && C > 0, "null/size");
  hipLaunchKernelGGL(proto_kl_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(s), emb, labels, proto, row_ws, B, D, C);
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, as_stream(s), row_ws, out1, B, 1);
  return check_launch("gsl_proto_kl_fwd");
}

__global__ __launch_bounds__(256) void proto_kl_bwd_kernel(const float* __restrict__ emb, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ proto, const float* __restrict__ coef,
                                                           float scale, float* demb, int B, int D, int C, int accumulate) {
  fp16_sat_on();
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= B) return;
  const float* a = emb + (size_t)r * D;
  const long y = (long)labels[r];
  if (y < 0 || y >= C) {       // see proto_kl_rows_kernel
    for (int d = lane; d < D; d += 64) demb[(size_t)r * D + d] = __int_as_float(0x7fc00000);
    return;
  }
  const float* t = proto + (size_t)y * D;
  const float la = row_lse(a, D, lane), lt = row_lse(t, D, lane);
  const float k = coef[0] * scale;
  for (int d = lane; d < D; d += 64) {
    const float g = k * (expf(a[d] - la) - expf(t[d] - lt));
    float* o = demb + (size_t)r * D + d;
    *o = accumulate ? (*o + g) : g;
  }
}
extern "C" int gsl_proto_kl_bwd(const float* emb, const int64_t* labels, const float* proto, const float* coef, float scale,
                                float* demb, int B, int D, int C, int accumulate, gsl_stream_t s) {
  GSL_CHECK_ARG(emb && labels && proto && coef && demb && B > 0 && D > 0 && C > 0, "null/size");
  hipLaunchKernelGGL(proto_kl_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(s), emb, labels, proto, coef, scale, demb, B, D, C, accumulate);
  return check_launch("gsl_proto_kl_bwd");
}

// =====================================================================================
// The scalar tail of the step (engine_cl.py:65-125): total = beta*relu(BND - CE_f) + CE_r + alpha*L_s + w_f*relu(BND_pro - KL_f)
// + w_r*KL_r from the SUMS produced by the kernels above, the 8 meter values, and the 5 partial derivatives the backward hands
// to those kernels as upstream gradients. One thread: it replaces ~35 one-element torch kernels per step (3.5 % of the step at the
// reference's batch 48, where every launch counts). Same f32 operations in the same order as the torch expression it replaces.
// =====================================================================================
__global__ void loss_combine_kernel(const float* ce_r_sum, const float* ce_f_sum, const float* kl_f_sum, const float* kl_r_sum,
                                    const float* structure, const float* hit_r, const float* hit_f, float n_r, float n_f, float beta,
                                    float BND, float alpha, float w_f, float w_r, float BND_pro, float* total, float* meters,
                                    float* coefs) {
  fp16_sat_on();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float loss_remain = ce_r_sum[0] / n_r;
  const float hinge_f = BND - ce_f_sum[0] / n_f;
  const float loss_forget = fmaxf(hinge_f, 0.f);
  const float st = structure ? structure[0] : 0.f;
  float pro_f = 0.f, pro_r = 0.f, hinge_p = 0.f;
  if (kl_f_sum) { hinge_p = BND_pro - kl_f_sum[0] / n_f; pro_f = w_f * fmaxf(hinge_p, 0.f); }
  if (kl_r_sum) pro_r = w_r * (kl_r_sum[0] / n_r);
  const float tot = loss_forget * beta + loss_remain + st * alpha + (pro_f + pro_r);
  total[0] = tot;
  meters[0] = beta * loss_forget; meters[1] = loss_remain; meters[2] = tot; meters[3] = alpha * st;
  meters[4] = hit_f[0] * (100.0f / n_f); meters[5] = hit_r[0] * (100.0f / n_r); meters[7] = pro_r;
  // without the prototype term the reference still LOGS w_f * relu(BND_pro - 0) in losses_prototype_forget (engine_cl.py:103-110,
  // engine.py:118-125: prototype_loss_forget is the constant 0 there); the total does not contain it
  meters[6] = kl_f_sum ? pro_f : w_f * fmaxf(BND_pro, 0.f);
  coefs[0] = 1.0f / n_r;                                     // d total / d ce_r_sum
  coefs[1] = hinge_f > 0.f ? -beta / n_f : 0.f;              // d total / d ce_f_sum   (relu'(0) = 0 as in torch)
  coefs[2] = (kl_f_sum && hinge_p > 0.f) ? -w_f / n_f : 0.f; // d total / d kl_f_sum
  coefs[3] = kl_r_sum ? w_r / n_r : 0.f;                     // d total / d kl_r_sum
  coefs[4] = alpha;                                          // d total / d structure
}
// Data-parallel form: the eight batch sums arrive as ONE all-reduced device array (gslora_hip/step.py packs and sum-all-reduces them
// before the hinges so that relu(BND - mean CE_f) / relu(BND_pro - mean KL_f) see the GLOBAL batch means, the reference's single-GPU /
// nn.DataParallel semantics, train_own_forget_cl.py:494-497): pack8 = [ce_r, ce_f, hit_r, hit_f, n_r, n_f, kl_f, kl_r]. The batch sizes
// are read from the pack, so no host value depends on the other ranks.
__global__ void loss_combine_pack_kernel(const float* pack, const float* structure, int has_proto, float beta, float BND, float alpha,
                                         float w_f, float w_r, float BND_pro, float* total, float* meters, float* coefs) {
  fp16_sat_on();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float n_r = pack[4], n_f = pack[5];
  const float loss_remain = pack[0] / n_r;
  const float hinge_f = BND - pack[1] / n_f;
  const float loss_forget = fmaxf(hinge_f, 0.f);
  const float st = structure ? structure[0] : 0.f;
  float pro_f = 0.f, pro_r = 0.f, hinge_p = 0.f;
  if (has_proto) { hinge_p = BND_pro - pack[6] / n_f; pro_f = w_f * fmaxf(hinge_p, 0.f); pro_r = w_r * (pack[7] / n_r); }
  const float tot = loss_forget * beta + loss_remain + st * alpha + (pro_f + pro_r);
  total[0] = tot;
  meters[0] = beta * loss_forget; meters[1] = loss_remain; meters[2] = tot; meters[3] = alpha * st;
  meters[4] = pack[3] * (100.0f / n_f); meters[5] = pack[2] * (100.0f / n_r); meters[7] = pro_r;
  meters[6] = has_proto ? pro_f : w_f * fmaxf(BND_pro, 0.f);
  coefs[0] = 1.0f / n_r;
  coefs[1] = hinge_f > 0.f ? -beta / n_f : 0.f;
  coefs[2] = (has_proto && hinge_p > 0.f) ? -w_f / n_f : 0.f;
  coefs[3] = has_proto ? w_r / n_r : 0.f;
  coefs[4] = alpha;
}
extern "C" int gsl_loss_combine_pack(const float* pack8, const float* structure, int has_proto, float beta, float BND, float alpha,
                                     float w_f, float w_r, float BND_pro, float* total, float* meters8, float* coefs5, gsl_stream_t s) {
  GSL_CHECK_ARG(pack8 && total && meters8 && coefs5, "null");
  hipLaunchKernelGGL(loss_combine_pack_kernel, dim3(1), dim3(64), 0, as_stream(s), pack8, structure, has_proto, beta, BND, alpha, w_f, w_r,
                     BND_pro, total, meters8, coefs5);
  return check_launch("gsl_loss_combine_pack");
}
extern "C" int gsl_loss_combine(const float* ce_r_sum, const float* ce_f_sum, const float* kl_f_sum, const float* kl_r_sum,
                                const float* structure, const float* hit_r, const float* hit_f, float n_r, float n_f, float beta,
                                float BND, float alpha, float w_f, float w_r, float BND_pro, float* total, float* meters8,
                                float* coefs5, gsl_stream_t s) {
  GSL_CHECK_ARG(ce_r_sum && ce_f_sum && hit_r && hit_f && total && meters8 && coefs5 && n_r > 0.f && n_f > 0.f, "null/size");
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, as_stream(s), ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, hit_r,
                     hit_f, n_r, n_f, beta, BND, alpha, w_f, w_r, BND_pro, total, meters8, coefs5);
  return check_launch("gsl_loss_combine");
}

// =====================================================================================
// The whole loss section of a single-process step in ONE launch (the launch-bound regime: few-shot batches replay ~20 one-block kernels
// here — CE rows + sums for the remain and forget rows, prototype KL rows + sums, the scalar tail, and the four backward kernels that
// turn its five coefficients into dlogits / demb). One workgroup of 16 waves: a wave owns every 16th row; row statistics stay in LDS
// between the forward and the backward half. Same device functions, same fixed summation orders as the separate kernels
// (sum_rows_kernel's 256-lane partition included): coefficients and gradients bit-identical to the multi-launch path.
// rows [0, nr) are the remain batch, [nr, N) the forget batch (engine_cl.py:59-125); N <= GSL_LOSS_TAIL_MAX_ROWS.
// =====================================================================================
constexpr int LT_MAX = 256, LT_V = 16;      // rows per launch; values per lane of a row held in registers (C, D <= 64 * LT_V)
// a row in registers: element lane + 64 i in v[i] (the lane-strided order of row_softmax_stats / row_lse); every pass over the row then
// runs from registers — the workgroup is alone on its CU, each global pass would be an exposed L2 round trip
__device__ __forceinline__ void lt_load(const float* row, int n, int lane, float v[LT_V]) {
#pragma unroll
  for (int i = 0; i < LT_V; ++i) { const int c = lane + 64 * i; v[i] = c < n ? row[c] : 0.f; }
}
__device__ __forceinline__ float lt_lse(const float v[LT_V], int n, int lane) {      // = row_lse
  float m = -3.0e38f;
#pragma unroll
  for (int i = 0; i < LT_V; ++i) if (lane + 64 * i < n) m = fmaxf(m, v[i]);
  m = wave_max(m);
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < LT_V; ++i) if (lane + 64 * i < n) se += expf(v[i] - m);
  return m + logf(wave_sum(se));
}
__global__ __launch_bounds__(1024) void loss_tail_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int N, int nr,
                                                         int C, const float* __restrict__ emb, const float* __restrict__ proto, int D,
                                                         int Cp, const float* structure, float beta, float BND, float alpha, float w_f,
                                                         float w_r, float BND_pro, float* out14, float* __restrict__ dlogits,
                                                         float* __restrict__ demb) {
  fp16_sat_on();
  __shared__ float ce_s[LT_MAX], hit_s[LT_MAX], kl_s[LT_MAX], lse_s[LT_MAX], la_s[LT_MAX], lt_s[LT_MAX];
  __shared__ float sm[16], coef_s[5];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nf = N - nr;
  for (int r = wave; r < N; r += 16) {
    float lg[LT_V];
    lt_load(logits + (size_t)r * C, C, lane, lg);
    // row_softmax_stats on the registers: max with the first-index tie-break, then the log-sum-exp
    float m = -3.0e38f; int mi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < LT_V; ++i) { const int c = lane + 64 * i; if (c < C && lg[i] > m) { m = lg[i]; mi = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(m, o, 64); const int oi = __shfl_xor(mi, o, 64);
      if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < LT_V; ++i) if (lane + 64 * i < C) se += expf(lg[i] - m);
    se = wave_sum(se);
    const float lse = m + logf(se);
    const long yl = (long)labels[r];
    const bool y_ok = yl >= 0 && yl < C;      // out-of-range label: NaN loss / gradient row, no out-of-bounds read (see ce_rows_kernel)
    const int y = y_ok ? (int)yl : -1;
    if (lane == 0) { ce_s[r] = y_ok ? lse - logits[(size_t)r * C + y] : __int_as_float(0x7fc00000); hit_s[r] = (mi == y) ? 1.f : 0.f; lse_s[r] = lse; }
    if (emb) {
      if (yl < 0 || yl >= Cp) { if (lane == 0) kl_s[r] = __int_as_float(0x7fc00000); continue; }
      float a[LT_V], t[LT_V];
      lt_load(emb + (size_t)r * D, D, lane, a);
      lt_load(proto + (size_t)yl * D, D, lane, t);
      const float la = lt_lse(a, D, lane), lt = lt_lse(t, D, lane);
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < LT_V; ++i) if (lane + 64 * i < D) {
        const float ltd = t[i] - lt;
        acc += expf(ltd) * (ltd - (a[i] - la));
      }
      acc = wave_sum(acc);
      if (lane == 0) { kl_s[r] = acc; la_s[r] = la; lt_s[r] = lt; }
    }
  }
  __syncthreads();
  // the six batch sums, in sum_rows_kernel's order (256 lanes stride the rows, fixed-order combine; the other waves add exact zeros)
  auto range_sum = [&](const float* v, int base, int n) {
    float a = 0.f;
    if (threadIdx.x < 256) for (int r = threadIdx.x; r < n; r += 256) a += v[base + r];
    return block_sum(a, sm);
  };
  const float ce_r = range_sum(ce_s, 0, nr), hit_r = range_sum(hit_s, 0, nr);
  const float ce_f = range_sum(ce_s, nr, nf), hit_f = range_sum(hit_s, nr, nf);
  float kl_f = 0.f, kl_r = 0.f;
  if (emb) { kl_f = range_sum(kl_s, nr, nf); kl_r = range_sum(kl_s, 0, nr); }
  if (threadIdx.x == 0) {      // gsl_loss_combine, operation for operation
    const float n_r = (float)nr, n_f = (float)nf;
    const float loss_remain = ce_r / n_r;
    const float hinge_f = BND - ce_f / n_f;
    const float loss_forget = fmaxf(hinge_f, 0.f);
    const float st = structure ? structure[0] : 0.f;
    float pro_f = 0.f, pro_r = 0.f, hinge_p = 0.f;
    if (emb) { hinge_p = BND_pro - kl_f / n_f; pro_f = w_f * fmaxf(hinge_p, 0.f); pro_r = w_r * (kl_r / n_r); }
    const float tot = loss_forget * beta + loss_remain + st * alpha + (pro_f + pro_r);
    float* meters = out14 + 1;
    float* coefs = out14 + 9;
    out14[0] = tot;
    meters[0] = beta * loss_forget; meters[1] = loss_remain; meters[2] = tot; meters[3] = alpha * st;
    meters[4] = hit_f * (100.0f / n_f); meters[5] = hit_r * (100.0f / n_r); meters[7] = pro_r;
    meters[6] = emb ? pro_f : w_f * fmaxf(BND_pro, 0.f);
    coef_s[0] = coefs[0] = 1.0f / n_r;
    coef_s[1] = coefs[1] = hinge_f > 0.f ? -beta / n_f : 0.f;
    coef_s[2] = coefs[2] = (emb && hinge_p > 0.f) ? -w_f / n_f : 0.f;
    coef_s[3] = coefs[3] = emb ? w_r / n_r : 0.f;
    coef_s[4] = coefs[4] = alpha;
  }
  __syncthreads();
  for (int r = wave; r < N; r += 16) {      // gsl_ce_bwd / gsl_proto_kl_bwd with the coefficients above (upstream gradient 1)
    const float k = coef_s[r < nr ? 0 : 1] * 1.0f;
    const long yl = (long)labels[r];
    const bool y_ok = yl >= 0 && yl < C;
    const int y = y_ok ? (int)yl : -1;
    const float lse = lse_s[r];
    float lg[LT_V];
    lt_load(logits + (size_t)r * C, C, lane, lg);
#pragma unroll
    for (int i = 0; i < LT_V; ++i) { const int c = lane + 64 * i; if (c < C) dlogits[(size_t)r * C + c] = y_ok ? k * (expf(lg[i] - lse) - (c == y ? 1.f : 0.f)) : __int_as_float(0x7fc00000); }
    if (emb) {
      if (yl < 0 || yl >= Cp) {
        for (int d = lane; d < D; d += 64) demb[(size_t)r * D + d] = __int_as_float(0x7fc00000);
        continue;
      }
      float a[LT_V], t[LT_V];
      lt_load(emb + (size_t)r * D, D, lane, a);
      lt_load(proto + (size_t)yl * D, D, lane, t);
      const float la = la_s[r], lt = lt_s[r];
      const float kk = coef_s[r < nr ? 3 : 2] * 1.0f;
#pragma unroll
      for (int i = 0; i < LT_V; ++i) { const int d = lane + 64 * i; if (d < D) demb[(size_t)r * D + d] = kk * (expf(a[i] - la) - expf(t[i] - lt)); }
    }
  }
}
extern "C" int gsl_loss_tail_max_rows(void) { return LT_MAX; }
extern "C" int gsl_loss_tail(const float* logits, const int64_t* labels, int N, int nr, int C, const float* emb, const float* proto, int D,
                             int Cp, const float* structure, float beta, float BND, float alpha, float w_f, float w_r, float BND_pro,
                             float* out14, float* dlogits, float* demb, gsl_stream_t s) {
  GSL_CHECK_ARG(logits && labels && out14 && dlogits && N > 0 && N <= LT_MAX && nr > 0 && nr < N && C > 0 && C <= 64 * LT_V && D <= 64 * LT_V,
                "null/size (0 < nr < N <= 256 rows, C and D <= 1024)");
  GSL_CHECK_ARG(!emb || (proto && demb && D > 0 && Cp > 0), "prototype term: emb, proto and demb together");
  hipLaunchKernelGGL(loss_tail_kernel, dim3(1), dim3(1024), 0, as_stream(s), logits, labels, N, nr, C, emb, proto, D, Cp, structure, beta, BND,
                     alpha, w_f, w_r, BND_pro, out14, dlogits, demb);
  return check_launch("gsl_loss_tail");
}
