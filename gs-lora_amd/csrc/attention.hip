// attention.hip — softmax(Q Kᵀ · scale) V for head_dim 64, no mask, T <= 224 tokens
// (reference vit_pytorch_face/vit_face.py:358-376; T = 197 for both ViT-P8S8 and ViT-B/16).
//
// One workgroup per (image, head): the whole K/V (or Q/dO) panel of a head fits in LDS
// (197 x 64 bf16 = 25 KB), so there is no online-softmax loop — each wave owns 16-query (or
// 16-key) tiles and keeps a full score row-block in registers.
//
// bf16 path (MFMA v_mfma_f32_16x16x32_bf16, f32 accumulate):
//   * scores are computed TRANSPOSED (Sᵀ = K Qᵀ) so that a lane owns one query column: the row
//     softmax is an in-lane reduction + two xor-shuffles (16, 32), and the probabilities are already
//     in B-operand layout for the second matmul (Oᵀ = Vᵀ Pᵀ) — no LDS round trip for P.
//   * the k-slot permutation trick: B-operand slot (g, idx) of the second MFMA holds key
//     32*pair + 16*(idx/4) + 4*g + idx%4 — exactly what the C layout of two adjacent score tiles
//     delivers — and the A operand (Vᵀ) is gathered from the row-major V panel with the same
//     permutation by two LDS transpose reads (ds_read_b64_tr_b16), so the contraction is unchanged.
//   * backward has two phases with no atomics: dQ (waves own query tiles; K and V panels in LDS)
//     and dK/dV (waves own key tiles; Q and dO panels in LDS); probabilities are recomputed
//     from the saved log-sum-exp (flash-style), delta = rowsum(dO∘O) is produced by the dQ phase.
//     T > 64 runs both phases in ONE launch (attn_bwd_fused_bf16_kernel); the two-kernel form stays for
//     T <= 64 and as the bit-exact reference of the tests.
//   * forward, T in (64, 208] and enough (image, head) items: a persistent wave-specialised kernel
//     (attn_fwd_bf16_pers_kernel: 13 compute waves + 3 loader waves per CU) streams the next item's
//     panels under the current item's softmax; bit-identical to the one-item-per-workgroup kernel.
// f32 path (parity mode): v_mfma_f32_16x16x4_f32 kernels (exact f32: a k-ordered fmaf chain per output), K / V or Q / dO panels in LDS;
//   the cls-query kernels of the last block are templated on the element type.
#include <stdlib.h>

#include <type_traits>

#include "gsl_common.h"

using namespace gsl;
#include "gsl_h16.h"
GSL_OPNS_BEGIN

typedef op16x8_t bf16x8_t;      // MFMA operand in this translation unit's 16-bit format (gsl_common.h)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int HD = 64;    // head dim
// Row-major LDS panels are read two ways: ds_read_b128 row fragments (lane l: row l % 16, 16-byte chunk l / 16) and
// ds_read_b64_tr_b16 transpose reads (32-lane halves = 8 consecutive rows, 8 banks each). 160-byte rows (40 banks): the 8 rows of a
// transpose read land on 8 disjoint bank octets, and the ds_read_b128 service groups of gfx950 ({0-3, 12-15, 20-27}, ... — NOT 16
// consecutive lanes, MI355X_MICROARCH.md section LDS) see 16 distinct 4-bank groups. PMC SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE in
// the fused backward: 144-byte rows 45 %, 160-byte rows with the chunk index XOR-ed by bit 3 of the row (round 1, designed for
// 16-consecutive-lane groups) 31 %, 160-byte rows as they are 0 % (tools/probes/attn_variants.sh; 176-byte rows: 45 %).
constexpr int KLD = 80;
__device__ __forceinline__ int lds_off(int row, int col) { return row * KLD + col; }     // element offset of (row, col) in a panel

union Frag {
  uint4 u;
  uint2 h[2];
  bf16x8_t v;
};

__device__ __forceinline__ f32x4_t mfma16(const bf16x8_t a, const bf16x8_t b, const f32x4_t c) {
  return GSL_MFMA16(a, b, c, 0, 0, 0);
}

// stage a [T][64] bf16 panel (row stride ld elements in global) into row-major LDS [TP][KLD], zero rows >= T
template <int TP>
__device__ __forceinline__ void stage_rowmajor(bf16_t* dst, const bf16_t* src, long ld, int T) {
  for (int idx = threadIdx.x; idx < TP * 8; idx += blockDim.x) {
    const int t = idx >> 3, c = idx & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t < T) v = *reinterpret_cast<const uint4*>(src + (size_t)t * ld + c * 8);
    *reinterpret_cast<uint4*>(dst + lds_off(t, c * 8)) = v;
  }
}
// two panels at once, every global load issued before the first LDS store: the rolled loop above waits for each 16-byte load before it
// stores it (load, s_waitcnt vmcnt(0), ds_write per iteration: 8 serialised HBM round trips for a K + V pair, ~14 k of the 57 k cycles
// a fused-backward workgroup lives — profiles/r04_notes.md). NT = threads of the workgroup.
template <int TP, int NT>
__device__ __forceinline__ void stage_rowmajor2(bf16_t* d0, const bf16_t* s0, bf16_t* d1, const bf16_t* s1, long ld, int T) {
  constexpr int NIT = (TP * 8 + NT - 1) / NT;
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t v0[NIT], v1[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = threadIdx.x + it * NT, t = min(idx >> 3, T - 1), c = idx & 7;
    v0[it] = *reinterpret_cast<const u32x4_t*>(s0 + (size_t)t * ld + c * 8);
    v1[it] = *reinterpret_cast<const u32x4_t*>(s1 + (size_t)t * ld + c * 8);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = threadIdx.x + it * NT, t = idx >> 3, c = idx & 7;
    if (idx < TP * 8) {
      const u32x4_t z = {0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4_t*>(d0 + lds_off(t, c * 8)) = (t < T) ? v0[it] : z;
      *reinterpret_cast<u32x4_t*>(d1 + lds_off(t, c * 8)) = (t < T) ? v1[it] : z;
    }
  }
}
// the same for the per-item kernels, whose workgroup size is a launch parameter (>= MINT threads) and whose two panels may have different
// row strides: iteration count for MINT threads, the stride is blockDim.x, out-of-range iterations are predicated off. (With the count for
// 256 threads the 1024-thread forward of the few-shot regime carried 14 dead vector registers sets and SPILLED: 0.84 -> 1.00 ms per step.)
template <int TP, int MINT>
__device__ __forceinline__ void stage_rowmajor2_rt(bf16_t* d0, const bf16_t* s0, long ld0, bf16_t* d1, const bf16_t* s1, long ld1, int T) {
  constexpr int NIT = (TP * 8 + MINT - 1) / MINT;      // MINT = the smallest workgroup the kernel is launched with
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t v0[NIT], v1[NIT];
  const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = threadIdx.x + it * blockDim.x, t = idx >> 3, c = idx & 7;
    v0[it] = z; v1[it] = z;
    if (idx < TP * 8 && t < T) {
      v0[it] = *reinterpret_cast<const u32x4_t*>(s0 + (size_t)t * ld0 + c * 8);
      v1[it] = *reinterpret_cast<const u32x4_t*>(s1 + (size_t)t * ld1 + c * 8);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = threadIdx.x + it * blockDim.x, t = idx >> 3, c = idx & 7;
    if (idx < TP * 8) {
      *reinterpret_cast<u32x4_t*>(d0 + lds_off(t, c * 8)) = v0[it];
      *reinterpret_cast<u32x4_t*>(d1 + lds_off(t, c * 8)) = v1[it];
    }
  }
}
__device__ __forceinline__ bf16x8_t lds_frag_rm(const bf16_t* base, int row, int ks, int fc) {
  return *reinterpret_cast<const bf16x8_t*>(base + lds_off(row, ks * 32 + fc * 8));
}
// A operand X^T[dt*16 + fr][keys] in the k-slot permutation (slot (g, idx) = key 32*pair + 16*(idx/4) + 4*g + idx%4), gathered from the
// ROW-MAJOR panel X[key][KLD] with gfx950's LDS transpose read (ds_read_b64_tr_b16).
// Within a 16-lane group lane i supplies the address of 4 contiguous elements (row i/4, columns 4*(i%4)..) of a 4 x 16 block and lane
// l receives column l%16 of that block (probed on MI355X, tools/probes/tr_read_probe.hip): with block rows = keys
// 32*pair + 16*q + 4*g .. +3 (g = lane group, q = which of the two reads) lane l ends up with X[those keys][dt*16 + l%16] in k-slots
// 4q..4q+3 — the permutation above. No transposed LDS image, no register-transposed staging pass
// (that pass and the second copy cost 21 % of the backward: 1103 -> 870 us at B = 1024).
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t* lds_v4s_p;
__device__ __forceinline__ bf16x8_t lds_frag_trr(const bf16_t* base, int dt, int pair, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int row = pair * 32 + 4 * g + (i >> 2), col = dt * 16 + (i & 3) * 4;
  union { v4s_t h[2]; bf16x8_t v; } f;
  f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(base + lds_off(row, col)));
  f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(base + lds_off(row + 16, col)));
  return f.v;
}
__device__ __forceinline__ bf16x8_t gl_frag(const bf16_t* rowptr, int ks, int fc) {
  return *reinterpret_cast<const bf16x8_t*>(rowptr + ks * 32 + fc * 8);
}
__device__ __forceinline__ void store4bf(bf16_t* p, const f32x4_t v, float mul) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2o(v[0] * mul, v[1] * mul), pack2o(v[2] * mul, v[3] * mul));
}

// Item order (round 4): hardware places workgroup w on XCD w % 8, and the tensors o / dO / dqkv are TOKEN-major — the H heads of an image own
// adjacent 128-byte segments of the same rows. With items in plain (image, head) order the heads of one image are spread over all 8
// XCDs (8 L2s each see one 128-byte piece of every row). item_remap() hands the 8 consecutive workgroups of ONE XCD the heads of ONE image
// (B % 8 == 0; otherwise the plain order): rows are then read / written whole by one L2 within a short window.
// seq = position in dispatch order (blockIdx, or blockIdx + k * gridDim of a persistent kernel with gridDim % 8 == 0)
__device__ __forceinline__ int item_remap(int seq, int H, int on) {
  if (!on) return seq;
  const int x = seq & 7, j = seq >> 3;
  return ((j / H) * 8 + x) * H + (j % H);
}

// hm (layout of the qkv INPUT of the bf16 kernels): 0 = token-major [B*T, 3*H*64] (row stride 3*H*64; the three panels of a head are 64
// columns wide at column offsets h*64, (H+h)*64, (2H+h)*64), 1 = head-major [B][H][3][T][64] (each (image, head) item is one contiguous
// block of three [T, 64] panels: every panel row is a full 128-byte line next to its neighbours instead of a 128-byte segment every
// 3*H*128 bytes). Outputs (o, dqkv) are always token-major — they are A operands of row-major GEMMs.
// =====================================================================================
// forward (bf16)
// =====================================================================================
// NT = threads per workgroup: 512 (8 waves own two query tiles each), or 1024 when there are fewer (image, head) items than CUs (few-shot
// batches): 16 waves own one tile each — the item is a latency chain on an otherwise idle CU, half as long with twice the waves.
template <int NKT, int NT = 512>
__global__ __launch_bounds__(NT, (NT == 512 ? 2 : 4)) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                            float* __restrict__ lse, int T, int H, float scale, int abl, int hm) {
  GSL_OP16_KERNEL_ENTRY();
  constexpr int TP = NKT * 16;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[TP * KLD];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[TP * KLD];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long ldi = hm ? (long)HD : 3L * H * HD, ko = hm ? (long)T * HD : (long)H * HD;
  const bf16_t* qb = qkv + (hm ? (size_t)(b * H + h) * 3 * T * HD : (size_t)b * T * ldi + h * HD);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fc = lane >> 4;
  bf16x8_t qn0, qn1;   // Q fragments of the NEXT query tile: their global-load latency hides under this tile's work
  {                    // (the first tile's are issued before the panel staging and land under it)
    const bf16_t* qrow = qb + (size_t)min(wave * 16 + fr, T - 1) * ldi;
    qn0 = gl_frag(qrow, 0, fc); qn1 = gl_frag(qrow, 1, fc);
  }
  if (abl != 2) stage_rowmajor2_rt<TP, (NKT == 4 ? 256 : NT)>(Ks, qb + ko, ldi, Vs, qb + 2 * ko, ldi, T);
  __syncthreads();
  if (abl == 1) return;
  const int nqt = (T + 15) / 16;
  const int ktf = T >> 4;      // key tiles below this index are completely valid
  const int nwaves = blockDim.x >> 6;
  for (int qt = wave; qt < nqt; qt += nwaves) {
    asm volatile("" ::: "memory");   // keep the K / V^T fragment reads inside the loop (LICM would pin 224 VGPRs)
    const int qr = qt * 16 + fr;
    const bf16x8_t qf0 = qn0, qf1 = qn1;
    if (qt + nwaves < nqt) {
      const bf16_t* qrow = qb + (size_t)min((qt + nwaves) * 16 + fr, T - 1) * ldi;
      qn0 = gl_frag(qrow, 0, fc); qn1 = gl_frag(qrow, 1, fc);
    }
    f32x4_t s[NKT];
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      acc = mfma16(lds_frag_rm(Ks, kt * 16 + fr, 0, fc), qf0, acc);
      acc = mfma16(lds_frag_rm(Ks, kt * 16 + fr, 1, fc), qf1, acc);
      // the softmax is VALU-bound (52 scores per lane and query tile): the running max is taken on the raw scores (scale > 0), the
      // scale and log2(e) are folded into one fma in front of v_exp_f32, and only key tiles >= T/16 (a wave-uniform test) hold padded keys
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kt >= ktf) { if (kt * 16 + fc * 4 + r >= T) acc[r] = -3.0e38f; }
        m = fmaxf(m, acc[r]);
      }
      s[kt] = acc;
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float c2 = scale * 1.4426950408889634f, mc = m * c2;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c2, -mc)); l += s[kt][r]; }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    Frag pf[NKT / 2];
#pragma unroll
    for (int pr = 0; pr < NKT / 2; ++pr) {
      pf[pr].u = make_uint4(pack2o(s[2 * pr][0], s[2 * pr][1]), pack2o(s[2 * pr][2], s[2 * pr][3]),
                            pack2o(s[2 * pr + 1][0], s[2 * pr + 1][1]), pack2o(s[2 * pr + 1][2], s[2 * pr + 1][3]));
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < NKT / 2; ++pr) acc = mfma16(lds_frag_trr(Vs, dt, pr, lane), pf[pr].v, acc);
      // acc[r] = O[q = fr][d = dt*16 + fc*4 + r]
      if (qr < T) store4bf(o + ((size_t)b * T + qr) * (H * HD) + h * HD + dt * 16 + fc * 4, acc, inv);
    }
    if (fc == 0 && qr < T) lse[((size_t)b * H + h) * T + qr] = m * scale + __logf(l);
  }
}

// =====================================================================================
// forward (bf16, 64 < T <= 208), persistent + wave-specialised: one 16-wave workgroup per CU loops over (batch, head) items.
// The one-item-per-workgroup kernel above runs its two phases strictly one after the other chip-wide (measured at B = 1024: panel
// staging alone 112 us, compute alone 165 us, full kernel 277 us = the sum, even with two resident workgroups per CU). Here
//   * waves 0..12 own one query tile each and only compute: phase 1 = Q K^T + softmax (reads the Q and K panels), phase 2 = P V
//     (reads the V panel), one workgroup barrier after each phase;
//   * waves 13..15 only move data: during phase 1 of item i they put V(i) (requested one phase earlier) into LDS and request
//     Q, K of item i+1 into registers; during phase 2 they put those into LDS and request V(i+1). A panel is overwritten in the
//     phase in which nobody reads it, and every HBM request has a whole compute phase to land.
// The loader waves hold up to 72 VGPRs of in-flight data, the compute waves none: the two roles share one register budget (128).
// Barriers are raw s_barrier + lgkmcnt waits: a __syncthreads() would also drain the loaders' outstanding global loads (vmcnt).
// =====================================================================================
// A 16 x 64 tile held in C layout (lane (fr, fc): row fr, columns dt * 16 + 4 fc .. + 3 of acc[dt]) leaves as FULL 128-byte rows: through a
// wave-private LDS area (row stride LD elements), then 16 bytes per lane, 8 lanes per row, 2 store instructions per tile. The fragment-
// layout store (store4bf: 8 bytes per lane, 16 rows x 32 bytes per instruction, 4 instructions per tile) costs ~115 cycles of the CU's
// vector-memory pipe per instruction: 156 of them per item kept that pipe busy for ~18 k cycles and every load queued behind them
// (profiles/r05_j_attn_merged.md). Same values (v * mul rounded once), rows >= nvalid are not written.
template <int LD>
__device__ __forceinline__ void store_tile_rows(bf16_t* stage, const f32x4_t (&acc)[4], float mul, bf16_t* gbase, long gld, int nvalid, int lane) {
  const int fr = lane & 15, fc = lane >> 4;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
    *reinterpret_cast<uint2*>(stage + fr * LD + dt * 16 + fc * 4) =
        make_uint2(pack2o(acc[dt][0] * mul, acc[dt][1] * mul), pack2o(acc[dt][2] * mul, acc[dt][3] * mul));
  const int r = lane >> 3, c = lane & 7;
  const uint4 v0 = *reinterpret_cast<const uint4*>(stage + r * LD + c * 8);
  const uint4 v1 = *reinterpret_cast<const uint4*>(stage + (r + 8) * LD + c * 8);
  if (r < nvalid) *reinterpret_cast<uint4*>(gbase + (size_t)r * gld + c * 8) = v0;
  if (r + 8 < nvalid) *reinterpret_cast<uint4*>(gbase + (size_t)(r + 8) * gld + c * 8) = v1;
}

__device__ __forceinline__ void wg_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// FAST: T > (NKT - 2) * 16, so only key tile NKT - 2 needs the per-element validity mask — a compile-time property of the unrolled
// tile loop. (With the run-time first-partial-tile index the compiler materialises one predicate per element and tile, parks them in
// VGPR lanes and pays two v_readlane + one v_cndmask per score element: 12 of 29 VALU instructions per tile step.) Key tile NKT - 1
// is empty for every T this kernel accepts (T <= 208) and is not computed at all. Same values, same order: bit-identical.
template <int NKT, bool FAST>
__global__ __launch_bounds__(1024) void attn_fwd_bf16_pers_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                                  float* __restrict__ lse, int T, int H, float scale, int nitems, int hm, int imap) {
  GSL_OP16_KERNEL_ENTRY();
  constexpr int TP = NKT * 16;
  constexpr int NCW = 13;                 // compute waves = query tiles (host: T <= 208)
  constexpr int NST = 9;                  // loader steps per panel: 24 rows x 8 chunks per step, 9 * 24 = 216 >= 208 rows
  __shared__ __attribute__((aligned(16))) bf16_t Qs[TP * KLD];
  __shared__ __attribute__((aligned(16))) bf16_t Ks[TP * KLD];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[TP * KLD];
  constexpr int OLD = 72;                 // row stride of the output staging tiles (144 B)
  __shared__ __attribute__((aligned(16))) bf16_t Os[NCW * 16 * OLD];      // one 16 x 64 tile per compute wave: outputs leave as full rows (store_tile_rows)
  const long ld = hm ? (long)HD : 3L * H * HD, ko = hm ? (long)T * HD : (long)H * HD;      // row stride / K-panel offset of the qkv input
  auto item_base = [&](int seq) { const int it = item_remap(seq, H, imap); return qkv + (hm ? (size_t)it * 3 * T * HD : (size_t)(it / H) * T * ld + (it % H) * HD); };
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fr = lane & 15, fc = lane >> 4;
  // rows >= T of the panels are zero for every item: written once
  for (int idx = threadIdx.x; idx < (TP - T) * 8; idx += 1024) {
    const int t = T + (idx >> 3), c = idx & 7;
    const uint4 z = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(Qs + lds_off(t, c * 8)) = z;
    *reinterpret_cast<uint4*>(Ks + lds_off(t, c * 8)) = z;
    *reinterpret_cast<uint4*>(Vs + lds_off(t, c * 8)) = z;
  }
  if (wave >= NCW) {
    // ------------------------------------------------------------------ loader waves
    // (data registers are first-class vector values and every request is unconditional — the last item re-requests itself —
    // so that the three register sets stay in VGPRs: conditional assignments to uint4 arrays ended up in scratch memory)
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const int li = (wave - NCW) * 64 + lane, col = (li & 7) * 8, row0 = li >> 3;
    u32x4_t dq[NST], dk[NST], dv[NST];
    int item = blockIdx.x;
    // Per-lane source offsets as 32-bit BYTE offsets from a wave-uniform panel base (an item's panels span < 2^31 bytes): one VGPR per row
    // instead of a 64-bit address pair — with 108 data registers in flight the pairs cost this kernel 4 spilled VGPRs, and a scratch access
    // rides the CU's in-order vector-memory pipe (code object: .vgpr_spill_count 0 now). The K / V panels are ko / 2 ko
    // elements behind Q: added to the uniform base.
    unsigned goff[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) goff[k] = (unsigned)(min(row0 + 24 * k, T - 1) * (int)ld + col) * 2u;
    auto ld16 = [](const char* base, unsigned off) { return *reinterpret_cast<const u32x4_t*>(base + off); };
    const char* qb = reinterpret_cast<const char*>(item_base(item));
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      dq[k] = ld16(qb, goff[k]);
      dk[k] = ld16(qb + 2 * ko, goff[k]);
    }
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int r = min(row0 + 24 * k, T - 1);      // rows past T - 1 re-write row T - 1 with its own data
      *reinterpret_cast<u32x4_t*>(Qs + lds_off(r, col)) = dq[k];
      *reinterpret_cast<u32x4_t*>(Ks + lds_off(r, col)) = dk[k];
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < NST; ++k) dv[k] = ld16(qb + 4 * ko, goff[k]);
    wg_barrier_lds();                                   // Q, K of the first item are in LDS
    for (; item < nitems; item += gridDim.x) {
      const int nxt = (item + (int)gridDim.x < nitems) ? item + (int)gridDim.x : item;
      const char* nb = reinterpret_cast<const char*>(item_base(nxt));
      // phase 1 of `item`: V(item) -> LDS, request Q, K of the next item
#pragma unroll
      for (int k = 0; k < NST; ++k) *reinterpret_cast<u32x4_t*>(Vs + lds_off(min(row0 + 24 * k, T - 1), col)) = dv[k];
      asm volatile("" ::: "memory");      // requests strictly after the deposit: the register sets never live together
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        dq[k] = ld16(nb, goff[k]);
        dk[k] = ld16(nb + 2 * ko, goff[k]);
      }
      wg_barrier_lds();
      // phase 2 of `item`: Q, K of the next item -> LDS, request its V
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int r = min(row0 + 24 * k, T - 1);
        *reinterpret_cast<u32x4_t*>(Qs + lds_off(r, col)) = dq[k];
        *reinterpret_cast<u32x4_t*>(Ks + lds_off(r, col)) = dk[k];
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < NST; ++k) dv[k] = ld16(nb + 4 * ko, goff[k]);
      wg_barrier_lds();
    }
    return;
  }
  // -------------------------------------------------------------------- compute waves: query tile = wave
  const int ktf = T >> 4;      // key tiles below this index are completely valid
  const float c2 = scale * 1.4426950408889634f;
  const int qr = wave * 16 + fr;
  wg_barrier_lds();
  for (int seq = blockIdx.x; seq < nitems; seq += gridDim.x) {
    const int item = item_remap(seq, H, imap);
    const int b = item / H, h = item % H;
    const bf16x8_t qf0 = lds_frag_rm(Qs, qr, 0, fc), qf1 = lds_frag_rm(Qs, qr, 1, fc);
    constexpr int NKV = NKT - 1;          // key tiles that can hold a valid key (host: T <= (NKT - 1) * 16)
    f32x4_t s[NKV];
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NKV; ++kt) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      acc = mfma16(lds_frag_rm(Ks, kt * 16 + fr, 0, fc), qf0, acc);
      acc = mfma16(lds_frag_rm(Ks, kt * 16 + fr, 1, fc), qf1, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (FAST ? (kt == NKT - 2) : (kt >= ktf)) { if (kt * 16 + fc * 4 + r >= T) acc[r] = -3.0e38f; }
        m = fmaxf(m, acc[r]);
      }
      s[kt] = acc;
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mc = m * c2;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKV; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c2, -mc)); l += s[kt][r]; }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    Frag pf[NKT / 2];
#pragma unroll
    for (int pr = 0; pr < NKT / 2; ++pr) {
      if (2 * pr + 1 < NKV)
        pf[pr].u = make_uint4(pack2o(s[2 * pr][0], s[2 * pr][1]), pack2o(s[2 * pr][2], s[2 * pr][3]),
                              pack2o(s[2 * pr + 1][0], s[2 * pr + 1][1]), pack2o(s[2 * pr + 1][2], s[2 * pr + 1][3]));
      else        // the empty tile's probabilities are exactly zero
        pf[pr].u = make_uint4(pack2o(s[2 * pr][0], s[2 * pr][1]), pack2o(s[2 * pr][2], s[2 * pr][3]), 0u, 0u);
    }
    const float inv = 1.0f / l;
    wg_barrier_lds();                                   // V(item) is in LDS; Q / K panels may be overwritten from here on
    f32x4_t acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      acc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < NKT / 2; ++pr) acc[dt] = mfma16(lds_frag_trr(Vs, dt, pr, lane), pf[pr].v, acc[dt]);
    }
    // (round 5: 2 full-row store instructions per tile instead of 4 fragment-layout ones; `inv` is per ROW of the tile = per lane column fr here:
    //  the scaling happens before the staging, in the accumulator layout, exactly as store4bf did)
    store_tile_rows<OLD>(Os + wave * 16 * OLD, acc, inv, o + ((size_t)b * T + wave * 16) * (H * HD) + h * HD, (long)H * HD, min(16, T - wave * 16), lane);
    if (fc == 0 && qr < T) lse[((size_t)b * H + h) * T + qr] = m * scale + __logf(l);
    wg_barrier_lds();                                   // Q, K of the next item are in LDS; V may be overwritten
  }
}

// =====================================================================================
// backward dQ (bf16): waves own query tiles
// =====================================================================================
template <int NKT>
__global__ __launch_bounds__(512) void attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                               const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                               bf16_t* __restrict__ dqkv, float* __restrict__ delta, int T,
                                                               int H, float scale, int abl, int hm) {
  GSL_OP16_KERNEL_ENTRY();
  constexpr int TP = NKT * 16;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[TP * KLD];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[TP * KLD];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * HD, ldo = (long)H * HD;
  const long ldi = hm ? (long)HD : ld, ko = hm ? (long)T * HD : (long)H * HD;
  const bf16_t* qb = qkv + (hm ? (size_t)(b * H + h) * 3 * T * HD : (size_t)b * T * ld + h * HD);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fc = lane >> 4;
  const int nwaves = blockDim.x >> 6;
  // this wave's first query tile: Q / dO / O fragments come straight from global memory; issued before the panel staging so that
  // their latency lands under it (a wave owns at most two tiles when T = 197)
  bf16x8_t qf0, qf1;
  Frag dof0, dof1, of0, of1;
  auto load_tile = [&](int qt) {
    const int qrc = min(qt * 16 + fr, T - 1);
    const bf16_t* qrow = qb + (size_t)qrc * ldi;
    const bf16_t* dorow = d_o + ((size_t)b * T + qrc) * ldo + h * HD;
    const bf16_t* orow = o + ((size_t)b * T + qrc) * ldo + h * HD;
    qf0 = gl_frag(qrow, 0, fc); qf1 = gl_frag(qrow, 1, fc);
    dof0.v = gl_frag(dorow, 0, fc); dof1.v = gl_frag(dorow, 1, fc);
    of0.v = gl_frag(orow, 0, fc); of1.v = gl_frag(orow, 1, fc);
  };
  load_tile(wave);
  if (abl != 2) stage_rowmajor2_rt<TP, (NKT == 4 ? 256 : 512)>(Ks, qb + ko, ldi, Vs, qb + 2 * ko, ldi, T);
  __syncthreads();
  if (abl == 1) return;
  const int nqt = (T + 15) / 16;
  const int ktf = T >> 4;      // key tiles below this index are completely valid
  for (int qt = wave; qt < nqt; qt += nwaves) {
    asm volatile("" ::: "memory");   // 8 waves per block: keep the fragment reads in the loop (<= 256 registers)
    const int qr = qt * 16 + fr, qrc = min(qr, T - 1);
    if (qt != wave) load_tile(qt);
    float dl = 0.f;
    {
      const uint32_t a[8] = {dof0.u.x, dof0.u.y, dof0.u.z, dof0.u.w, dof1.u.x, dof1.u.y, dof1.u.z, dof1.u.w};
      const uint32_t c[8] = {of0.u.x, of0.u.y, of0.u.z, of0.u.w, of1.u.x, of1.u.y, of1.u.z, of1.u.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float a0, a1, c0, c1;
        unpack2o(a[i], a0, a1); unpack2o(c[i], c0, c1);
        dl += a0 * c0;
        dl += a1 * c1;
      }
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const float c2 = scale * 1.4426950408889634f, lq2 = lse[((size_t)b * H + h) * T + qrc] * 1.4426950408889634f;
    Frag dsf[NKT / 2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      sa = mfma16(lds_frag_rm(Ks, kt * 16 + fr, 0, fc), qf0, sa);
      sa = mfma16(lds_frag_rm(Ks, kt * 16 + fr, 1, fc), qf1, sa);
      dp = mfma16(lds_frag_rm(Vs, kt * 16 + fr, 0, fc), dof0.v, dp);
      dp = mfma16(lds_frag_rm(Vs, kt * 16 + fr, 1, fc), dof1.v, dp);
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {      // dS without the softmax scale: it is applied once to the 16 dQ accumulators below
        float p = __builtin_amdgcn_exp2f(fmaf(sa[r], c2, -lq2));
        if (kt >= ktf) { if (kt * 16 + fc * 4 + r >= T) p = 0.f; }
        ds[r] = p * (dp[r] - dl);
      }
      if ((kt & 1) == 0) { dsf[kt / 2].u.x = pack2o(ds[0], ds[1]); dsf[kt / 2].u.y = pack2o(ds[2], ds[3]); }
      else { dsf[kt / 2].u.z = pack2o(ds[0], ds[1]); dsf[kt / 2].u.w = pack2o(ds[2], ds[3]); }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < NKT / 2; ++pr) acc = mfma16(lds_frag_trr(Ks, dt, pr, lane), dsf[pr].v, acc);
      if (qr < T) store4bf(dqkv + ((size_t)b * T + qr) * ld + h * HD + dt * 16 + fc * 4, acc, scale);
    }
    if (fc == 0 && qr < T) delta[((size_t)b * H + h) * T + qr] = dl;
  }
}

// =====================================================================================
// backward dK/dV (bf16): waves own key tiles
// =====================================================================================
// NT = key tiles owned by a wave: 2 halves the LDS fragment traffic per MFMA (157 VGPRs, one workgroup per CU), 1 fits 128 VGPRs and
// two workgroups per CU — the kernel is latency-bound, so occupancy wins (default for T > 64; GSL_ATTN_NT development knob).
template <int NKT, int NT>
__global__ __launch_bounds__(512, (NT == 1 ? 4 : 2)) void attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                bf16_t* __restrict__ dqkv, int T, int H, float scale, int abl, int hm) {
  GSL_OP16_KERNEL_ENTRY();
  constexpr int TP = NKT * 16;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[TP * KLD];
  __shared__ __attribute__((aligned(16))) bf16_t Os[TP * KLD];   // dO row-major
  __shared__ __attribute__((aligned(16))) float lse_s[TP];
  __shared__ __attribute__((aligned(16))) float del_s[TP];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * HD, ldo = (long)H * HD;
  const long ldi = hm ? (long)HD : ld, ko = hm ? (long)T * HD : (long)H * HD;
  const bf16_t* qb = qkv + (hm ? (size_t)(b * H + h) * 3 * T * HD : (size_t)b * T * ld + h * HD);
  const bf16_t* dob = d_o + (size_t)b * T * ldo + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fc = lane >> 4;
  const int nwaves = blockDim.x >> 6;
  // K / V fragments of the wave's key tiles come straight from global memory: the first group is issued before the panel staging
  bf16x8_t kf[NT][2], vf[NT][2];
  int kr[NT];
  auto load_keys = [&](int kp) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      kr[t] = (kp * NT + t) * 16 + fr;
      const int krc = min(kr[t], T - 1);
      const bf16_t* krow = qb + (size_t)krc * ldi + ko;
      const bf16_t* vrow = qb + (size_t)krc * ldi + 2 * ko;
      kf[t][0] = gl_frag(krow, 0, fc); kf[t][1] = gl_frag(krow, 1, fc);
      vf[t][0] = gl_frag(vrow, 0, fc); vf[t][1] = gl_frag(vrow, 1, fc);
    }
  };
  load_keys(wave);
  if (abl != 2) stage_rowmajor2_rt<TP, (NKT == 4 ? 256 : 512)>(Qs, qb, ldi, Os, dob, ldo, T);
  for (int t = threadIdx.x; t < TP; t += blockDim.x) {
    lse_s[t] = (t < T) ? lse[((size_t)b * H + h) * T + t] * 1.4426950408889634f : 1.0e30f;   // log2 units; padded queries -> p = 0
    del_s[t] = (t < T) ? delta[((size_t)b * H + h) * T + t] : 0.f;
  }
  __syncthreads();
  if (abl == 1) return;
  const int nkt = (T + 15) / 16;
  const float c2 = scale * 1.4426950408889634f;
  // NT = 2: each wave owns TWO adjacent key tiles: every Q / dO / Q^T / dO^T fragment read from LDS feeds two MFMAs (one per
  // key tile). The kernel is LDS-bandwidth-bound (1 KB of fragment reads per MFMA when a wave owns a single tile).
  for (int kp = wave; kp * NT < nkt; kp += nwaves) {
    if (kp != wave) load_keys(kp);
    f32x4_t adk[NT][4], adv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { adk[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; adv[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int qp = 0; qp < NKT / 2; ++qp) {
      Frag pf[NT], dsf[NT];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * qp + half;
        const bf16x8_t q0 = lds_frag_rm(Qs, qt * 16 + fr, 0, fc), q1 = lds_frag_rm(Qs, qt * 16 + fr, 1, fc);
        const bf16x8_t g0 = lds_frag_rm(Os, qt * 16 + fr, 0, fc), g1 = lds_frag_rm(Os, qt * 16 + fr, 1, fc);
        const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[qt * 16 + fc * 4]);
        const float4 d4 = *reinterpret_cast<const float4*>(&del_s[qt * 16 + fc * 4]);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sa = mfma16(q0, kf[t][0], sa);   // S[q = qt*16+fc*4+r][key = fr of tile t]
          sa = mfma16(q1, kf[t][1], sa);
          dp = mfma16(g0, vf[t][0], dp);   // dP[q][key]
          dp = mfma16(g1, vf[t][1], dp);
          float p[4], ds[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p[r] = __builtin_amdgcn_exp2f(fmaf(sa[r], c2, -lv[r]));
            ds[r] = p[r] * (dp[r] - dv[r]);          // softmax scale applied once to the dK accumulators at the store
          }
          if (half == 0) {
            pf[t].u.x = pack2o(p[0], p[1]); pf[t].u.y = pack2o(p[2], p[3]);
            dsf[t].u.x = pack2o(ds[0], ds[1]); dsf[t].u.y = pack2o(ds[2], ds[3]);
          } else {
            pf[t].u.z = pack2o(p[0], p[1]); pf[t].u.w = pack2o(p[2], p[3]);
            dsf[t].u.z = pack2o(ds[0], ds[1]); dsf[t].u.w = pack2o(ds[2], ds[3]);
          }
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8_t ot = lds_frag_trr(Os, dt, qp, lane), qtf = lds_frag_trr(Qs, dt, qp, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          adv[t][dt] = mfma16(ot, pf[t].v, adv[t][dt]);     // dV^T[d][key]
          adk[t][dt] = mfma16(qtf, dsf[t].v, adk[t][dt]);   // dK^T[d][key]
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (kr[t] < T) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          bf16_t* base = dqkv + ((size_t)b * T + kr[t]) * ld + h * HD + dt * 16 + fc * 4;
          store4bf(base + H * HD, adk[t][dt], scale);
          store4bf(base + 2 * H * HD, adv[t][dt], 1.0f);
        }
      }
    }
  }
}

// =====================================================================================
// backward, both phases in one workgroup (bf16, T > 64): the dQ phase (waves own query tiles; K / V panels in LDS) and the dK/dV
// phase (waves own key tiles; Q / dO panels in the SAME LDS) of one (batch, head) run back to back. The two-kernel form streams
// qkv and dO twice and round-trips delta through HBM; here delta = rowsum(dO * O) never leaves LDS, one launch disappears and
// the second phase starts while the CU's other workgroup is still in its first (735 -> 653 us per layer in the step). The panels
// of the second phase still come over the fabric (PMC FETCH_SIZE 1.81 GB vs 1.88 GB for the two kernels: no L2 reuse at a ~10 us
// distance). Same arithmetic, same per-element operation order as the two kernels above: bit-identical results.
// =====================================================================================
// FAST: (NKT - 2) * 16 < T <= (NKT - 1) * 16 (T = 197 with NKT = 14). As in the forward, the validity mask of phase A is then compiled
// for key tile NKT - 2 only, and tile NKT - 1 — no valid key / query, probabilities exactly zero — is left out of both phases at
// compile time (a run-time uniform branch inside the unrolled tile loop makes the compiler sink all exponentials below it and spill
// the score tiles). Same values in the same order as the generic form: bit-identical.
template <int NKT, bool FAST, int NT = 512>      // NT = 1024: sixteen waves, one tile each (fewer items than CUs; see attn_fwd_bf16_kernel)
__global__ __launch_bounds__(NT, 4) void attn_bwd_fused_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                     const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                                     bf16_t* __restrict__ dqkv, int T, int H, float scale,
                                                                     unsigned long long* __restrict__ stamps, int hm, int imap) {
  GSL_OP16_KERNEL_ENTRY();
  constexpr int TP = NKT * 16;
  // development (GSL_ATTN_STAMPS = device address of 256 x 8 u64): cycle stamps of every 64th workgroup
  unsigned long long* dbg = (stamps && blockIdx.x < 64 * 256 && (blockIdx.x % 64) == 0) ? stamps + (blockIdx.x / 64) * 8 : nullptr;   // uniform
#define GSL_ATTN_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = __builtin_readcyclecounter(); } while (0)
  GSL_ATTN_STAMP(0);
  __shared__ __attribute__((aligned(16))) bf16_t P0[TP * KLD];   // phase A: K, phase B: Q
  __shared__ __attribute__((aligned(16))) bf16_t P1[TP * KLD];   // phase A: V, phase B: dO
  __shared__ __attribute__((aligned(16))) float lse_s[TP];       // log2 units; padded queries 1e30 -> p = 0
  __shared__ __attribute__((aligned(16))) float del_s[TP];
  const int item = item_remap(blockIdx.x, H, imap);
  const int b = item / H, h = item % H;
  const long ld = 3L * H * HD, ldo = (long)H * HD;
  const bool nostore = hm & 2;      // development ablation (GSL_ATTN_ABL=4): no output stores
  hm &= 1;
  const long ldi = hm ? (long)HD : ld, ko = hm ? (long)T * HD : (long)H * HD;      // qkv INPUT: row stride, K-panel offset (V at 2 ko)
  const bf16_t* qb = qkv + (hm ? (size_t)(b * H + h) * 3 * T * HD : (size_t)b * T * ld + h * HD);
  const bf16_t* dob = d_o + (size_t)b * T * ldo + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fc = lane >> 4;
  const int nwaves = blockDim.x >> 6;
  const int nqt = (T + 15) / 16;
  const int ktf = T >> 4;      // key tiles below this index are completely valid
  const float c2 = scale * 1.4426950408889634f;
  // ------------------------------------------------------------------ phase A: dQ (and delta, kept in LDS)
  Frag qf0, qf1, dof0, dof1;      // Q / dO fragments of the wave's current query tile (kept past phase A: they are deposited into the phase-B panels)
  {
    Frag of0, of1;
    auto load_tile = [&](int qt) {
      const int qrc = min(qt * 16 + fr, T - 1);
      const bf16_t* qrow = qb + (size_t)qrc * ldi;
      const bf16_t* dorow = dob + (size_t)qrc * ldo;
      const bf16_t* orow = o + ((size_t)b * T + qrc) * ldo + h * HD;
      qf0.v = gl_frag(qrow, 0, fc); qf1.v = gl_frag(qrow, 1, fc);
      dof0.v = gl_frag(dorow, 0, fc); dof1.v = gl_frag(dorow, 1, fc);
      of0.v = gl_frag(orow, 0, fc); of1.v = gl_frag(orow, 1, fc);
    };
    load_tile(wave);
    const float lse_v = (threadIdx.x < TP && (int)threadIdx.x < T) ? lse[((size_t)b * H + h) * T + threadIdx.x] : 0.f;      // (requested with the panels)
    stage_rowmajor2<TP, NT>(P0, qb + ko, P1, qb + 2 * ko, ldi, T);
    if (threadIdx.x < TP) {
      lse_s[threadIdx.x] = ((int)threadIdx.x < T) ? lse_v * 1.4426950408889634f : 1.0e30f;
      del_s[threadIdx.x] = 0.f;
    }
    __syncthreads();
    GSL_ATTN_STAMP(1);
    for (int qt = wave; qt < nqt; qt += nwaves) {
      asm volatile("" ::: "memory");
      const int qr = qt * 16 + fr;
      if (qt != wave) load_tile(qt);
      float dl = 0.f;
      {
        const uint32_t a[8] = {dof0.u.x, dof0.u.y, dof0.u.z, dof0.u.w, dof1.u.x, dof1.u.y, dof1.u.z, dof1.u.w};
        const uint32_t c[8] = {of0.u.x, of0.u.y, of0.u.z, of0.u.w, of1.u.x, of1.u.y, of1.u.z, of1.u.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float a0, a1, c0, c1;
          unpack2o(a[i], a0, a1); unpack2o(c[i], c0, c1);
          dl += a0 * c0;
          dl += a1 * c1;
        }
      }
      dl += __shfl_xor(dl, 16, 64);
      dl += __shfl_xor(dl, 32, 64);
      const float lq2 = lse_s[min(qr, TP - 1)];
      Frag dsf[NKT / 2];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (FAST && kt == NKT - 1) {             // no valid key in this tile: dS = 0
          dsf[kt / 2].u.z = 0u; dsf[kt / 2].u.w = 0u;
          continue;
        }
        f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        sa = mfma16(lds_frag_rm(P0, kt * 16 + fr, 0, fc), qf0.v, sa);
        sa = mfma16(lds_frag_rm(P0, kt * 16 + fr, 1, fc), qf1.v, sa);
        dp = mfma16(lds_frag_rm(P1, kt * 16 + fr, 0, fc), dof0.v, dp);
        dp = mfma16(lds_frag_rm(P1, kt * 16 + fr, 1, fc), dof1.v, dp);
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(sa[r], c2, -lq2));
          if (FAST ? (kt == NKT - 2) : (kt >= ktf)) { if (kt * 16 + fc * 4 + r >= T) p = 0.f; }
          ds[r] = p * (dp[r] - dl);
        }
        if ((kt & 1) == 0) { dsf[kt / 2].u.x = pack2o(ds[0], ds[1]); dsf[kt / 2].u.y = pack2o(ds[2], ds[3]); }
        else { dsf[kt / 2].u.z = pack2o(ds[0], ds[1]); dsf[kt / 2].u.w = pack2o(ds[2], ds[3]); }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pr = 0; pr < NKT / 2; ++pr) acc = mfma16(lds_frag_trr(P0, dt, pr, lane), dsf[pr].v, acc);
        if (qr < T && !nostore) store4bf(dqkv + ((size_t)b * T + qr) * ld + h * HD + dt * 16 + fc * 4, acc, scale);
      }
      if (fc == 0 && qr < T) del_s[qr] = dl;
    }
  }
  // ------------------------------------------------------------------ phase B: dK / dV, one key tile per wave and round
  bf16x8_t kf[2], vf[2];
  int kr;
  auto load_keys = [&](int kp) {
    kr = kp * 16 + fr;
    const int krc = min(kr, T - 1);
    const bf16_t* krow = qb + (size_t)krc * ldi + ko;
    const bf16_t* vrow = qb + (size_t)krc * ldi + 2 * ko;
    kf[0] = gl_frag(krow, 0, fc); kf[1] = gl_frag(krow, 1, fc);
    vf[0] = gl_frag(vrow, 0, fc); vf[1] = gl_frag(vrow, 1, fc);
  };
  // The Q / dO panels of phase B. Every wave still holds the Q / dO fragments of the LAST query tile it processed in the operand layout
  // (lane (fr, fc): row fr, 16-byte chunk fc of each 32-column half) — exactly a ds_write_b128 into the row-major panel — so those
  // rows never come from global memory again. Only the FIRST tiles of the waves that ran two tiles (tiles 0 .. nqt - nwaves - 1: 5 of 13
  // at T = 197) are re-read: the waves without a second tile fetch them into registers while the others finish. Rows >= T of both
  // panels stay zero from the K / V staging. (Before: all 13 tiles of both panels were re-read, 412 MB per launch at B = 1024.)
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  constexpr int NPF = 12;
  const int first_early = max(nqt - nwaves, 0);                             // tiles that need the re-read = first tiles of two-tile waves
  const int prow = min(first_early * 16, T), tot = 2 * prow * 8;           // 16-byte chunks to re-read (Q rows, then dO rows)
  const int net = (nwaves - first_early) * 64;                               // threads of the waves without a second tile
  const bool pre = first_early < nwaves && net * NPF >= tot;                 // uniform (nqt <= 14, 8 waves: always true)
  const bool early = pre && wave >= first_early;
  const int last_qt = (wave + nwaves < nqt) ? wave + nwaves : wave;          // the tile whose fragments this wave holds now
  u32x4_t pf[NPF];
  if (early && tot > 0) {
    const int et = (wave - first_early) * 64 + lane;
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int c = min(et + k * net, tot - 1);
      const int pnl = c >= prow * 8, rc = c - pnl * (prow * 8);
      const bf16_t* src = pnl ? dob + (size_t)(rc >> 3) * ldo : qb + (size_t)(rc >> 3) * ldi;
      pf[k] = *reinterpret_cast<const u32x4_t*>(src + (rc & 7) * 8);
    }
  }
  // the K / V fragments of the wave's FIRST key tile come out of the K / V panels that are still in LDS (rows >= T are zero there; such
  // rows are never stored): 8 of the 13 tiles' second read of K and V never leaves the CU (-250 MB per launch at B = 1024)
  kr = min(wave * 16 + fr, TP - 1);      // (waves past the last key tile read a valid row and never use it)
  kf[0] = lds_frag_rm(P0, kr, 0, fc); kf[1] = lds_frag_rm(P0, kr, 1, fc);
  vf[0] = lds_frag_rm(P1, kr, 0, fc); vf[1] = lds_frag_rm(P1, kr, 1, fc);
  // (the second tile's fragments — waves 0 .. 4 at T = 197 — stay global loads: holding them across the panel re-staging costs 13
  //  spilled VGPRs under the 128-register cap and 60 us, measured)
  GSL_ATTN_STAMP(2);      // wave 0 done with its phase-A tiles
  __syncthreads();             // every wave is done with the K / V panels (and del_s is complete)
  GSL_ATTN_STAMP(3);
  if (pre) {
    if (early && tot > 0) {
      const int et = (wave - first_early) * 64 + lane;
#pragma unroll
      for (int k = 0; k < NPF; ++k) {
        const int c = et + k * net;
        if (c < tot) {
          const int pnl = c >= prow * 8, rc = c - pnl * (prow * 8);
          *reinterpret_cast<u32x4_t*>((pnl ? P1 : P0) + lds_off(rc >> 3, (rc & 7) * 8)) = pf[k];
        }
      }
    }
    const int dr = last_qt * 16 + fr;          // deposit the held tile (tiles >= nqt do not exist: waves past the last tile hold nothing)
    if (last_qt < nqt && dr < T) {
      *reinterpret_cast<uint4*>(P0 + lds_off(dr, fc * 8)) = qf0.u;
      *reinterpret_cast<uint4*>(P0 + lds_off(dr, 32 + fc * 8)) = qf1.u;
      *reinterpret_cast<uint4*>(P1 + lds_off(dr, fc * 8)) = dof0.u;
      *reinterpret_cast<uint4*>(P1 + lds_off(dr, 32 + fc * 8)) = dof1.u;
    }
  } else {
    stage_rowmajor<TP>(P0, qb, ldi, T);
    stage_rowmajor<TP>(P1, dob, ldo, T);
  }
  __syncthreads();
  GSL_ATTN_STAMP(4);
  for (int kp = wave; kp < nqt; kp += nwaves) {
    if (kp != wave) load_keys(kp);
    f32x4_t adk[4], adv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { adk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; adv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    // one pair of query tiles; FAST: the last pair's second tile has no valid query (P = dS = 0) and is a separate instance
    auto pair_step = [&](int qp, auto only_first) {
      constexpr bool ONE = decltype(only_first)::value;
      Frag pf, dsf;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * qp + half;
        if (ONE && half == 1) {
          pf.u.z = 0u; pf.u.w = 0u; dsf.u.z = 0u; dsf.u.w = 0u;
          continue;
        }
        const bf16x8_t q0 = lds_frag_rm(P0, qt * 16 + fr, 0, fc), q1 = lds_frag_rm(P0, qt * 16 + fr, 1, fc);
        const bf16x8_t g0 = lds_frag_rm(P1, qt * 16 + fr, 0, fc), g1 = lds_frag_rm(P1, qt * 16 + fr, 1, fc);
        const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[qt * 16 + fc * 4]);
        const float4 d4 = *reinterpret_cast<const float4*>(&del_s[qt * 16 + fc * 4]);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
        f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        sa = mfma16(q0, kf[0], sa);
        sa = mfma16(q1, kf[1], sa);
        dp = mfma16(g0, vf[0], dp);
        dp = mfma16(g1, vf[1], dp);
        float p[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = __builtin_amdgcn_exp2f(fmaf(sa[r], c2, -lv[r]));
          ds[r] = p[r] * (dp[r] - dv[r]);
        }
        if (half == 0) {
          pf.u.x = pack2o(p[0], p[1]); pf.u.y = pack2o(p[2], p[3]);
          dsf.u.x = pack2o(ds[0], ds[1]); dsf.u.y = pack2o(ds[2], ds[3]);
        } else {
          pf.u.z = pack2o(p[0], p[1]); pf.u.w = pack2o(p[2], p[3]);
          dsf.u.z = pack2o(ds[0], ds[1]); dsf.u.w = pack2o(ds[2], ds[3]);
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8_t ot = lds_frag_trr(P1, dt, qp, lane), qtf = lds_frag_trr(P0, dt, qp, lane);
        adv[dt] = mfma16(ot, pf.v, adv[dt]);
        adk[dt] = mfma16(qtf, dsf.v, adk[dt]);
      }
    };
#pragma unroll 1
    for (int qp = 0; qp < (FAST ? NKT / 2 - 1 : NKT / 2); ++qp) pair_step(qp, std::false_type{});
    if constexpr (FAST) pair_step(NKT / 2 - 1, std::true_type{});
    if (kr < T && !nostore) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16_t* base = dqkv + ((size_t)b * T + kr) * ld + h * HD + dt * 16 + fc * 4;
        store4bf(base + H * HD, adk[dt], scale);
        store4bf(base + 2 * H * HD, adv[dt], 1.0f);
      }
    }
    GSL_ATTN_STAMP(kp == wave ? 5 : 6);
  }
}

#undef GSL_ATTN_STAMP

// =====================================================================================
// backward, MERGED (16-bit, 192 < T <= 208, at least two items per CU; round 5): persistent, one 16-wave workgroup per CU.
// The fused kernel above computes the score tiles twice (S and dP in the dQ phase AND in the dK/dV phase: 7 matrix products and two
// exponentials per score), runs 13 tiles on 8 waves (two rounds, the second 5/8 full) and re-stages its panels between the phases.
// Here every score tile is computed ONCE, by the wave that owns its key tile:
//   * waves 0..12 own one key tile each (K / V fragments in registers, dK / dV accumulators in registers) and walk the query pairs:
//     S = Q K^T, dP = dO V^T, P, dS exactly as the dK/dV phase of the fused kernel — and park the 16-bit dS tile in an LDS buffer
//     Ds[key][query] (one ds_write_b64 per tile: the C layout holds 4 consecutive queries of one key);
//   * dQ^T += K^T dS^T then reads dS back with the LDS transpose read in the k-slot permutation of the dQ phase (the same MFMA operands
//     in the same order as the fused kernel), as 52 units (query tile, 16 columns of dQ) dealt round-robin to the 13 waves. Ds holds
//     128 queries, so the item runs as P1a (query pairs 0..3) | P2a (dQ tiles 0..7) | P1b (pairs 4..6) | P2b (dQ tiles 8..12): four
//     workgroup barriers per item. The K panel P2 needs is deposited by the key owners from their fragments;
//   * waves 13..15 move data one item ahead (registers, like the forward's loaders): dO and O during P1a, delta = rowsum(dO o O) and
//     the Q request during P2a, the Q / dO deposit into the panels during P2b (nobody reads them there). delta is accumulated in the
//     order of the fused kernel's lanes (chunk fc of the first 32 columns, then chunk fc of the second 32, then the xor-16 / xor-32
//     sums), so every value of the kernel is the fused kernel's: bit-identical dqkv.
// 5 matrix products instead of 7 (1 768 MFMAs per item instead of 2 548), one exponential per score instead of two, every byte of
// q / k / v / dO / o read from HBM once. LDS: three 208-row panels (KLD) + Ds [208][DSLD] + lse / delta (double-buffered) = 163 072 B.
// =====================================================================================
constexpr int DSLD = 144;      // elements per Ds row: 288 B, the 8 rows of a transpose read land on 8 disjoint bank octets
// lds_frag_trr for a panel with row stride LD; HALF: only rows pair*32 .. pair*32 + 15 exist (k-slots 4..7 are zero)
template <int LD, bool HALF>
__device__ __forceinline__ bf16x8_t lds_frag_trr_g(const bf16_t* base, int dt, int pair, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int row = pair * 32 + 4 * g + (i >> 2), col = dt * 16 + (i & 3) * 4;
  union { v4s_t h[2]; bf16x8_t v; } f;
  f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(base + row * LD + col));
  if constexpr (HALF) f.h[1] = v4s_t{0, 0, 0, 0};
  else f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_p)(base + (row + 16) * LD + col));
  return f.v;
}

template <int NKT>      // (NKT - 2) * 16 < T <= (NKT - 1) * 16: T = 197 with NKT = 14
__global__ __launch_bounds__(1024) void attn_bwd_merged_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                               const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                               bf16_t* __restrict__ dqkv, int T, int H, float scale, int nitems, int hm, int imap,
                                                               unsigned long long* __restrict__ stamps) {
  GSL_OP16_KERNEL_ENTRY();
  // development (GSL_ATTN_STAMPS = device address of 2048 u64): cycle stamps of the THIRD item of every 32nd workgroup, 16 slots per wave:
  // item start | P1a | barrier | P2a | barrier | P1b pairs | P1b last pair | stores + next keys issued | barrier | P2b | barrier
  unsigned long long* dbg = (stamps && (blockIdx.x & 31) == 0 && blockIdx.x < 256) ? stamps + (blockIdx.x >> 5) * 256 : nullptr;      // uniform
#define GSL_MSTAMP(i) do { if (dbg && seq == (int)(blockIdx.x + 2 * gridDim.x) && lane == 0) dbg[wave * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
  constexpr int NCW = NKT - 1;            // compute waves = key tiles that can hold a valid key (13)
  constexpr int TP = NCW * 16;            // panel rows (208)
  constexpr int NST = 9;                  // loader steps per panel: 24 rows x 8 chunks per step
  constexpr int NP = NKT / 2;             // query pairs (7); the last one holds a single tile
  constexpr int NPA = 4;                  // pairs of the first half: queries 0 .. 127 = dQ tiles 0 .. 7
  constexpr float LOG2E = 1.4426950408889634f;
  // ONE array in a fixed order: what a phase touches together sits within the 16-bit immediate offset of a DS instruction from one per-lane
  // base register (Q | dO | lse | delta for P1, Ds | K for P2). As separate arrays the linker's order put the panels > 64 KB apart and every
  // pair step re-derived its addresses with 20 v_add_u32 (of 56 VALU instructions); now 10.
  __shared__ __attribute__((aligned(16))) bf16_t smem[3 * TP * KLD + TP * DSLD + 8 * TP];
  bf16_t* const Qs = smem;                                           // [TP][KLD]
  bf16_t* const Gs = Qs + TP * KLD;                                  // dO
  float (*const lse_s)[TP] = reinterpret_cast<float (*)[TP]>(Gs + TP * KLD);      // [2][TP], log2 units; padded queries 1e30 -> p = 0
  float (*const del_s)[TP] = lse_s + 2;                              // [2][TP]
  bf16_t* const Ds = reinterpret_cast<bf16_t*>(del_s + 2);           // dS [key][query of the half], [TP][DSLD]
  bf16_t* const Ks = Ds + TP * DSLD;
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const long ld = 3L * H * HD, ldo = (long)H * HD;
  const long ldi = hm ? (long)HD : ld, ko = hm ? (long)T * HD : (long)H * HD;      // qkv INPUT: row stride, K-panel offset (V at 2 ko)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fr = lane & 15, fc = lane >> 4;
  // rows >= T of the panels are zero and the padded queries' lse / delta are 1e30 / 0 for every item: written once
  for (int idx = threadIdx.x; idx < (TP - T) * 8; idx += 1024) {
    const int t = T + (idx >> 3), c = idx & 7;
    const uint4 z = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(Qs + lds_off(t, c * 8)) = z;
    *reinterpret_cast<uint4*>(Gs + lds_off(t, c * 8)) = z;
    *reinterpret_cast<uint4*>(Ks + lds_off(t, c * 8)) = z;
  }
  if ((int)threadIdx.x < TP - T) {
    lse_s[0][T + threadIdx.x] = 1.0e30f; lse_s[1][T + threadIdx.x] = 1.0e30f;
    del_s[0][T + threadIdx.x] = 0.f; del_s[1][T + threadIdx.x] = 0.f;
  }
  if (wave >= NCW) {
    // ------------------------------------------------------------------ loader waves (see attn_fwd_bf16_pers_kernel for the register rules)
    const int li = (wave - NCW) * 64 + lane, col = (li & 7) * 8, row0 = li >> 3, ch = li & 7;
    u32x4_t rq[NST], rg[NST], ro[NST];
    // (row0 is laundered through an empty asm in every helper: the 9 + 9 loop-invariant per-step offsets would otherwise be hoisted out
    //  of the item loop and live next to 72 data registers — 13 spilled VGPRs)
    auto req_go = [&](int seq) {
      int r0 = row0; asm volatile("" : "+v"(r0));
      const int it = item_remap(seq, H, imap), b = it / H, h = it % H;
      const bf16_t* gb = d_o + (size_t)b * T * ldo + h * HD;
      const bf16_t* ob = o + (size_t)b * T * ldo + h * HD;
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const size_t g = (size_t)min(r0 + 24 * k, T - 1) * ldo + col;
        rg[k] = *reinterpret_cast<const u32x4_t*>(gb + g);
        ro[k] = *reinterpret_cast<const u32x4_t*>(ob + g);
      }
    };
    auto req_q = [&](int seq) {
      int r0 = row0; asm volatile("" : "+v"(r0));
      const int it = item_remap(seq, H, imap), b = it / H, h = it % H;
      const bf16_t* qb = qkv + (hm ? (size_t)it * 3 * T * HD : (size_t)b * T * ld + h * HD);
#pragma unroll
      for (int k = 0; k < NST; ++k) rq[k] = *reinterpret_cast<const u32x4_t*>(qb + (size_t)min(r0 + 24 * k, T - 1) * ldi + col);
    };
    // delta of the rows this lane's 8-lane group holds, in the fused kernel's order: lanes ch < 4 hold chunk fc = ch of the first 32
    // columns, lanes ch >= 4 chunk fc = ch - 4 of the second 32: the chain runs through the first, then through the second
    auto chain = [&](float dl, const u32x4_t a, const u32x4_t c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a0, a1, c0, c1;
        unpack2o(a[i], a0, a1); unpack2o(c[i], c0, c1);
        dl += a0 * c0;      // (the fused kernel's source form: with fp16 operands the compiler turns each pair into one v_dot2_f32_f16,
        dl += a1 * c1;      //  which rounds once — an explicit fmaf chain differs in the last bit of 0.02 % of the dS values)
      }
      return dl;
    };
    // lane exchanges inside the 8-lane row group as DPP moves (a __shfl is a ds_bpermute round trip through the LDS queue: 27 of them
    // per item made the loaders the critical path of the phase — 13 - 16 k cycles, profiles/r05_j_attn_merged.md)
    auto dpp = [&](float v, auto ctrl) {
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    auto put_delta_lse = [&](int seq, int pn) {
      int r0 = row0; asm volatile("" : "+v"(r0));
      __builtin_amdgcn_s_setprio(2);                    // ~600 instructions against the compute waves' ~5 000 per phase: do not queue behind them
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const float p1 = chain(0.f, rg[k], ro[k]);
        const float init = dpp(p1, std::integral_constant<int, 0x114>{});       // row_shr:4: lane ch takes lane ch - 4 (used by ch >= 4 only)
        float t = chain(init, rg[k], ro[k]);
        t += dpp(t, std::integral_constant<int, 0xB1>{});                       // quad_perm [1,0,3,2]: lane ^ 1
        t += dpp(t, std::integral_constant<int, 0x4E>{});                       // quad_perm [2,3,0,1]: lane ^ 2
        const int r = r0 + 24 * k;
        if (ch == 4 && r < T) del_s[pn][r] = t;
      }
      __builtin_amdgcn_s_setprio(0);
      const int it = item_remap(seq, H, imap);
      const float* lp = lse + (size_t)it * T;      // lse is [B][H][T] = [item][T]
      if (li < T) lse_s[pn][li] = lp[li] * LOG2E;
      if (li + 192 < T) lse_s[pn][li + 192] = lp[li + 192] * LOG2E;
    };
    auto deposit = [&]() {
      int r0 = row0; asm volatile("" : "+v"(r0));
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int r = min(r0 + 24 * k, T - 1);      // rows past T - 1 re-write row T - 1 with its own data
        *reinterpret_cast<u32x4_t*>(Qs + lds_off(r, col)) = rq[k];
        *reinterpret_cast<u32x4_t*>(Gs + lds_off(r, col)) = rg[k];
      }
    };
    int seq = blockIdx.x, p = 0;
    req_go(seq);
    put_delta_lse(seq, 0);
    asm volatile("" ::: "memory");
    req_q(seq);
    deposit();
    wg_barrier_lds();                                   // the first item's Q / dO panels, lse and delta are in LDS
    for (; seq < nitems; seq += gridDim.x, p ^= 1) {
      const int nxt = (seq + (int)gridDim.x < nitems) ? seq + (int)gridDim.x : seq;
      GSL_MSTAMP(0);
      req_go(nxt);                                      // P1a
      GSL_MSTAMP(1);
      wg_barrier_lds();
      GSL_MSTAMP(2);
      GSL_MSTAMP(3);
      wg_barrier_lds();                                 // P2a
      GSL_MSTAMP(4);
      put_delta_lse(nxt, p ^ 1);                        // P1b (the longest phase): delta, then the Q request once the o registers are dead
      GSL_MSTAMP(5);
      asm volatile("" ::: "memory");
      req_q(nxt);
      GSL_MSTAMP(6);
      GSL_MSTAMP(7);
      wg_barrier_lds();
      GSL_MSTAMP(8);
      deposit();                                        // P2b
      GSL_MSTAMP(9);
      wg_barrier_lds();
      GSL_MSTAMP(10);
    }
    return;
  }
  // -------------------------------------------------------------------- compute waves: key tile = wave
  const float c2 = scale * LOG2E;
  const int kr = wave * 16 + fr, krc = min(kr, T - 1);
  const bool kvalid = kr < T;
  Frag kf[2], vf[2];
  auto load_keys = [&](int seq) {
    const int it = item_remap(seq, H, imap), b = it / H, h = it % H;
    const bf16_t* qb = qkv + (hm ? (size_t)it * 3 * T * HD : (size_t)b * T * ld + h * HD);
    const bf16_t* krow = qb + (size_t)krc * ldi + ko;
    const bf16_t* vrow = qb + (size_t)krc * ldi + 2 * ko;
    kf[0].v = gl_frag(krow, 0, fc); kf[1].v = gl_frag(krow, 1, fc);
    vf[0].v = gl_frag(vrow, 0, fc); vf[1].v = gl_frag(vrow, 1, fc);
  };
  load_keys(blockIdx.x);
  wg_barrier_lds();
  int p = 0;
  for (int seq = blockIdx.x; seq < nitems; seq += gridDim.x, p ^= 1) {
    const int item = item_remap(seq, H, imap);
    const int b = item / H, h = item % H;
    if (kvalid) {                                       // the K panel of P2 (rows >= T stay zero)
      *reinterpret_cast<uint4*>(Ks + lds_off(kr, fc * 8)) = kf[0].u;
      *reinterpret_cast<uint4*>(Ks + lds_off(kr, 32 + fc * 8)) = kf[1].u;
    }
    f32x4_t adk[4], adv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { adk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; adv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    // one pair of query tiles (the dK/dV phase of the fused kernel + the dS deposit); qc0 = column of the pair's first query in Ds
    auto pair_step = [&](int qp, int qc0, auto only_first) {
      constexpr bool ONE = decltype(only_first)::value;
      Frag pf, dsf;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * qp + half;
        if (ONE && half == 1) {
          pf.u.z = 0u; pf.u.w = 0u; dsf.u.z = 0u; dsf.u.w = 0u;
          continue;
        }
        const bf16x8_t q0 = lds_frag_rm(Qs, qt * 16 + fr, 0, fc), q1 = lds_frag_rm(Qs, qt * 16 + fr, 1, fc);
        const bf16x8_t g0 = lds_frag_rm(Gs, qt * 16 + fr, 0, fc), g1 = lds_frag_rm(Gs, qt * 16 + fr, 1, fc);
        const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[p][qt * 16 + fc * 4]);
        const float4 d4 = *reinterpret_cast<const float4*>(&del_s[p][qt * 16 + fc * 4]);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
        f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        sa = mfma16(q0, kf[0].v, sa);
        sa = mfma16(q1, kf[1].v, sa);
        dp = mfma16(g0, vf[0].v, dp);
        dp = mfma16(g1, vf[1].v, dp);
        float pr[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = __builtin_amdgcn_exp2f(fmaf(sa[r], c2, -lv[r]));
          ds[r] = pr[r] * (dp[r] - dv[r]);
        }
        uint2 w;
        if (half == 0) {
          pf.u.x = pack2o(pr[0], pr[1]); pf.u.y = pack2o(pr[2], pr[3]);
          dsf.u.x = pack2o(ds[0], ds[1]); dsf.u.y = pack2o(ds[2], ds[3]);
          w = make_uint2(dsf.u.x, dsf.u.y);
        } else {
          pf.u.z = pack2o(pr[0], pr[1]); pf.u.w = pack2o(pr[2], pr[3]);
          dsf.u.z = pack2o(ds[0], ds[1]); dsf.u.w = pack2o(ds[2], ds[3]);
          w = make_uint2(dsf.u.z, dsf.u.w);
        }
        if (!kvalid) w = make_uint2(0u, 0u);            // a key past T contributes nothing to dQ (the fused kernel masks p there)
        *reinterpret_cast<uint2*>(Ds + kr * DSLD + qc0 + half * 16 + fc * 4) = w;
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8_t ot = lds_frag_trr_g<KLD, ONE>(Gs, dt, qp, lane), qtf = lds_frag_trr_g<KLD, ONE>(Qs, dt, qp, lane);
        adv[dt] = mfma16(ot, pf.v, adv[dt]);
        adk[dt] = mfma16(qtf, dsf.v, adk[dt]);
      }
    };
    // dQ of query tile qt (tile qtl of the half in Ds): the dQ phase of the fused kernel with dS read back from LDS; the tile leaves as
    // full rows through the wave-private staging area stg (row stride SLD)
    auto dq_tile = [&](int qt, int qtl, bf16_t* stg, auto sld) {
      f32x4_t acc[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < NP - 1; ++pr) {
        const bf16x8_t bq = lds_frag_trr_g<DSLD, false>(Ds, qtl, pr, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = mfma16(lds_frag_trr_g<KLD, false>(Ks, dt, pr, lane), bq, acc[dt]);
      }
      {
        const bf16x8_t bq = lds_frag_trr_g<DSLD, true>(Ds, qtl, NP - 1, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = mfma16(lds_frag_trr_g<KLD, true>(Ks, dt, NP - 1, lane), bq, acc[dt]);
      }
      store_tile_rows<decltype(sld)::value>(stg, acc, scale, dqkv + ((size_t)b * T + qt * 16) * ld + h * HD, ld, min(16, T - qt * 16), lane);
    };
    // ---- P1a
    GSL_MSTAMP(0);
#pragma unroll 1
    for (int qp = 0; qp < NPA; ++qp) pair_step(qp, qp * 32, std::false_type{});
    GSL_MSTAMP(1);
    wg_barrier_lds();
    GSL_MSTAMP(2);
    // ---- P2a
    if (wave < NPA * 2) dq_tile(wave, wave, Qs + lds_off(wave * 16, 0), std::integral_constant<int, KLD>{});      // (the Q panel is dead until the loaders' deposit in P2b)
    GSL_MSTAMP(3);
    wg_barrier_lds();
    GSL_MSTAMP(4);
    // ---- P1b
#pragma unroll 1
    for (int qp = NPA; qp < NP - 1; ++qp) pair_step(qp, (qp - NPA) * 32, std::false_type{});
    GSL_MSTAMP(5);
    pair_step(NP - 1, (NP - 1 - NPA) * 32, std::true_type{});
    GSL_MSTAMP(6);
    {       // dK / dV leave as full rows through columns 80 .. 143 of the wave's own Ds rows (dead since P2a; P2b reads columns 0 .. 79)
      bf16_t* stg = Ds + wave * 16 * DSLD + 80;
      bf16_t* gk = dqkv + ((size_t)b * T + wave * 16) * ld + h * HD + H * HD;
      const int nv = min(16, T - wave * 16);
      store_tile_rows<DSLD>(stg, adk, scale, gk, ld, nv, lane);
      store_tile_rows<DSLD>(stg, adv, 1.0f, gk + H * HD, ld, nv, lane);
    }
    load_keys((seq + (int)gridDim.x < nitems) ? seq + (int)gridDim.x : seq);      // the next item's K / V fragments land under P2b
    GSL_MSTAMP(7);
    wg_barrier_lds();
    GSL_MSTAMP(8);
    // ---- P2b
    if (wave < NCW - NPA * 2) dq_tile(NPA * 2 + wave, wave, Ds + wave * 16 * DSLD + 80, std::integral_constant<int, DSLD>{});      // (columns 80 .. 143 of Ds are unused in the second half)
    GSL_MSTAMP(9);
    wg_barrier_lds();
    GSL_MSTAMP(10);
  }
}
#undef GSL_MSTAMP

// =====================================================================================
// f32 parity kernels (matrix cores, round 4; the thread-per-row VALU kernels of rounds 1 - 3 are gone)
// =====================================================================================
// f32 forward on the matrix cores (round 4: the engines evaluate in f32 by default, and the thread-per-query kernel this replaces —
// q in registers, sequential fmaf over d, two passes over the keys — was 31 % of an f32 evaluation batch: 13.3 ms per layer at 2 560 images). v_mfma_f32_16x16x4_f32 is an exact-f32 k-ordered fmaf chain, so the scores
// S = q . k are BIT-IDENTICAL to a sequential fmaf over d ascending from 0 (what the thread-per-row kernels of rounds 1 - 3 computed), and so are the row maximum and every expf argument; the sums over the
// keys (l and the P V accumulation) run in a different — still exact f32 — order. Same structure as the bf16 kernels: scores transposed
// (S^T = K Q^T: a lane owns one query column), all key tiles of a query tile in registers, two-pass softmax, O^T = V^T P^T with the
// k-slot permutation slot g <-> key 4 g + r so that register r of a score tile IS the B operand of step r. K / V panels row-major in LDS
// with 68-float rows (both fragment reads conflict-free). 8 waves, one workgroup per (image, head).
constexpr int FLD = 68;
template <int TP>
__global__ __launch_bounds__(512, 2) void attn_fwd_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                                float* __restrict__ lse, int T, int H, float scale) {
  GSL_OP16_KERNEL_ENTRY();
  constexpr int NKT = TP / 16;
  __shared__ __attribute__((aligned(16))) float Ks[TP * FLD];
  __shared__ __attribute__((aligned(16))) float Vs[TP * FLD];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * HD;
  const float* qb = qkv + (size_t)b * T * ld + h * HD;
  {      // both panels, every load in flight before the first LDS store; rows >= T are zero
    constexpr int NIT = (TP * 16 + 511) / 512;
    f32x4_t kv[NIT], vv[NIT];      // (ext vectors: HIP's float4 struct behind a select ends up in scratch memory)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = threadIdx.x + it * 512, t = min(idx >> 4, T - 1), c = idx & 15;
      kv[it] = *reinterpret_cast<const f32x4_t*>(qb + (size_t)t * ld + H * HD + c * 4);
      vv[it] = *reinterpret_cast<const f32x4_t*>(qb + (size_t)t * ld + 2 * H * HD + c * 4);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = threadIdx.x + it * 512, t = idx >> 4, c = idx & 15;
      if (idx < TP * 16) {
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4_t*>(Ks + t * FLD + c * 4) = (t < T) ? kv[it] : z;
        *reinterpret_cast<f32x4_t*>(Vs + t * FLD + c * 4) = (t < T) ? vv[it] : z;
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int nqt = (T + 15) / 16;
  for (int qt = wave; qt < nqt; qt += 8) {
    const int qr = qt * 16 + fr, qrc = min(qr, T - 1);
    float q[16];
#pragma unroll
    for (int st = 0; st < 16; ++st) q[st] = qb[(size_t)qrc * ld + 4 * st + g];
    f32x4_t sc[NKT];
    float m = -3.0e38f;
    static_assert(NKT % 2 == 0, "key tiles are processed in pairs");
#pragma unroll
    for (int kt = 0; kt < NKT; kt += 2) {      // two key tiles = two independent accumulator chains (a dependent 16x16x4 MFMA needs 40 cycles, an independent one 32)
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float* kp = Ks + (kt * 16 + fr) * FLD + g;
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[4 * st], q[st], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[16 * FLD + 4 * st], q[st], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0[r] = (kt * 16 + 4 * g + r < T) ? acc0[r] * scale : -3.0e38f;      // (dot * scale, as the backward kernels recompute it)
        acc1[r] = (kt * 16 + 16 + 4 * g + r < T) ? acc1[r] * scale : -3.0e38f;
        m = fmaxf(m, fmaxf(acc0[r], acc1[r]));
      }
      sc[kt] = acc0; sc[kt + 1] = acc1;
      __builtin_amdgcn_sched_barrier(0);      // (keeps the 224 fragment reads of the unrolled key loop from being hoisted into one register-spilling batch)
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = (kt * 16 + 4 * g + r < T) ? expf(sc[kt][r] - m) : 0.f;
        sc[kt][r] = pv; l += pv;
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    f32x4_t oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)      // four independent accumulator chains
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(kt * 16 + 4 * g + r) * FLD + dt * 16 + fr], sc[kt][r], oacc[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (qr < T) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<float4*>(o + ((size_t)b * T + qr) * (H * HD) + h * HD + dt * 16 + 4 * g) =
            make_float4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
    }
    if (g == 0 && qr < T) lse[((size_t)b * H + h) * T + qr] = m + logf(l);
  }
}

// ---- f32 backward on the matrix cores (round 4; same operand layouts and k-slot permutation as attn_fwd_f32_mfma_kernel). The thread-per-row
// kernels of this section took 22 ms per layer at 1 024 images — 39 % of an f32 training step; here a layer is two launches:
//   dQ:   waves own query tiles; K / V panels in LDS. S^T = K Q^T and dP^T = V dO^T (16 + 16 MFMAs per key tile), p = expf(S scale - lse),
//         dS = p (dP - delta) scale in C layout = the B operand of dQ^T += K^T dS^T (slot g <-> key 4 g + r); delta = rowsum(dO o) goes to
//         the workspace the dK / dV launch reads.
//   dK/dV: waves own key tiles; Q / dO panels in LDS. S = Q K^T and dP = dO V^T with the roles swapped (a lane owns a key column and four
//         queries), then dV^T += dO^T P and dK^T += Q^T dS over all query tiles.
// Scores and dP are k-ordered fmaf chains over d ascending; sums over keys / queries run in slot order (exact f32).
template <int TP>
__device__ __forceinline__ void stage_f32_pad2(float* d0, const float* s0, long ld0, float* d1, const float* s1, long ld1, int T) {
  constexpr int NIT = (TP * 16 + 511) / 512;      // 512-thread workgroups; rows of FLD floats; rows >= T are zero
  f32x4_t a[NIT], c[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = threadIdx.x + it * 512, t = min(idx >> 4, T - 1), ch = idx & 15;
    a[it] = *reinterpret_cast<const f32x4_t*>(s0 + (size_t)t * ld0 + ch * 4);
    c[it] = *reinterpret_cast<const f32x4_t*>(s1 + (size_t)t * ld1 + ch * 4);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = threadIdx.x + it * 512, t = idx >> 4, ch = idx & 15;
    if (idx < TP * 16) {
      const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4_t*>(d0 + t * FLD + ch * 4) = (t < T) ? a[it] : z;
      *reinterpret_cast<f32x4_t*>(d1 + t * FLD + ch * 4) = (t < T) ? c[it] : z;
    }
  }
}

template <int TP>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_f32_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                                     const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                     float* __restrict__ dqkv, float* __restrict__ delta, int T, int H, float scale) {
  GSL_OP16_KERNEL_ENTRY();
  constexpr int NKT = TP / 16;
  __shared__ __attribute__((aligned(16))) float Ks[TP * FLD];
  __shared__ __attribute__((aligned(16))) float Vs[TP * FLD];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * HD, ldo = (long)H * HD;
  const float* qb = qkv + (size_t)b * T * ld + h * HD;
  stage_f32_pad2<TP>(Ks, qb + H * HD, ld, Vs, qb + 2 * H * HD, ld, T);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int nqt = (T + 15) / 16;
  for (int qt = wave; qt < nqt; qt += 8) {
    const int qr = qt * 16 + fr, qrc = min(qr, T - 1);
    const float* gr = d_o + ((size_t)b * T + qrc) * ldo + h * HD;
    const float* orow = o + ((size_t)b * T + qrc) * ldo + h * HD;
    float q[16], gq[16];
    float dl = 0.f;
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      q[st] = qb[(size_t)qrc * ld + 4 * st + g];
      gq[st] = gr[4 * st + g];
      dl = fmaf(gq[st], orow[4 * st + g], dl);
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const float lq = lse[((size_t)b * H + h) * T + qrc];
    f32x4_t dsr[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      const float* kp = Ks + (kt * 16 + fr) * FLD + g;
      const float* vp = Vs + (kt * 16 + fr) * FLD + g;
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        sa = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[4 * st], q[st], sa, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[4 * st], gq[st], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = (kt * 16 + 4 * g + r < T) ? expf(sa[r] * scale - lq) : 0.f;
        dsr[kt][r] = pv * (dp[r] - dl) * scale;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4_t dqa[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dqa[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dqa[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(kt * 16 + 4 * g + r) * FLD + dt * 16 + fr], dsr[kt][r], dqa[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (qr < T) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(dqkv + ((size_t)b * T + qr) * ld + h * HD + dt * 16 + 4 * g) = dqa[dt];
    }
    if (g == 0 && qr < T) delta[((size_t)b * H + h) * T + qr] = dl;
  }
}

template <int TP>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_f32_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                      const float* __restrict__ lse, const float* __restrict__ delta,
                                                                      float* __restrict__ dqkv, int T, int H, float scale) {
  GSL_OP16_KERNEL_ENTRY();
  constexpr int NQT = TP / 16;
  __shared__ __attribute__((aligned(16))) float Qs[TP * FLD];
  __shared__ __attribute__((aligned(16))) float Gs[TP * FLD];   // dO
  __shared__ __attribute__((aligned(16))) float lse_s[TP];
  __shared__ __attribute__((aligned(16))) float del_s[TP];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * HD, ldo = (long)H * HD;
  const float* qb = qkv + (size_t)b * T * ld + h * HD;
  stage_f32_pad2<TP>(Qs, qb, ld, Gs, d_o + (size_t)b * T * ldo + h * HD, ldo, T);
  if (threadIdx.x < TP) {
    const bool in = (int)threadIdx.x < T;
    lse_s[threadIdx.x] = in ? lse[((size_t)b * H + h) * T + threadIdx.x] : 0.f;
    del_s[threadIdx.x] = in ? delta[((size_t)b * H + h) * T + threadIdx.x] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int nkt = (T + 15) / 16;
  for (int kp = wave; kp < nkt; kp += 8) {
    const int kr = kp * 16 + fr, krc = min(kr, T - 1);
    float k[16], v[16];
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      k[st] = qb[(size_t)krc * ld + H * HD + 4 * st + g];
      v[st] = qb[(size_t)krc * ld + 2 * H * HD + 4 * st + g];
    }
    f32x4_t adk[4], adv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { adk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; adv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 2
    for (int qt = 0; qt < NQT; ++qt) {
      f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      const float* qp = Qs + (qt * 16 + fr) * FLD + g;
      const float* gp = Gs + (qt * 16 + fr) * FLD + g;
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        sa = __builtin_amdgcn_mfma_f32_16x16x4f32(qp[4 * st], k[st], sa, 0, 0, 0);      // D[query 4 g + r][key fr]
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gp[4 * st], v[st], dp, 0, 0, 0);
      }
      const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(&lse_s[qt * 16 + 4 * g]);
      const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(&del_s[qt * 16 + 4 * g]);
      float pv[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[r] = (qt * 16 + 4 * g + r < T) ? expf(sa[r] * scale - l4[r]) : 0.f;
        ds[r] = pv[r] * (dp[r] - d4[r]) * scale;
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (qt * 16 + 4 * g + r) * FLD + dt * 16 + fr;
          adv[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs[row], pv[r], adv[dt], 0, 0, 0);
          adk[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[row], ds[r], adk[dt], 0, 0, 0);
        }
    }
    if (kr < T) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        float* base = dqkv + ((size_t)b * T + kr) * ld + h * HD + dt * 16 + 4 * g;
        *reinterpret_cast<f32x4_t*>(base + H * HD) = adk[dt];
        *reinterpret_cast<f32x4_t*>(base + 2 * H * HD) = adv[dt];
      }
    }
  }
}

// =====================================================================================
// backward for a cls-only output gradient (last block): rank-1 structure, one thread per key
//   p_j = exp(q0.k_j*scale - lse0), D = dO0.O0, ds_j = p_j (dO0.v_j - D) scale
//   dq0 = sum_j ds_j k_j ; dk_j = ds_j q0 ; dv_j = p_j dO0 ; dq_t = 0 for t > 0
// =====================================================================================
// Row-cooperative form: 8 lanes share one key row (8 head dims each: 16-byte bf16 / 2 x 16-byte f32 accesses), so a wave instruction
// touches 8 complete 128-byte K / V / dK / dV rows instead of 64 different ones (the thread-per-key form moved 1.03 GB at 2.4 TB/s).
// The two dot products are finished with three xor-shuffles inside the 8-lane group; dq0 is reduced in a fixed order.
// Operand addressing of the two cls kernels. qkv_layout 0: token-major qkv [B*T, 3*H*64]; 1: head-major [B][H][3][T][64]; 2 (round 3):
// the last block's projection computes Q for the cls rows only — kv token-major [B*T, 2*H*64] (k | v) + q_cls [B, H*64].
struct ClsAddr {
  long ldi, ko, vo;      // row stride of the K / V rows of one (image, head), offsets of the K and V panels from `base`
};
template <typename T>
__device__ __forceinline__ const T* cls_base(const T* qkv, int b, int h, int Tn, int H, int layout, ClsAddr& a) {
  const long ld3 = 3L * H * HD, ld2 = 2L * H * HD;
  if (layout == 1) { a.ldi = HD; a.ko = (long)Tn * HD; a.vo = 2L * Tn * HD; return qkv + (size_t)(b * H + h) * 3 * Tn * HD; }
  if (layout == 2) { a.ldi = ld2; a.ko = 0; a.vo = (long)H * HD; return qkv + (size_t)b * Tn * ld2 + h * HD; }
  a.ldi = ld3; a.ko = (long)H * HD; a.vo = 2L * H * HD;
  return qkv + (size_t)b * Tn * ld3 + h * HD;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_cls_kernel(const T* __restrict__ qkv, const T* __restrict__ q_cls, const T* __restrict__ o,
                                                           const T* __restrict__ d_o_cls, const float* __restrict__ lse,
                                                           T* __restrict__ dqkv, T* __restrict__ dq_cls, int Tn, int H, float scale, int hm,
                                                           int cls_compact) {
  GSL_OP16_KERNEL_ENTRY();
  __shared__ float q0[HD], g0[HD], red[32][HD];
  __shared__ float sD;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long ldo = (long)H * HD;
  ClsAddr ad;
  const T* qb = cls_base(qkv, b, h, Tn, H, hm, ad);
  // outputs: layouts 0 / 1 -> dqkv token-major [B*T, 3*H*64] (dQ rows of the other tokens zero-filled); layout 2 -> dkv [B*T, 2*H*64] + dq_cls
  const long ldd = (hm == 2 ? 2L : 3L) * H * HD, dko = (hm == 2) ? 0 : (long)H * HD, dvo = dko + (long)H * HD;
  T* db = dqkv + (size_t)b * Tn * ldd + h * HD;
  const int tid = threadIdx.x;
  if (tid < HD) {
    q0[tid] = Elem<T>::ld(hm == 2 ? q_cls + (size_t)b * ldo + h * HD + tid : qb + tid);
    g0[tid] = Elem<T>::ld(d_o_cls + (size_t)b * ldo + h * HD + tid);
    float v = g0[tid] * Elem<T>::ld(o + (size_t)b * (cls_compact ? 1 : Tn) * ldo + h * HD + tid);
    v = wave_sum(v);
    if (tid == 0) sD = v;
  }
  __syncthreads();
  const float D = sD, l0 = lse[((size_t)b * H + h) * (cls_compact ? 1 : Tn)];
  const int grp = tid >> 3, sub = tid & 7;      // 32 row groups x 8 lanes; lane `sub` owns head dims sub*8 .. sub*8+7
  float qs[8], gs[8], dq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { qs[i] = q0[sub * 8 + i]; gs[i] = g0[sub * 8 + i]; dq[i] = 0.f; }
  for (int j0 = 0; j0 < Tn; j0 += 32) {
    const int j = j0 + grp;
    const bool live = j < Tn;
    const int jc = live ? j : Tn - 1;
    const T* kr = qb + (size_t)jc * ad.ldi + ad.ko + sub * 8;
    const T* vr = qb + (size_t)jc * ad.ldi + ad.vo + sub * 8;
    float kv[8], vv[8];
    Elem<T>::ld4(kr, kv); Elem<T>::ld4(kr + 4, kv + 4);
    Elem<T>::ld4(vr, vv); Elem<T>::ld4(vr + 4, vv + 4);
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s = fmaf(qs[i], kv[i], s); dp = fmaf(gs[i], vv[i], dp); }
#pragma unroll
    for (int sh = 1; sh < 8; sh <<= 1) { s += __shfl_xor(s, sh, 64); dp += __shfl_xor(dp, sh, 64); }
    const float p = expf(s * scale - l0);
    const float ds = p * (dp - D) * scale;
    if (live) {
      float a[8], bb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = ds * qs[i]; bb[i] = p * gs[i]; dq[i] = fmaf(ds, kv[i], dq[i]); }
      T* dk = db + (size_t)j * ldd + dko + sub * 8;
      T* dv = db + (size_t)j * ldd + dvo + sub * 8;
      Elem<T>::st4(dk, a); Elem<T>::st4(dk + 4, a + 4);
      Elem<T>::st4(dv, bb); Elem<T>::st4(dv + 4, bb + 4);
      if (hm != 2 && j > 0) {   // dQ of the non-cls tokens is exactly zero
        const float z[4] = {0.f, 0.f, 0.f, 0.f};
        Elem<T>::st4(db + (size_t)j * ldd + sub * 8, z); Elem<T>::st4(db + (size_t)j * ldd + sub * 8 + 4, z);
      }
    }
  }
  // dq0[d] = sum over the 32 row groups, fixed order
#pragma unroll
  for (int i = 0; i < 8; ++i) red[grp][sub * 8 + i] = dq[i];
  __syncthreads();
  if (tid < HD) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) t += red[g][tid];
    Elem<T>::st(hm == 2 ? dq_cls + (size_t)b * ldo + h * HD + tid : db + tid, t);
  }
}

// Forward of the same case: in the LAST block only the cls query's output is ever consumed (the head pools x[:, 0] and everything after
// the attention is token-wise), so its attention is one query row per (image, head) against the full K / V panels — a streaming
// softmax-weighted sum, HBM-bound on the K / V rows (8 lanes per row as above). o_cls [B, H*64], lse_cls [B, H].
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_cls_kernel(const T* __restrict__ qkv, const T* __restrict__ q_cls, T* __restrict__ o_cls,
                                                           float* __restrict__ lse_cls, int Tn, int H, float scale, int hm) {
  GSL_OP16_KERNEL_ENTRY();
  __shared__ float q0[HD], sc[256], red[32][HD];
  __shared__ float sm[16];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long ldo = (long)H * HD;
  ClsAddr ad;
  const T* qb = cls_base(qkv, b, h, Tn, H, hm, ad);
  const int tid = threadIdx.x;
  if (tid < HD) q0[tid] = Elem<T>::ld(hm == 2 ? q_cls + (size_t)b * ldo + h * HD + tid : qb + tid);
  __syncthreads();
  const int grp = tid >> 3, sub = tid & 7;
  float qs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) qs[i] = q0[sub * 8 + i];
  for (int j0 = 0; j0 < Tn; j0 += 32) {
    const int j = j0 + grp;
    const int jc = j < Tn ? j : Tn - 1;
    const T* kr = qb + (size_t)jc * ad.ldi + ad.ko + sub * 8;
    float kv[8];
    Elem<T>::ld4(kr, kv); Elem<T>::ld4(kr + 4, kv + 4);
    float sdot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sdot = fmaf(qs[i], kv[i], sdot);
#pragma unroll
    for (int sh = 1; sh < 8; sh <<= 1) sdot += __shfl_xor(sdot, sh, 64);
    if (j < Tn && sub == 0) sc[j] = sdot * scale;
  }
  __syncthreads();
  float m = -3.0e38f;
  for (int j = tid; j < Tn; j += 256) m = fmaxf(m, sc[j]);
  m = block_max(m, sm);
  float e = 0.f;
  for (int j = tid; j < Tn; j += 256) { const float p = expf(sc[j] - m); sc[j] = p; e += p; }
  e = block_sum(e, sm);        // (its barriers also publish the p values)
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int j0 = 0; j0 < Tn; j0 += 32) {
    const int j = j0 + grp;
    if (j < Tn) {
      const T* vr = qb + (size_t)j * ad.ldi + ad.vo + sub * 8;
      float vv[8];
      Elem<T>::ld4(vr, vv); Elem<T>::ld4(vr + 4, vv + 4);
      const float p = sc[j];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(p, vv[i], acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[grp][sub * 8 + i] = acc[i];
  __syncthreads();
  if (tid < HD) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) t += red[g][tid];       // fixed order
    Elem<T>::st(o_cls + (size_t)b * ldo + h * HD + tid, t / e);
  }
  if (tid == 0) lse_cls[(size_t)b * H + h] = m + logf(e);
}

extern "C" int GSL_ENTRY(gsl_attention_fwd_cls)(const void* qkv, const void* q_cls, void* o_cls, float* lse_cls, int B, int T, int H, float scale,
                                     int dtype, int qkv_layout, gsl_stream_t s) {
  GSL_FORWARD_H16(dtype, h16_gsl_attention_fwd_cls(qkv, q_cls, o_cls, lse_cls, B, T, H, scale, dtype, qkv_layout, s));
  GSL_CHECK_ARG(qkv && o_cls && lse_cls && B > 0 && T > 1 && T <= 256 && H > 0, "null/size (T <= 256)");
  GSL_CHECK_ARG(qkv_layout >= 0 && qkv_layout <= 2 && (qkv_layout != 2 || q_cls), "qkv_layout: 0 token-major, 1 head-major, 2 kv + q_cls");
  const dim3 grid(B * H), blk(256);
  if (dtype == GSL_OP16)
    hipLaunchKernelGGL(attn_fwd_cls_kernel<bf16_t>, grid, blk, 0, as_stream(s), (const bf16_t*)qkv, (const bf16_t*)q_cls, (bf16_t*)o_cls, lse_cls, T, H, scale, qkv_layout);
#if GSL_HAS_F32
  else if (dtype == GSL_F32)
    hipLaunchKernelGGL(attn_fwd_cls_kernel<float>, grid, blk, 0, as_stream(s), (const float*)qkv, (const float*)q_cls, (float*)o_cls, lse_cls, T, H, scale, qkv_layout);
#endif
  else return fail(GSL_ERR_ARG, "gsl_attention_fwd_cls: bad dtype%s %ld", "", dtype);
  return check_launch("gsl_attention_fwd_cls");
}

extern "C" int GSL_ENTRY(gsl_attention_bwd_cls)(const void* qkv, const void* q_cls, const void* o, const void* d_o_cls, const float* lse, void* dqkv,
                                     void* dq_cls, int B, int T, int H, float scale, int dtype, int qkv_layout, int cls_compact,
                                     gsl_stream_t s) {
  GSL_FORWARD_H16(dtype, h16_gsl_attention_bwd_cls(qkv, q_cls, o, d_o_cls, lse, dqkv, dq_cls, B, T, H, scale, dtype, qkv_layout, cls_compact, s));
  GSL_CHECK_ARG(qkv && o && d_o_cls && lse && dqkv && B > 0 && T > 1 && H > 0, "null/size");
  GSL_CHECK_ARG(qkv_layout >= 0 && qkv_layout <= 2 && (qkv_layout != 2 || (q_cls && dq_cls)), "qkv_layout: 0 token-major, 1 head-major, 2 kv + q_cls / dq_cls");
  const dim3 grid(B * H), blk(256);
  if (dtype == GSL_OP16)
    hipLaunchKernelGGL(attn_bwd_cls_kernel<bf16_t>, grid, blk, 0, as_stream(s), (const bf16_t*)qkv, (const bf16_t*)q_cls, (const bf16_t*)o,
                       (const bf16_t*)d_o_cls, lse, (bf16_t*)dqkv, (bf16_t*)dq_cls, T, H, scale, qkv_layout, cls_compact);
#if GSL_HAS_F32
  else if (dtype == GSL_F32)
    hipLaunchKernelGGL(attn_bwd_cls_kernel<float>, grid, blk, 0, as_stream(s), (const float*)qkv, (const float*)q_cls, (const float*)o,
                       (const float*)d_o_cls, lse, (float*)dqkv, (float*)dq_cls, T, H, scale, qkv_layout, cls_compact);
#endif
  else return fail(GSL_ERR_ARG, "gsl_attention_bwd_cls: bad dtype%s %ld", "", dtype);
  return check_launch("gsl_attention_bwd_cls");
}

// Development knobs (libgslora_hip_dev.so only; the product library reads nothing from the environment):
//   GSL_ATTN_PERSISTENT=0 one item per workgroup; GSL_ATTN_ABL=1 staging only, 2 no staging; GSL_ATTN_BWD_SPLIT=1 the two-kernel backward;
//   GSL_ATTN_NT key tiles per wave of the dK/dV kernel; GSL_ATTN_STAMPS device address of a cycle-stamp buffer;
//   GSL_ATTN_BWD_MERGED=0 the fused two-phase backward instead of the merged one, GSL_ATTN_BWD_MERGED_MIN items per CU from which the merged one runs.
#ifdef GSL_DEV
static inline int attn_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline unsigned long long* attn_stamps() { const char* sp = getenv("GSL_ATTN_STAMPS"); return sp ? reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 0)) : nullptr; }
#else
static inline constexpr int attn_env(const char*, int dflt) { return dflt; }
static inline constexpr unsigned long long* attn_stamps() { return nullptr; }
#endif
static inline int attn_persistent() { return attn_env("GSL_ATTN_PERSISTENT", 1); }
static inline int attn_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount;
    else n = 256;
  }
  return n;
}
static inline int attn_abl() { return attn_env("GSL_ATTN_ABL", 0); }

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" int GSL_ENTRY(gsl_attention_fwd)(const void* qkv, void* o, float* lse, int B, int T, int H, float scale, int dtype,
                                 int qkv_layout, gsl_stream_t s) {
  GSL_FORWARD_H16(dtype, h16_gsl_attention_fwd(qkv, o, lse, B, T, H, scale, dtype, qkv_layout, s));
  GSL_CHECK_ARG(qkv && o && lse && B > 0 && T > 1 && H > 0, "null/size");
  GSL_CHECK_ARG(qkv_layout == 0 || (qkv_layout == 1 && dtype == GSL_OP16), "qkv_layout: 0 token-major, 1 head-major (bf16 kernels only)");
  const int hm = qkv_layout;
  GSL_CHECK_ARG(T <= 224, "T <= 224 tokens (single-panel attention)");
  hipStream_t st = as_stream(s);
  const dim3 grid(B * H), blk(256);
  // (item_remap for the forward: measured +1 % — 212 -> 215 us at B = 1024 —, so the plain order stays; the backward gains 2.3 %: profiles/r04_notes.md)
  const int imap = (B % 8 == 0 && attn_num_cus() % 8 == 0) ? attn_env("GSL_ATTN_ITEM_REMAP_FWD", 0) : 0;
  if (dtype == GSL_OP16) {
    if (T <= 64) hipLaunchKernelGGL(attn_fwd_bf16_kernel<4>, grid, blk, 0, st, (const bf16_t*)qkv, (bf16_t*)o, lse, T, H, scale, attn_abl(), hm);
    else if (attn_persistent() && T <= 208 && B * H >= 2 * attn_num_cus())
    {
      if (T > 192) hipLaunchKernelGGL((attn_fwd_bf16_pers_kernel<14, true>), dim3(attn_num_cus()), dim3(1024), 0, st, (const bf16_t*)qkv, (bf16_t*)o, lse, T, H, scale, B * H, hm, imap);
      else hipLaunchKernelGGL((attn_fwd_bf16_pers_kernel<14, false>), dim3(attn_num_cus()), dim3(1024), 0, st, (const bf16_t*)qkv, (bf16_t*)o, lse, T, H, scale, B * H, hm, imap);
    }
    else if (B * H < attn_num_cus()) hipLaunchKernelGGL((attn_fwd_bf16_kernel<14, 1024>), grid, dim3(1024), 0, st, (const bf16_t*)qkv, (bf16_t*)o, lse, T, H, scale, attn_abl(), hm);
    else hipLaunchKernelGGL(attn_fwd_bf16_kernel<14>, grid, dim3(512), 0, st, (const bf16_t*)qkv, (bf16_t*)o, lse, T, H, scale, attn_abl(), hm);
  }
#if GSL_HAS_F32
  else if (dtype == GSL_F32) {
    if (T <= 64) hipLaunchKernelGGL(attn_fwd_f32_mfma_kernel<64>, grid, dim3(512), 0, st, (const float*)qkv, (float*)o, lse, T, H, scale);
    else hipLaunchKernelGGL(attn_fwd_f32_mfma_kernel<224>, grid, dim3(512), 0, st, (const float*)qkv, (float*)o, lse, T, H, scale);
  }
#endif
  else return fail(GSL_ERR_ARG, "gsl_attention_fwd: bad dtype%s %ld", "", dtype);
  return check_launch("gsl_attention_fwd");
}

extern "C" int GSL_ENTRY(gsl_attention_bwd)(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv,
                                 float* delta_ws, int B, int T, int H, float scale, int dtype, int qkv_layout, gsl_stream_t s) {
  GSL_FORWARD_H16(dtype, h16_gsl_attention_bwd(qkv, o, d_o, lse, dqkv, delta_ws, B, T, H, scale, dtype, qkv_layout, s));
  GSL_CHECK_ARG(qkv && o && d_o && lse && dqkv && delta_ws && B > 0 && T > 1 && H > 0, "null/size");
  GSL_CHECK_ARG(qkv_layout == 0 || (qkv_layout == 1 && dtype == GSL_OP16), "qkv_layout: 0 token-major, 1 head-major (bf16 kernels only)");
  const int hm = qkv_layout;
  GSL_CHECK_ARG(T <= 224, "T <= 224 tokens (single-panel attention)");
  hipStream_t st = as_stream(s);
  const dim3 grid(B * H), blk(256);
  if (dtype == GSL_OP16) {
    const bf16_t* q = (const bf16_t*)qkv; const bf16_t* oo = (const bf16_t*)o; const bf16_t* g = (const bf16_t*)d_o;
    bf16_t* dq = (bf16_t*)dqkv;
    const int imap = (B % 8 == 0) ? attn_env("GSL_ATTN_ITEM_REMAP", 1) : 0;      // heads of an image on one XCD (item_remap)
    if (T <= 64) {
      hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<4>, grid, blk, 0, st, q, oo, g, lse, dq, delta_ws, T, H, scale, attn_abl(), hm);
      hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<4, 2>), grid, blk, 0, st, q, g, lse, delta_ws, dq, T, H, scale, attn_abl(), hm);
    } else if (attn_env("GSL_ATTN_BWD_SPLIT", 0) == 0) {
      // (a persistent wave-specialised form like the forward's was measured slower here: 795 vs 736 us at B = 1024 — its four
      //  workgroup-wide barriers per item cost more than the hidden staging saves; profiles/r01_gemm_ab.md)
      unsigned long long* stp = attn_stamps();
      // round 5: every score tile computed once (attn_bwd_merged_kernel; bit-identical to the fused kernel, GSL_ATTN_BWD_MERGED=0 in the dev build)
      // (from 8 items per CU: a persistent workgroup pays its prologue and a ragged last round — at 4.5 items per CU, ViT-B/16 48 + 48 x 12
      //  heads, the fused kernel is 4 % faster per launch: profiles/r05_notes.md)
      if (T > 192 && T <= 208 && B * H >= attn_env("GSL_ATTN_BWD_MERGED_MIN", 8) * attn_num_cus() && attn_env("GSL_ATTN_BWD_MERGED", 1))
        hipLaunchKernelGGL((attn_bwd_merged_kernel<14>), dim3(attn_num_cus()), dim3(1024), 0, st, q, oo, g, lse, dq, T, H, scale, B * H, hm, (attn_num_cus() % 8 == 0) ? imap : 0, stp);
      else
      if (B * H < attn_num_cus()) {      // fewer items than CUs: sixteen waves per item
        if (T > 192 && T <= 208) hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<14, true, 1024>), grid, dim3(1024), 0, st, q, oo, g, lse, dq, T, H, scale, stp, hm, imap);
        else hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<14, false, 1024>), grid, dim3(1024), 0, st, q, oo, g, lse, dq, T, H, scale, stp, hm, imap);
      }
      else if (T > 192 && T <= 208) hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<14, true>), grid, dim3(512), 0, st, q, oo, g, lse, dq, T, H, scale, stp, hm | (attn_abl() == 4 ? 2 : 0), imap);
      else hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<14, false>), grid, dim3(512), 0, st, q, oo, g, lse, dq, T, H, scale, stp, hm, imap);
    } else {        // development knob GSL_ATTN_BWD_SPLIT=1: the two-kernel form
      hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<14>, grid, dim3(512), 0, st, q, oo, g, lse, dq, delta_ws, T, H, scale, attn_abl(), hm);
      {       // measured at B = 1024, T = 197: NT = 1 (two workgroups per CU) 410 us, NT = 2 480 us
        if (attn_env("GSL_ATTN_NT", 1) == 1) hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<14, 1>), grid, dim3(512), 0, st, q, g, lse, delta_ws, dq, T, H, scale, attn_abl(), hm);
        else hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<14, 2>), grid, dim3(512), 0, st, q, g, lse, delta_ws, dq, T, H, scale, attn_abl(), hm); }
    }
  }
#if GSL_HAS_F32
  else if (dtype == GSL_F32) {
    const float* q = (const float*)qkv; const float* oo = (const float*)o; const float* g = (const float*)d_o;
    float* dq = (float*)dqkv;
    if (T <= 64) {
      hipLaunchKernelGGL(attn_bwd_dq_f32_mfma_kernel<64>, grid, dim3(512), 0, st, q, oo, g, lse, dq, delta_ws, T, H, scale);
      hipLaunchKernelGGL(attn_bwd_dkv_f32_mfma_kernel<64>, grid, dim3(512), 0, st, q, g, lse, delta_ws, dq, T, H, scale);
    } else {
      hipLaunchKernelGGL(attn_bwd_dq_f32_mfma_kernel<224>, grid, dim3(512), 0, st, q, oo, g, lse, dq, delta_ws, T, H, scale);
      hipLaunchKernelGGL(attn_bwd_dkv_f32_mfma_kernel<224>, grid, dim3(512), 0, st, q, g, lse, delta_ws, dq, T, H, scale);
    }
  }
#endif
  else return fail(GSL_ERR_ARG, "gsl_attention_bwd: bad dtype%s %ld", "", dtype);
  return check_launch("gsl_attention_bwd");
}
GSL_OPNS_END
