// gemm.hip — dense NT GEMM  acc = alpha * (A1·W1ᵀ + A2·W2ᵀ)  with fused epilogues.
//
// Replaces F.linear / loralib.Linear.forward (reference vit_pytorch_face/vit_face.py:330-334,
// 349-356, 531) and their autograd dX. The second K segment carries the rank-r LoRA term:
// [x | s·(x Aᵀ)] · [W | B]ᵀ  ==  x Wᵀ + s·(x Aᵀ) Bᵀ, so the adapter costs one extra K tile on the
// matrix cores instead of two skinny GEMMs and an elementwise add.
//
// bf16 path (v_mfma_f32_16x16x32_bf16, f32 accumulate). Production tiles: 256x256x64 with the 8-phase ping-pong schedule
//   (gemm_bf16_p8_kernel, N >= 512; also the in-kernel-LoRA form), 256x128x64 3-stage ring (skinny N), 128x128x64 single stage
//   (M < 1024). Kept as measured alternatives behind GSL_GEMM_VARIANT: the single-phase 256x256 kernels (4) and the 256x128x32
//   two-workgroups-per-CU tile (9). Operands are swapped in the MFMA (mfma(W, A)) so each lane owns 4 consecutive output columns
//   of one row. Global->LDS staging is the LDS-DMA (global_load_lds_dwordx4); the 16-byte-chunk XOR swizzle (chunk ^= row & 7) is
//   applied on the DMA source address and again on the ds_read_b128 fragment read (0 bank conflicts measured). Block ids are
//   remapped so the N-tiles of one A row-panel run on one XCD, and N-tile j starts its K loop at K tile j. Epilogue operands
//   (outputs, residual, GELU') go through a wave-private LDS staging area and touch HBM as full rows; outputs use non-temporal
//   stores. gsl_gemm_nt_lora_mulgrad (end of file) is the FFN2-dX GEMM with the two LoRA-gradient reductions of its tiles fused
//   into the epilogue. What was tried and measured is in profiles/r01_gemm_ab.md and DESIGN.md section 4.
// f32 path (parity mode): 64x64x16 tile, 4x4 outputs per thread, sequential fmaf over k.
#include <stdlib.h>

#include "gsl_common.h"
#include <type_traits>

using namespace gsl;
#include "gsl_h16.h"
GSL_OPNS_BEGIN

typedef op16x8_t bf16x8_t;      // MFMA operand in this translation unit's 16-bit format (gsl_common.h)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct EpiArgs {
  float alpha;
  const float* bias;
  const void* res;     // f32 (BIAS_RES_F32) or the operand dtype (BIAS_RES_BF16)
  const void* aux;
  void* out;
  void* out2;
  int ldo;
  const float* pos;
  const float* cls;
  int T;
  DropCfg drop;
  int M, N;
  int mbase;   // rows in front of this launch's row 0 in the caller's tensor (tail split, see launch_tail_split): enters the dropout counters only
  int remap;   // block-id -> tile mapping (development knob GSL_XCD_REMAP; 1 = XCD-contiguous)
  int pf;      // 8-phase kernel: after its K loop a workgroup touches the first A lines of the tile that takes a slot of its XCD next (see the kernel)
  int mrev;    // 8-phase kernel: tiles in reverse order (the consumer starts on the rows its producer wrote last: still in the 256 MB Infinity Cache)
  int stmode;  // output store flavour of the staged bf16 epilogue (development knob GSL_STORE_MODE, see store_stream16)
  int f16;     // BIAS_RES_BF16 / PATCH_BF16 templates: the residual stream (res in, out) is IEEE fp16 instead of bf16 (GSL_EPI_BIAS_RES_F16 / PATCH_F16)
  int krot;    // 8-phase kernel: N-tile j starts its K loop at K tile j (mod nk), so sibling tiles of one A panel do not miss on the same lines
  // gradient-fused MUL epilogue (gsl_gemm_nt_lora_mulgrad): operands of the two LoRA-gradient reductions that consume this tile
  // STORE with a consumer-side LayerNorm (GSL_EPI_STORE_LN / STORE_QKV_HM_LN): A is the RAW residual stream x, W the weight with gamma folded in
  // (W' = W * gamma along K), and the epilogue finishes the normalisation: out = rstd[m] * (acc - mean[m] * c[n]) + d[n], c = rowsum(W'), d = W beta (+ bias)
  const float* ln_mean; const float* ln_rstd; const float* ln_c; const float* ln_d; int ln_rs;      // ln_rs: row m reads mean / rstd [m * ln_rs] (1, or T for the cls rows of a [B*T] tensor)
  const bf16_t* gu1; int ldgu1;   // U1 [M, >= 16]: G1[n, j] = sum_m out[m, n] * U1[m, j]
  const bf16_t* gy2;              // Y2 [M, N] (row stride ldo): G2[n, j] = sum_m Y2[m, n] * t[m, j]
  float* gpart1; float* gpart2;   // per-M-tile partial sums [M tiles][N][gR]
  int gR;                         // 8 or 16
  int hmT, hmH;                   // STORE_QKV_HM: tokens per image and heads (0 = plain row-major output)
  int o4_delay;                   // development, overlap GEMM (gemm_o4.inc, GSL_O4_DELAY): cycles the second workgroup of a CU's first round holds back (0 = none)
  int stamps_all;                 // development (GSL_P8_STAMPS_ALL): every workgroup stamps
  unsigned long long* stamps;     // development (GSL_P8_STAMPS = device address of 256 x 4 u64): cycle stamps of every 64th workgroup of the 8-phase kernel
};

// ---- 8-bit GELU' (bf16 speed mode). The second output of the fused FFN1 epilogue, g' = GELU'(a) * keep / (1 - p), is read once, by the
// FFN2-dX epilogue, as a multiplier; GELU' lives in [-0.129, 1.129], so the tensor is stored as an unsigned fixed-point code
//   q = rne(GELU'(a) * keep * 200 + 26)  in [0, 252]   (v_cvt_pk_u8_f32 rounds to nearest; step 0.005; keep = 0 -> q = 26 -> decodes to exactly 0)
//   g' = (q - 26) * 0.005 / (1 - p)
// i.e. an absolute error <= 0.0025 / (1 - p) where bf16 has a relative one of 2^-9: half the bytes of the [M, mlp] tensor that the
// FFN1 forward writes and the HBM-bound FFN2-dX epilogue reads (profiles/r03_*). GSL_EPI_BIAS_GELU_G8 writes it, GSL_EPI_MUL_G8 /
// gsl_gemm_nt_lora_mulgrad(aux_u8) read it; both take the dropout rate of the forward for the 1 / (1 - p).
constexpr float G8_K = 200.0f, G8_O = 26.0f;
template <int EPI> constexpr bool epi_is_mul() { return EPI == GSL_EPI_MUL || EPI == GSL_EPI_MUL_G8; }
template <int EPI> constexpr bool epi_is_gelu() { return EPI == GSL_EPI_BIAS_GELU || EPI == GSL_EPI_BIAS_GELU_G8; }
// Layout of the code tensor: SLAB-MAJOR [N / 64][M][64] — the 64 columns a wave owns in both kernels are one contiguous 64-byte piece per
// row and consecutive rows follow each other, so the 16-row store instruction of the FFN1 epilogue writes 1 KB and the 8-row load
// instruction of the FFN2-dX epilogue reads 512 B of consecutive memory (row-major [M, N] made them 64-byte pieces 2 KB apart: +8 % write
// traffic by the PMC counters). The tensor is private to this pair of epilogues (never a GEMM operand); N % 64 == 0.
__device__ __forceinline__ size_t g8_off(int M, int m, int n) { return ((size_t)(n >> 6) * (size_t)M + (size_t)m) * 64 + (size_t)(n & 63); }
// kq = 200 * (1 - p): g (already scaled by keep / (1 - p)) -> code
__device__ __forceinline__ uint32_t g8_pack4(const float g[4], float kq) {
  uint32_t w = 0u;
  w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[0], kq, G8_O), 0, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[1], kq, G8_O), 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[2], kq, G8_O), 2, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(g[3], kq, G8_O), 3, w);
  return w;
}
// sq = 0.005 / (1 - p): code -> g'
__device__ __forceinline__ void g8_unpack4(uint32_t w, float sq, float g[4]) {
  const float o = -G8_O * sq;
  g[0] = fmaf((float)(w & 0xffu), sq, o);
  g[1] = fmaf((float)((w >> 8) & 0xffu), sq, o);
  g[2] = fmaf((float)((w >> 16) & 0xffu), sq, o);
  g[3] = fmaf((float)(w >> 24), sq, o);
}

// ---- GELU / GELU' of the 8-phase kernel's BIAS_GELU_G8 epilogue as ONE table gather per element (round 4). The epilogue was bound by
// VALU issue (profiles/r03_h_wg_timeline.md: 30 k cycles per 256x256 tile with the matrix pipe idle; ~27 instructions per element, 14 of
// them the A&S erf / exp / rcp pair, computed to 1.5e-7 for a value that is stored as bf16 and an 8-bit code). tools/gen_gelu_table.py:
// entry i = { f32 bits of Phi(a_i) with a 16-bit mantissa | 8-bit GELU'(a_i) code }, a_i = -4.5 + i * 9 / 4095. The workgroup copies the
// 16 KB table into LDS behind the K-loop stages (16 LDS-DMA pieces behind the prologue's requests), the epilogue clamps, scales and truncates
// the pre-activation into a byte address (v_med3, v_fma, v_cvt_u32, v_and), gathers with ds_read_b32, applies the dropout mask to the
// ENTRY (one v_cndmask for both outputs: a dropped element takes {Phi = 0, code 26}) and multiplies a / (1 - p) with the entry read as a
// float (the code byte is < 2^-15 relative noise). Error of Phi at the nearest grid point <= phi(0) * 0.0011 = 4.5e-4 (a quarter of the bf16
// half-ulp of h around a = 0), of GELU' <= 9e-4 + half a code step. profiles/r04_valu_rate.md prices the instruction classes.
constexpr int GT_N = 4096;                       // entries
constexpr float GT_R = 4.5f;                     // table range [-R, R]
constexpr float GT_K4 = 4.0f * (GT_N - 1) / (2.0f * GT_R);      // byte-address scale: 4 / D = 1820
constexpr float GT_C4 = (GT_R * (GT_N - 1) / (2.0f * GT_R) + 0.5f) * 4.0f;      // (R / D + 1/2) * 4 = 8192
constexpr uint32_t GT_DROPPED = 0x0000001Au;     // {Phi = 0, code G8_O}
__device__ __attribute__((aligned(16))) const uint32_t GELU_G8_TAB[GT_N] = {
#include "gelu_g8_table.inc"
};
typedef __attribute__((address_space(3))) const uint32_t* lds_u32p_t;
// tab_c4 = GT_C4 + LDS byte address of the table (exact in f32: < 2^24)
// the same lookup from the pre-activation already multiplied by the dropout scale s = 1 / (1 - p): rs = R s, k4s = K4 / s
__device__ __forceinline__ uint32_t gelu_tab_entry_scaled(float as, float rs, float k4s, float tab_c4) {
  const float t = fmaf(__builtin_amdgcn_fmed3f(as, -rs, rs), k4s, tab_c4);
  return *(lds_u32p_t)(uintptr_t)(((uint32_t)t) & ~3u);
}
__device__ __forceinline__ uint32_t gelu_tab_entry(float a, float tab_c4) {
  const float t = fmaf(__builtin_amdgcn_fmed3f(a, -GT_R, GT_R), GT_K4, tab_c4);
  return *(lds_u32p_t)(uintptr_t)(((uint32_t)t) & ~3u);
}

// ---- epilogue, split in two: the arithmetic on one (row m, 4 consecutive columns n..n+3) fragment, and the store.
// N % 4 == 0 is enforced by the host wrapper. v = primary output, g = second output (GELU' of BIAS_GELU).
// bp: the 4 bias values of columns n..n+3 already in registers (the staged epilogues load them once per wave: a per-fragment
// global load + s_waitcnt vmcnt(0) serialises the VALU-bound epilogue), or nullptr to load them here.
// alpha is honoured by the STORE / STORE_F32 / MUL epilogues only (the only callers that pass alpha != 1 are the LoRA
// down-projections); the host wrapper rejects alpha != 1 for the others.
// HASW: w0 is the first-stage dropout hash value of the fragment's first element pair (drop_w0), advanced by the caller with one
// add per fragment instead of a 64-bit index and a quarter-rate multiply here.
template <int EPI, typename T, bool HASW = false>
__device__ __forceinline__ void epi_math(const EpiArgs& e, int m, int n, float v[4], float g[4], const float* bp = nullptr, uint32_t w0 = 0u) {
  const size_t off = (size_t)m * e.ldo + n;
  const uint64_t lin = (uint64_t)(m + e.mbase) * (uint64_t)e.N + (uint64_t)n;
  if constexpr (EPI == GSL_EPI_STORE || EPI == GSL_EPI_STORE_F32 || epi_is_mul<EPI>()) {
    if (e.alpha != 1.0f) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= e.alpha;
    }
  }
  float bq[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == GSL_EPI_BIAS_RES_F32 || epi_is_gelu<EPI>() || EPI == GSL_EPI_PATCH || EPI == GSL_EPI_BIAS_RES_BF16 ||
                EPI == GSL_EPI_PATCH_BF16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = bp ? bp[i] : e.bias[n + i];
  }
  if constexpr (EPI == GSL_EPI_STORE || EPI == GSL_EPI_STORE_F32) {
    if (e.ln_rstd) {          // consumer-side LayerNorm (fragment-path kernels; the staged epilogue applies it from preloaded values and passes bp)
      if (!bp) {
        const float rs = e.ln_rstd[(size_t)m * e.ln_rs], rm = rs * e.ln_mean[(size_t)m * e.ln_rs];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], rs, fmaf(-rm, e.ln_c[n + i], e.ln_d[n + i]));
      }
    } else if (e.bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += bp ? bp[i] : e.bias[n + i];      // (bp: the staged epilogue's preloaded values — no global load per fragment)
    }
  } else if constexpr (EPI == GSL_EPI_BIAS_RES_F32) {
    float r[4], dm[4];
    Elem<float>::ld4(reinterpret_cast<const float*>(e.res) + off, r);
    drop_mul4(e.drop, lin, dm);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (v[i] + bq[i]) * dm[i] + r[i];
  } else if constexpr (EPI == GSL_EPI_BIAS_RES_BF16) {      // the residual stream in 2 bytes per element (bf16, or fp16 with e.f16): f32 arithmetic, one rounding on store
    float r[4], dm[4];
    {
      const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(e.res) + off);
      unpack2s(t.x, e.f16, r[0], r[1]); unpack2s(t.y, e.f16, r[2], r[3]);
    }
    drop_mul4(e.drop, lin, dm);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (v[i] + bq[i]) * dm[i] + r[i];
  } else if constexpr (epi_is_gelu<EPI>()) {
    float dm[4];
    if constexpr (HASW) drop_mul4_w(e.drop, w0, dm);
    else drop_mul4(e.drop, lin, dm);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = v[i] + bq[i];
      float ga, gpa;
      if constexpr (sizeof(T) == 2) gelu_pair_fast(a, ga, gpa);      // bf16 speed mode
      else { ga = gelu_f(a); gpa = e.out2 ? gelu_grad_f(a) : 0.f; }   // f32 parity mode: exact erf (no second output in eval mode: GELU' is skipped, wave-uniformly)
      v[i] = ga * dm[i];
      g[i] = gpa * dm[i];
    }
  } else if constexpr (EPI == GSL_EPI_MUL) {
    float a[4];
    Elem<T>::ld4(reinterpret_cast<const T*>(e.aux) + off, a);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= a[i];
  } else if constexpr (EPI == GSL_EPI_MUL_G8) {
    float a[4];
    g8_unpack4(*reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(e.aux) + g8_off(e.M, m, n)), 1.0f / (G8_K) * e.drop.scale, a);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= a[i];
  } else if constexpr (EPI == GSL_EPI_PATCH || EPI == GSL_EPI_PATCH_BF16) {
    const int tok = m % e.T;
    float dm[4];
    drop_mul4(e.drop, lin, dm);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float base = (tok == 0) ? e.cls[n + i] : (v[i] + bq[i]);
      v[i] = (base + e.pos[(size_t)tok * e.N + n + i]) * dm[i];
    }
  }
}
template <int EPI> constexpr bool epi_out_is_f32() { return EPI == GSL_EPI_STORE_F32 || EPI == GSL_EPI_BIAS_RES_F32 || EPI == GSL_EPI_PATCH; }

// direct (fragment-layout) store: 16-byte stores for f32 outputs, 8-byte for bf16 outputs
template <int EPI, typename T>
__device__ __forceinline__ void epilogue4(const EpiArgs& e, int m, int n, float v[4]) {
  if (m >= e.M || n >= e.N) return;
  float g[4];
  epi_math<EPI, T>(e, m, n, v, g);
  size_t off = (size_t)m * e.ldo + n;
  if constexpr (EPI == GSL_EPI_STORE) {
    if (e.hmT) {          // STORE_QKV_HM: (row b T + t, column (which, h, d)) -> [b][h][which][t][d]
      const int b = m / e.hmT, t = m - b * e.hmT, pn = n >> 6, which = pn / e.hmH, h = pn - which * e.hmH;
      off = ((((size_t)b * e.hmH + h) * 3 + which) * (size_t)e.hmT + t) * 64 + (n & 63);
    }
  }
  if constexpr (epi_out_is_f32<EPI>()) {
    Elem<float>::st4(reinterpret_cast<float*>(e.out) + off, v);
  } else {
    if constexpr (EPI == GSL_EPI_BIAS_RES_BF16 || EPI == GSL_EPI_PATCH_BF16)      // the stream output: bf16 or fp16
      *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(e.out) + off) = make_uint2(pack2s(v[0], v[1], e.f16), pack2s(v[2], v[3], e.f16));
    else
      Elem<T>::st4(reinterpret_cast<T*>(e.out) + off, v);
    if constexpr (EPI == GSL_EPI_BIAS_GELU) { if (e.out2) Elem<T>::st4(reinterpret_cast<T*>(e.out2) + off, g); }
    if constexpr (EPI == GSL_EPI_BIAS_GELU_G8) { if (e.out2) *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(e.out2) + g8_off(e.M, m, n)) = g8_pack4(g, G8_K / e.drop.scale); }
  }
}

// Store flavour of the staged epilogues. PRODUCT build: a compile-time constant — as a run-time EpiArgs field every one of the 16 - 24 stores of a
// wave's epilogue sat behind a three-way uniform branch (plus s_waitcnt lgkmcnt(0) straight behind its ds_read_b128), and a staged epilogue of
// ~400 instructions took 10 - 15 k cycles whether or not anything was stored (profiles/r06_k_epilogue_branches.md). Development build: the
// GSL_STORE_MODE knob.
#ifndef GSL_STMODE
#define GSL_STMODE 1      // output stores of the staged epilogues: 0 plain, 1 non-temporal, 2 sc1 (store_stream16)
#endif
#if defined(GSL_DEV) && !defined(GSL_STMODE_FIXED)      // (-DGSL_STMODE_FIXED: a dev build with the product's compile-time store mode, for stamps)
#define GSL_STMODE_OF(e) ((e).stmode)
#else
#define GSL_STMODE_OF(e) GSL_STMODE
#endif
// 16-byte output store. mode 0: plain (write-back, the line stays in the XCD's L2), 1: non-temporal hint, 2: sc1 (the line is dropped
// from L2): the GEMM outputs are streamed once and are 4x larger than the operand panels they would otherwise evict.
__device__ __forceinline__ void store_stream16(void* p, const uint4 v, int mode) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t w = {v.x, v.y, v.z, v.w};
#ifdef GSL_DEV
  if (mode == 3) return;      // development (GSL_STORE_MODE=3): no store at all — what an epilogue costs without its HBM writes
#endif
  if (mode == 1) __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(p));
  else if (mode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
  else *reinterpret_cast<u32x4_t*>(p) = w;
}

// bf16-output epilogue of one wave's (NI*16 rows) x 64 columns sub-tile, staged through a wave-private LDS region so
// that global stores are 16 bytes per lane over FULL 128-byte rows (8 lanes per row, 8 rows per instruction).
// Measured: the fragment-layout 8-byte stores touch 16 partial (32-byte) lines per instruction and bound the K = 512
// GEMMs by store transactions, not bytes (f32 output with 2x the bytes costs +5 %; profiles/r01_gemm_ab.md).
// cst: wave-private LDS, NOUT * 64 rows * CLD bf16. acc tiles are processed in chunks of 4 row-fragments (64 rows).
constexpr int CLD = 72;   // 144-byte rows: 16-byte aligned for ds_read_b128, 2-way at worst on the ds_write_b64
// SEQ = false: both BIAS_GELU outputs staged side by side (2 x 64 rows per wave). SEQ = true: one 64-row region per wave,
// the second output waits in registers and is staged after the first was copied out (half the LDS: two workgroups per CU).
// bias_lds: this wave's 64 bias values staged in LDS by the caller (the 128-register SEQ kernel cannot afford 16 more VGPRs), or
// nullptr: the lane's 16 values are loaded into registers once.
// SMODE (STORE only; chosen ONCE per wave by the wrapper from wave-uniform launch arguments — decided per fragment, the same flags cost two taken
// branches and a dozen selects in front of each of the 32 fragments, ~9 k cycles of a 256x256 tile: profiles/r06_k_epilogue_branches.md):
//   0 = alpha * accumulator (no bias, no LayerNorm), 1 = the consumer-side LayerNorm, 2 = the general form (alpha, bias)
// DK (table epilogue): 1 = dropout on, 2 = off — decided once per wave like SMODE; 0 = tested where it is used. (Tried and dropped: the keep bits of
// the wave's sub-tile precomputed under the prologue wait + v_bfe_i32 / v_bfi_b32 selects — 17 of ~47 issue slots per fragment less and the epilogue's
// 13.0 k cycles did not move, while the prologue grew by 4 k: this epilogue is bound by its LDS traffic — 128 table gathers, 64 staging writes and 24
// row reads per wave —, not by VALU issue: profiles/r06_k_epilogue_branches.md.)
template <int EPI, int NI, bool SEQ, bool FULL, bool TAB = false, int SMODE = 2, int DK = 0>
__device__ __forceinline__ void epilogue_staged_bf16_impl(const EpiArgs& e, f32x4_t (&acc)[NI][4], bf16_t* cst, int mw, int nw, int lane,
                                                          const float* bias_lds, uint32_t tab_lds = 0u) {
  static_assert(!TAB || EPI == GSL_EPI_BIAS_GELU_G8, "the GELU table serves the 8-bit-code epilogue");
  // A dropped element's table entry is GT_DROPPED = 0x0000001A: the code byte of g' = 0 and, read as a float, a DENORMAL. From here to the end of
  // the wave f32 denormals are flushed (MODE.FP_DENORM[5:4] = 0), so the product (a * scale) * entry is EXACTLY +-0 for every finite a — without
  // the flush it was a ~3.6e-44 denormal times a: +-0 after the 16-bit rounding for |a| < ~1e3 only — and NaN for a = +-Inf / NaN
  // (torch: Inf * 0). No per-element instruction.
  if constexpr (TAB) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 0");
  const float tab_c4 = GT_C4 + (float)tab_lds;
  constexpr int NOUT = epi_is_gelu<EPI>() ? 2 : 1;
  constexpr bool G8 = (EPI == GSL_EPI_BIAS_GELU_G8);      // second output as the 8-bit GELU' code: staged as bytes (80-byte rows: conflict-free
  static_assert(!(G8 && SEQ), "the 8-bit GELU' output is staged beside the first one");      // dword writes), copied out as 64-byte row segments
  uint8_t* c8 = reinterpret_cast<uint8_t*>(cst + 64 * CLD);
  const float kq8 = G8_K / e.drop.scale;
  const int fr = lane & 15, fc = lane >> 4;
  const int crow = lane >> 3, cch = lane & 7;
  float bj[4][4];        // this lane's bias values for its 4 column fragments
  // consumer-side LayerNorm (STORE only, EpiArgs::ln_*): bj holds d[n], cj the column sums c[n]; the row statistics are loaded per 64-row chunk
  const bool ln = (EPI == GSL_EPI_STORE) && e.ln_rstd != nullptr;      // wave-uniform
  float cj[4][4];
  if (!bias_lds) {
    const float* bsrc = ln ? e.ln_d : e.bias;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nw + j * 16 + fc * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bj[j][i] = (bsrc && (FULL || n < e.N)) ? bsrc[n + i] : 0.f;
        if constexpr (EPI == GSL_EPI_STORE) cj[j][i] = (ln && (FULL || n < e.N)) ? e.ln_c[n + i] : 0.f;
      }
    }
  }
  // dropout: first-stage hash value of this lane's first fragment; fragment (i, j) is 16 i rows and 16 j columns further, i.e.
  // 8 N i + 8 j element pairs (N % 4 == 0): one uniform offset and one add per fragment
  const size_t rowoff0 = (size_t)(mw + crow) * (size_t)e.ldo + (size_t)(nw + cch * 8);     // copy-out: lane's row 0 of the wave tile
  // STORE_QKV_HM: row m = b T + t of the wave's 64 columns (one (q|k|v, head) panel) goes to [b][h][which][t][64]; the lane walks its
  // rows in steps of 8, so (b, t) is divided out once and advanced incrementally
  int ht = 0;
  size_t hrow = 0, hwrap = 0;        // running element offset of the lane's current row inside [b][h][which]; what a wrap into the next image adds
  const bool hm = (EPI == GSL_EPI_STORE) && e.hmT != 0;      // wave-uniform
  if constexpr (EPI == GSL_EPI_STORE) {
    if (e.hmT) {
      const int hb = (mw + crow) / e.hmT;
      ht = (mw + crow) - hb * e.hmT;
      const int pn = nw >> 6, which = pn / e.hmH, h = pn - which * e.hmH;
      const size_t img = (size_t)e.hmH * 3 * (size_t)e.hmT * 64;
      hrow = (size_t)hb * img + ((size_t)h * 3 + (size_t)which) * (size_t)e.hmT * 64 + (size_t)ht * 64 + (size_t)(cch * 8);
      hwrap = img - (size_t)e.hmT * 64;
    }
  }
  // table epilogue: lane bases of the two staging areas (LDS byte addresses, opaque to the optimiser so that they stay in registers), the bias
  // pre-multiplied by the dropout scale, and the table constants for the scaled pre-activation a' = a / (1 - p): index = clamp(a') * (K4 / s) + C4
  typedef __attribute__((address_space(3))) char* lds_cp_t;
  lds_cp_t dbase_l = (lds_cp_t)(cst + fr * CLD + fc * 4), cbase_l = (lds_cp_t)(c8 + fr * 80 + fc * 4);
  float bjs[4][4];
  const float tscale = e.drop.scale, tab_rs = GT_R * e.drop.scale, tab_k4s = GT_K4 / e.drop.scale;
  if constexpr (TAB) {
    asm volatile("" : "+v"(dbase_l), "+v"(cbase_l));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bjs[j][i] = bj[j][i] * tscale;
        asm volatile("" : "+v"(bjs[j][i]));      // (kept: the compiler would re-multiply bias * scale in front of every fragment)
      }
  }
  constexpr bool DROPW = epi_is_gelu<EPI>();
  uint32_t wbase = 0u, rowstep = 0u;
  if constexpr (DROPW) {
    if (e.drop.thr) {
      wbase = drop_w0(e.drop.key, ((uint64_t)(mw + e.mbase + fr) * (uint64_t)e.N + (uint64_t)(nw + fc * 4)) >> 1);
      rowstep = (8u * (uint32_t)e.N) * DROP_PHI;
    }
  }
  // consumer-side LayerNorm: rstd and rstd * mean of ALL of this lane's rows, loaded before the first chunk — loaded per chunk, the wait for the
  // second chunk's statistics (s_waitcnt vmcnt(0): loads and stores retire through one counter) also waited for the first chunk's eight stores to
  // reach memory
  float rsa[NI], rma[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) rsa[i] = rma[i] = 0.f;
  if constexpr (EPI == GSL_EPI_STORE) {
    if (ln) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int mr = min(mw + i * 16 + fr, e.M - 1);
        rsa[i] = e.ln_rstd[(size_t)mr * e.ln_rs];
        rma[i] = rsa[i] * e.ln_mean[(size_t)mr * e.ln_rs];
      }
    }
  }
#pragma unroll
  for (int ib = 0; ib < NI; ib += 4) {
    uint2 held[4][4];   // second output, packed, when SEQ
    const float rsv[4] = {rsa[ib], rsa[ib + 1], rsa[ib + 2], rsa[ib + 3]}, rmv[4] = {rma[ib], rma[ib + 1], rma[ib + 2], rma[ib + 3]};
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = ib + ii;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]}, g[4] = {0.f, 0.f, 0.f, 0.f};
        const int m = mw + i * 16 + fr, n = nw + j * 16 + fc * 4;
        if constexpr (EPI == GSL_EPI_STORE && SMODE != 2) {
          if constexpr (SMODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaf(v[q], rsv[ii], fmaf(-rmv[ii], cj[j][q], bj[j][q]));
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] *= e.alpha;      // (alpha = 1 for all but the LoRA down-projections; x * 1.0f is exact)
          }
          *reinterpret_cast<uint2*>(cst + (ii * 16 + fr) * CLD + j * 16 + fc * 4) = make_uint2(pack2o(v[0], v[1]), pack2o(v[2], v[3]));
          continue;
        }
        if constexpr (EPI == GSL_EPI_STORE) {
          if (ln) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaf(v[q], rsv[ii], fmaf(-rmv[ii], cj[j][q], bj[j][q]));
          }
        }
        if constexpr (TAB) {      // (rows / columns outside the matrix run the same arithmetic on finite values and are not copied out)
          uint32_t ent[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[q] = fmaf(v[q], tscale, bjs[j][q]);      // (acc + bias) / (1 - p): the dropout scale rides in the bias add (bjs = bias * scale)
            ent[q] = gelu_tab_entry_scaled(v[q], tab_rs, tab_k4s, tab_c4);
          }
          if (DK == 1 || (DK == 0 && e.drop.thr)) {
            const uint32_t w0 = wbase + ((uint32_t)i * rowstep + (uint32_t)(j * 8) * DROP_PHI);
            const uint32_t h0 = drop_finish(w0), h1 = drop_finish(w0 + DROP_PHI);
            ent[0] = ((h0 & 0xffffu) < e.drop.thr) ? GT_DROPPED : ent[0];
            ent[1] = ((h0 >> 16) < e.drop.thr) ? GT_DROPPED : ent[1];
            ent[2] = ((h1 & 0xffffu) < e.drop.thr) ? GT_DROPPED : ent[2];
            ent[3] = ((h1 >> 16) < e.drop.thr) ? GT_DROPPED : ent[3];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = v[q] * __uint_as_float(ent[q]);
          // staging addresses = one lane base each (kept in a register: the compiler otherwise rebuilds it from three values per fragment,
          // ~4 of the epilogue's ~59 VALU slots per fragment, and this epilogue is VALU-issue-bound) + a compile-time offset
          typedef unsigned int u32x2_lds_t __attribute__((ext_vector_type(2)));
          *reinterpret_cast<__attribute__((address_space(3))) u32x2_lds_t*>(dbase_l + ((ii * 16) * CLD + j * 16) * 2) =
              u32x2_lds_t{pack2o(v[0], v[1]), pack2o(v[2], v[3])};
          // the four code bytes: v_perm_b32 x 2 + v_or
          *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(cbase_l + (ii * 16) * 80 + j * 16) =
              __builtin_amdgcn_perm(ent[1], ent[0], 0x0c0c0400u) | __builtin_amdgcn_perm(ent[3], ent[2], 0x04000c0cu);
          continue;
        }
        if (FULL || (m < e.M && n < e.N)) {
          const uint32_t w0 = wbase + ((uint32_t)i * rowstep + (uint32_t)(j * 8) * DROP_PHI);
          if (bias_lds) {
            const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bias_lds + j * 16 + fc * 4);
            const float bl[4] = {b4[0], b4[1], b4[2], b4[3]};
            epi_math<EPI, bf16_t, DROPW>(e, m, n, v, g, bl, w0);
          } else {
            epi_math<EPI, bf16_t, DROPW>(e, m, n, v, g, bj[j], w0);      // (a pointer select here would push the arrays into scratch memory)
          }
        }
        bf16_t* d = cst + (ii * 16 + fr) * CLD + j * 16 + fc * 4;
        if constexpr (EPI == GSL_EPI_BIAS_RES_BF16 || EPI == GSL_EPI_PATCH_BF16) *reinterpret_cast<uint2*>(d) = make_uint2(pack2s(v[0], v[1], e.f16), pack2s(v[2], v[3], e.f16));
        else *reinterpret_cast<uint2*>(d) = make_uint2(pack2o(v[0], v[1]), pack2o(v[2], v[3]));
        if constexpr (G8) {
          *reinterpret_cast<uint32_t*>(c8 + (ii * 16 + fr) * 80 + j * 16 + fc * 4) = g8_pack4(g, kq8);
        } else if constexpr (NOUT == 2) {
          const uint2 pg = make_uint2(pack2o(g[0], g[1]), pack2o(g[2], g[3]));
          if constexpr (SEQ) held[ii][j] = pg; else *reinterpret_cast<uint2*>(d + 64 * CLD) = pg;
        }
      }
    // the wave's own DS operations execute in order: the reads below see the writes above (no barrier needed)
    // Copy-out (round 6): all eight row reads of the chunk are issued first, then the eight stores — one LDS round trip per chunk instead of one
    // per store —, the head-major row walk is branch-free (a step of 8 rows crosses at most one image boundary when T >= 8; the host checks),
    // and nothing wave-uniform is decided per store.
#pragma unroll
    for (int pass = 0; pass < (SEQ ? NOUT : 1); ++pass) {
      if (SEQ && pass == 1) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<uint2*>(cst + (ii * 16 + fr) * CLD + j * 16 + fc * 4) = held[ii][j];
      }
      uint4 val[8], val2[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        val[r] = *reinterpret_cast<const uint4*>(cst + (r * 8 + crow) * CLD + cch * 8);
        if constexpr (NOUT == 2 && !SEQ && !G8) val2[r] = *reinterpret_cast<const uint4*>(cst + (64 + r * 8 + crow) * CLD + cch * 8);
      }
      bf16_t* dst = reinterpret_cast<bf16_t*>((SEQ && pass == 1) ? e.out2 : e.out);
      size_t offs[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        // element offset = this lane's first row + a wave-uniform row step (scalar multiply): no per-store 64-bit multiply
        offs[r] = rowoff0 + (size_t)(ib * 16 + r * 8) * (size_t)e.ldo;
        if constexpr (EPI == GSL_EPI_STORE) {
          if (hm) {
            offs[r] = hrow;
            ht += 8; hrow += 8 * 64;
            const bool wrap = ht >= e.hmT;
            ht = wrap ? ht - e.hmT : ht;
            hrow = wrap ? hrow + hwrap : hrow;
          }
        }
      }
      if (dst) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int m = mw + ib * 16 + r * 8 + crow, n = nw + cch * 8;
          if (FULL || (m < e.M && n < e.N)) store_stream16(dst + offs[r], val[r], GSL_STMODE_OF(e));
        }
      }
      if constexpr (NOUT == 2 && !SEQ && !G8) {
        if (e.out2) {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int m = mw + ib * 16 + r * 8 + crow, n = nw + cch * 8;
            if (FULL || (m < e.M && n < e.N)) store_stream16(reinterpret_cast<bf16_t*>(e.out2) + offs[r], val2[r], GSL_STMODE_OF(e));
          }
        }
      }
    }
    if constexpr (G8) {      // 64 rows x 64 code bytes: 4 lanes x 16 B per row, 16 rows per instruction; the four reads first, then the four stores
      if (e.out2) {
        uint4 cv[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) cv[rr] = *reinterpret_cast<const uint4*>(c8 + (rr * 16 + (lane >> 2)) * 80 + (lane & 3) * 16);
        // slab-major code tensor: the wave's 64 columns are one slab (nw % 64 == 0), its rows follow each other 64 bytes apart
        uint8_t* cdst = reinterpret_cast<uint8_t*>(e.out2) + ((size_t)(nw >> 6) * (size_t)e.M + (size_t)(mw + ib * 16 + (lane >> 2))) * 64 + (size_t)((lane & 3) * 16);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int m = mw + ib * 16 + rr * 16 + (lane >> 2), n = nw + (lane & 3) * 16;
          if (FULL || (m < e.M && n < e.N)) store_stream16(cdst + rr * 16 * 64, cv[rr], GSL_STMODE_OF(e));
        }
      }
    }
  }
}
// tiles completely inside the matrix (all of them when M % 256 == 0, N % 128 == 0) skip every per-fragment bounds test
template <int EPI, int NI, bool SEQ = false, bool TAB = false>
__device__ __forceinline__ void epilogue_staged_bf16(const EpiArgs& e, f32x4_t (&acc)[NI][4], bf16_t* cst, int mw, int nw, int lane,
                                                     const float* bias_lds = nullptr, uint32_t tab_lds = 0u) {
  if constexpr (SEQ) {      // the 128-register kernel: two inlined copies of the epilogue make the allocator spill (measured: 81 VGPRs)
    epilogue_staged_bf16_impl<EPI, NI, SEQ, false>(e, acc, cst, mw, nw, lane, bias_lds);
  } else {
    const bool full = mw + NI * 16 <= e.M && nw + 64 <= e.N;
    if constexpr (EPI == GSL_EPI_STORE) {
      if (full && !bias_lds) {      // whole tiles (all of them at the step's shapes): the specialised forms
        if (e.ln_rstd != nullptr) { epilogue_staged_bf16_impl<EPI, NI, SEQ, true, TAB, 1>(e, acc, cst, mw, nw, lane, bias_lds, tab_lds); return; }
        if (!e.bias) { epilogue_staged_bf16_impl<EPI, NI, SEQ, true, TAB, 0>(e, acc, cst, mw, nw, lane, bias_lds, tab_lds); return; }
      }
    }
    if constexpr (TAB) {
      if (full) {
        if (e.drop.thr) epilogue_staged_bf16_impl<EPI, NI, SEQ, true, TAB, 2, 1>(e, acc, cst, mw, nw, lane, bias_lds, tab_lds);
        else epilogue_staged_bf16_impl<EPI, NI, SEQ, true, TAB, 2, 2>(e, acc, cst, mw, nw, lane, bias_lds, tab_lds);
        return;
      }
    }
    if (full) epilogue_staged_bf16_impl<EPI, NI, SEQ, true, TAB>(e, acc, cst, mw, nw, lane, bias_lds, tab_lds);
    else epilogue_staged_bf16_impl<EPI, NI, SEQ, false, TAB>(e, acc, cst, mw, nw, lane, bias_lds, tab_lds);
  }
}
// MUL epilogue (dX * GELU'): the aux operand is as large as the output. Loading it in fragment layout has the same partial-line
// problem as the stores had, so here the f32 accumulators are staged (same wave-private region, 64 rows x 68 floats), read back
// row-contiguous, and aux is loaded / the result stored as full 128-byte rows (16 B per lane). Same arithmetic as epi_math<MUL>:
// bf16( alpha * acc * f32(aux) ), one rounding — bit-identical to the fragment-layout path.
constexpr int CLF = 68;    // floats per staged row (272 B: 16-byte aligned)
template <int NI, int EPI = GSL_EPI_MUL>
__device__ __forceinline__ void epilogue_staged_mul(const EpiArgs& e, f32x4_t (&acc)[NI][4], float* cst, int mw, int nw, int lane) {
  constexpr bool G8 = (EPI == GSL_EPI_MUL_G8);      // aux = the 8-bit GELU' code: 8 bytes per lane and row instead of 16
  const int fr = lane & 15, fc = lane >> 4;
  const int crow = lane >> 3, cch = lane & 7;
  const bf16_t* aux = reinterpret_cast<const bf16_t*>(e.aux);
  const uint8_t* aux8 = reinterpret_cast<const uint8_t*>(e.aux);
  const float sq8 = e.drop.scale / G8_K;
  bf16_t* out = reinterpret_cast<bf16_t*>(e.out);
#pragma unroll
  for (int ib = 0; ib < NI; ib += 4) {
    uint4 ax[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {      // aux rows of this 64-row chunk: issued first, consumed after the LDS round trip
      const int m = min(mw + ib * 16 + r * 8 + crow, e.M - 1), n = min(nw + cch * 8, e.N - 8);
      if constexpr (G8) { const uint2 t = *reinterpret_cast<const uint2*>(aux8 + g8_off(e.M, m, n)); ax[r].x = t.x; ax[r].y = t.y; }
      else ax[r] = *reinterpret_cast<const uint4*>(aux + (size_t)m * e.ldo + n);
    }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4_t v = acc[ib + ii][j];
        v *= e.alpha;      // (x * 1.0f is exact: cheaper than a wave-uniform branch in front of every fragment)
        *reinterpret_cast<f32x4_t*>(cst + (ii * 16 + fr) * CLF + j * 16 + fc * 4) = v;
      }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = r * 8 + crow;
      const int m = mw + ib * 16 + row, n = nw + cch * 8;
      const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(cst + row * CLF + cch * 8);
      const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(cst + row * CLF + cch * 8 + 4);
      uint32_t o[4];
      if constexpr (G8) {
        float ga[4], gb[4];
        g8_unpack4(ax[r].x, sq8, ga);
        g8_unpack4(ax[r].y, sq8, gb);
        o[0] = pack2o(lo[0] * ga[0], lo[1] * ga[1]); o[1] = pack2o(lo[2] * ga[2], lo[3] * ga[3]);
        o[2] = pack2o(hi[0] * gb[0], hi[1] * gb[1]); o[3] = pack2o(hi[2] * gb[2], hi[3] * gb[3]);
      } else {
        const uint32_t a[4] = {ax[r].x, ax[r].y, ax[r].z, ax[r].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float c0 = (k < 2) ? lo[2 * k] : hi[2 * k - 4], c1 = (k < 2) ? lo[2 * k + 1] : hi[2 * k - 3];
          float a0, a1;
          unpack2o(a[k], a0, a1);
          o[k] = pack2o(c0 * a0, c1 * a1);
        }
      }
      if (m < e.M && n < e.N) store_stream16(out + (size_t)m * e.ldo + n, make_uint4(o[0], o[1], o[2], o[3]), GSL_STMODE_OF(e));
    }
  }
}
// MUL epilogue with the two LoRA-gradient reductions of its tile fused in (FFN2-dX: out = dZ = (dY W2 + t A2) * GELU'):
//   G1[n, j] = sum_m dZ[m, n] * U1[m, j]     (dB1: the standalone reduction re-reads the [M, N] tile this epilogue just produced)
//   G2[n, j] = sum_m Y2[m, n] * t[m, j]      (dA2: Y2 = h has the shape and tiling of aux; t = s dY B2^T already sits in LDS)
// Both are contractions over the ROW index, i.e. operands with the contraction index slow: the bf16 tiles go through wave-private
// LDS slabs and are gathered with ds_read_b64_tr_b16 exactly as in lora_grad_mfma_kernel (lora.hip); the matrix pipe is idle
// during the epilogue, so the 16 MFMAs per 32 rows are free. 32-row chunks keep four disjoint regions per wave:
//   [f32 accumulators 32 x 68][dZ 32 x 72 bf16][Y2 32 x 72 bf16][U1 32 x 16 bf16]; t lives behind the wave regions as [256][16].
// The two wave rows are summed through LDS, and every M tile writes its [256 cols][R] partial (fixed-order reduction afterwards).
constexpr int GF_STAGE_B = 32 * CLF * 4;                        // 8704
constexpr int GF_Y_B = 32 * CLD * 2;                            // 4608
constexpr int GF_U_B = 32 * 16 * 2;                             // 1024
constexpr int GF_WAVE_B = GF_STAGE_B + 2 * GF_Y_B + GF_U_B;     // 18944
constexpr int GF_T16_B = 256 * 16 * 2;                          // 8192
constexpr int GF_BLOCK_B = 8 * GF_WAVE_B + GF_T16_B;            // 159744 of the CU's 163840 bytes
typedef short gf_v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) gf_v4s_t* gf_lds_v4s_p;
union GfFrag { gf_v4s_t h[2]; bf16x8_t v; };
template <typename V> __device__ __forceinline__ V gf_ld(const void* p) {      // read-once epilogue operand: non-temporal load
  {
    if constexpr (sizeof(V) == 16) {
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
      const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
      return make_uint4(t[0], t[1], t[2], t[3]);
    } else {
      typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
      return make_uint2(t[0], t[1]);
    }
  }
}
// global operands of one 32-row round of the gradient-fused epilogue: g' and h (4 x 16 B per lane each), U1 (2 x 16 B for lanes 0..31's rows)
struct GfOperands { uint4 ax[4], hx[4], u1; };
// development ablations of the gradient-fused epilogue (build variant -DGSL_MG_ABL=1, mask in GSL_O4_DELAY; results are WRONG by design):
// 2 = no g' / h / U1 loads, 4 = no transposed reads / reduction MFMAs, 8 = no dZ / h / U1 LDS hand-over (with 4), 16 = no f32 staging round trip
#if defined(GSL_DEV) && defined(GSL_MG_ABL)
#define GSL_MG_ON(e, bit) (!((e).o4_delay & (bit)))
#else
#define GSL_MG_ON(e, bit) true
#endif
template <bool G8>
__device__ __forceinline__ void gf_request(const EpiArgs& e, GfOperands& g, int mw, int nw, int ic, int lane) {
  const int crow = lane >> 3, cch = lane & 7;
  const int ncl = min(nw + cch * 8, e.N - 8);
  const bf16_t* aux = reinterpret_cast<const bf16_t*>(e.aux);
  if (!GSL_MG_ON(e, 2)) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = min(mw + ic * 32 + r * 8 + crow, e.M - 1);
    // g' and h are read exactly once by exactly one workgroup: non-temporal loads, so that these 1.2 GB per launch do not
    // push the dY row panel — which the 8 N-tiles of a row panel share through the XCD's L2 — out between sibling tiles
    // (calibrated FETCH_SIZE: 2.0 GB fetched per launch against 1.45 GB of operands, profiles/r04_pmc_calibration.md)
    if constexpr (G8) {      // 8-bit GELU' codes: 8 bytes per lane and row (only .x / .y of the slot are live)
      const uint2 t = gf_ld<uint2>(reinterpret_cast<const uint8_t*>(e.aux) + g8_off(e.M, m, ncl));
      g.ax[r].x = t.x; g.ax[r].y = t.y;
    } else {
      g.ax[r] = gf_ld<uint4>(aux + (size_t)m * e.ldo + ncl);
    }
    g.hx[r] = gf_ld<uint4>(e.gy2 + (size_t)m * e.ldo + ncl);
  }
  // U1: 32 rows x 16 columns per round as ONE instruction (lane l: row l / 2, columns 8 (l % 2) ..): every vector-memory instruction
  // costs the CU's in-order memory pipe time in proportion to the lines it touches (profiles/r05_j_attn_merged.md); two instructions with
  // lanes 32..63 duplicating lanes 0..31 touched the same 32 lines twice (FFN2-dX 0.814 -> 0.80 ms; with a compact [M, 16] U1: 0.79)
  g.u1 = *reinterpret_cast<const uint4*>(e.gu1 + (size_t)min(mw + ic * 32 + (lane >> 1), e.M - 1) * e.ldgu1 + (lane & 1) * 8);
}
// go: the operands of round 0, requested by the caller (right after the K loop, in front of the rank-r tail); wreg: this wave's staging region,
// partner: the region of the wave of the OTHER wave row in the same wave column (wave ^ 4) — read only in the final hand-over, behind a barrier
template <int NI, bool G8>
__device__ __forceinline__ void epilogue_staged_mulgrad(const EpiArgs& e, f32x4_t (&acc)[NI][4], char* wreg, char* partner,
                                                        const bf16_t* t16w, int mw, int nw, int lane, int wm, int mtile, GfOperands& go) {
  float* cst = reinterpret_cast<float*>(wreg);
  bf16_t* yd = reinterpret_cast<bf16_t*>(wreg + GF_STAGE_B);
  bf16_t* yh = reinterpret_cast<bf16_t*>(wreg + GF_STAGE_B + GF_Y_B);
  bf16_t* ub = reinterpret_cast<bf16_t*>(wreg + GF_STAGE_B + 2 * GF_Y_B);
  const int fr = lane & 15, fc = lane >> 4;
  const int crow = lane >> 3, cch = lane & 7;
  const int trow = 4 * fc + (fr >> 2), tcol = (fr & 3) * 4;      // this lane's piece of a 4 x 16 block (transpose-read address)
  bf16_t* out = reinterpret_cast<bf16_t*>(e.out);
  f32x4_t g1[4], g2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) g1[t] = g2[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // The global operands of a 32-row round (g': 4 x 8 or 16 B per lane, h: 4 x 16 B, U1: 2 x 16 B) live in TWO register sets that
  // alternate between the rounds: round 0's were requested by the caller (under the rank-r tail), round 1's are requested here, and
  // round q + 2's as soon as round q has consumed its set (after its multiply / LDS hand-over). A round's operands are therefore in
  // flight for more than a whole round (~1.5) instead of half of one, which covers the round trip that grows from ~2.5 k to ~5 k
  // cycles when all CUs stream. The second set costs no registers at the kernel level: the K loop's operand fragments (72 VGPRs) are
  // dead here and every round frees 32 accumulator registers.
  GfOperands gob;
#if defined(GSL_DEV) && defined(GSL_MG_ABL)
  // ablation 0x100 / 0x200 / 0x300: every second wave starts ~1 k / 2 k / 3 k cycles late (are the eight waves of a tile fighting in lockstep?)
  if (((nw >> 6) & 1) && (e.o4_delay & 0x300)) {
    const int d = (e.o4_delay >> 8) & 3;
    if (d == 1) __builtin_amdgcn_s_sleep(16); else if (d == 2) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(48);
  }
#endif
#if defined(GSL_DEV) && defined(GSL_MG_ABL)
  // phase clocks of one wave (the stamping workgroup's wave 0): staging round trip / rows (multiply, stores, LDS hand-over) / requests /
  // reductions / final hand-over, summed over the four rounds, written behind the 256 x 4 stamp table
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, pt = __builtin_readcyclecounter();
#define GSL_MG_PHASE(i) { const unsigned long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - pt; pt = now_; }
#else
#define GSL_MG_PHASE(i)
#endif
  asm volatile("" ::: "memory");
  if (NI / 2 > 1) gf_request<G8>(e, gob, mw, nw, 1, lane);
  asm volatile("" ::: "memory");
  GSL_MG_PHASE(2)
  const float sq8 = e.drop.scale / G8_K;
#pragma unroll
  for (int ic = 0; ic < NI / 2; ++ic) {
    GfOperands& op = (ic & 1) ? gob : go;      // (the loop is fully unrolled: a compile-time choice)
    uint4 (&ax)[4] = op.ax;
    uint4 (&hx)[4] = op.hx;
    uint4 &u1 = op.u1;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4_t v = acc[ic * 2 + ii][j];
        v *= e.alpha;      // (x * 1.0f is exact: cheaper than a wave-uniform branch in front of every fragment)
        if (GSL_MG_ON(e, 16)) *reinterpret_cast<f32x4_t*>(cst + (ii * 16 + fr) * CLF + j * 16 + fc * 4) = v;
      }
    __builtin_amdgcn_sched_barrier(0);    // nothing that consumes the requested operands may be scheduled above the staging (it would drag their wait up)
    // the round's four staged rows are read back at once (round 6: one LDS round trip per round instead of one per row; staging round q + 1
    // right behind these reads, so that its round trip passes under round q's rows, changes nothing: measured, tools/probes/mulgrad_abl.py)
    f32x4_t slo[4], shi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (GSL_MG_ON(e, 16)) {
        slo[r] = *reinterpret_cast<const f32x4_t*>(cst + (r * 8 + crow) * CLF + cch * 8);
        shi[r] = *reinterpret_cast<const f32x4_t*>(cst + (r * 8 + crow) * CLF + cch * 8 + 4);
      } else { slo[r] = acc[ic * 2][r]; shi[r] = acc[ic * 2 + 1][r]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    GSL_MG_PHASE(0)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r * 8 + crow;
      const int m = mw + ic * 32 + row, n = nw + cch * 8;
      const f32x4_t lo = slo[r], hi = shi[r];
      uint32_t o[4];
      if constexpr (G8) {
        float ga[4], gb[4];
        g8_unpack4(ax[r].x, sq8, ga);
        g8_unpack4(ax[r].y, sq8, gb);
        o[0] = pack2o(lo[0] * ga[0], lo[1] * ga[1]); o[1] = pack2o(lo[2] * ga[2], lo[3] * ga[3]);
        o[2] = pack2o(hi[0] * gb[0], hi[1] * gb[1]); o[3] = pack2o(hi[2] * gb[2], hi[3] * gb[3]);
      } else {
        const uint32_t a[4] = {ax[r].x, ax[r].y, ax[r].z, ax[r].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float c0 = (k < 2) ? lo[2 * k] : hi[2 * k - 4], c1 = (k < 2) ? lo[2 * k + 1] : hi[2 * k - 3];
          float a0, a1;
          unpack2o(a[k], a0, a1);
          o[k] = pack2o(c0 * a0, c1 * a1);
        }
      }
      const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
      if (m < e.M && n < e.N) store_stream16(out + (size_t)m * e.ldo + n, ov, GSL_STMODE_OF(e));
      if (GSL_MG_ON(e, 8)) *reinterpret_cast<uint4*>(yd + row * CLD + cch * 8) = ov;
      // rows past M contribute nothing to the reductions (their dZ is finite: clamped operands) — masked where the operand is consumed,
      // not where it was requested, so that no wait for the loads sits in front of the staging above
      if (GSL_MG_ON(e, 8)) *reinterpret_cast<uint4*>(yh + row * CLD + cch * 8) = (m < e.M) ? hx[r] : make_uint4(0u, 0u, 0u, 0u);
    }
    if (GSL_MG_ON(e, 8)) {
      const bool in = mw + ic * 32 + (lane >> 1) < e.M;
      *reinterpret_cast<uint4*>(ub + lane * 8) = in ? u1 : make_uint4(0u, 0u, 0u, 0u);      // row lane / 2, half lane % 2: 16 bytes per lane, contiguous
    }
    asm volatile("" ::: "memory");      // the wave's DS operations execute in order; this only pins the compiler's order
    GSL_MG_PHASE(1)
    // this round's set is consumed: refill it for the round that uses it next
    if (ic + 2 < NI / 2) gf_request<G8>(e, op, mw, nw, ic + 2, lane);
    asm volatile("" ::: "memory");
    GSL_MG_PHASE(2)
    if (!GSL_MG_ON(e, 4)) continue;
    GfFrag b1, b2;
    b1.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gf_lds_v4s_p)(ub + trow * 16 + tcol));
    b1.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gf_lds_v4s_p)(ub + (trow + 16) * 16 + tcol));
    const bf16_t* tb = t16w + ic * 32 * 16;
    b2.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gf_lds_v4s_p)(tb + trow * 16 + tcol));
    b2.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gf_lds_v4s_p)(tb + (trow + 16) * 16 + tcol));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      GfFrag a1, a2;
      const bf16_t* p1 = yd + trow * CLD + t * 16 + tcol;
      const bf16_t* p2 = yh + trow * CLD + t * 16 + tcol;
      a1.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gf_lds_v4s_p)(p1));
      a1.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gf_lds_v4s_p)(p1 + 16 * CLD));
      a2.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gf_lds_v4s_p)(p2));
      a2.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gf_lds_v4s_p)(p2 + 16 * CLD));
      g1[t] = GSL_MFMA16(a1.v, b1.v, g1[t], 0, 0, 0);
      g2[t] = GSL_MFMA16(a2.v, b2.v, g2[t], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
    GSL_MG_PHASE(3)
  }
  // g?[t][r] = G?[n = nw + t*16 + 4*fc + r][j = fr]. The two wave rows of a wave column are summed through LDS and every M tile writes one
  // [64 cols][R] partial per wave column and gradient — 64 * R CONSECUTIVE floats of gpart, so the hand-over also serves as the transpose into
  // full-line stores (round 6: from the fragment layout each lane stored 32 single floats as 32-byte pieces 128 bytes apart — 32 store
  // instructions per wave, ~4.9 k cycles of the CU's in-order memory pipe per tile; now R / 4 sixteen-byte stores per wave). Every wave parks
  // its two partials in its own region as [64][GF_XS] (G1 in columns 0..15, G2 in 16..31; GF_XS = 36: the fragment writes are conflict-free),
  // one barrier, then the wm = 0 wave of the column sums and stores G1's partial and the wm = 1 wave G2's (wm 0's value + wm 1's: as before).
  if (!GSL_MG_ON(e, 128)) return;
  constexpr int GF_XS = 36;
  {
    float* x = reinterpret_cast<float*>(wreg);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[(t * 16 + 4 * fc + r) * GF_XS + fr] = g1[t][r];
        x[(t * 16 + 4 * fc + r) * GF_XS + 16 + fr] = g2[t][r];
      }
  }
  __syncthreads();
  {
    const float* x0 = reinterpret_cast<const float*>(wm == 0 ? wreg : partner) + wm * 16;      // the column's wm = 0 wave ...
    const float* x1 = reinterpret_cast<const float*>(wm == 0 ? partner : wreg) + wm * 16;      // ... and its wm = 1 wave; this wave's gradient
    float* dst = (wm == 0 ? e.gpart1 : e.gpart2) + ((size_t)mtile * e.N + nw) * e.gR;
    const int R = e.gR;
    if ((R & 3) == 0 && nw + 64 <= e.N) {
      for (int q = lane * 4; q < 64 * R; q += 256) {
        const int nl = q / R, j = q - nl * R;
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(x0 + nl * GF_XS + j);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(x1 + nl * GF_XS + j);
        *reinterpret_cast<f32x4_t*>(dst + q) = a + b;
      }
    } else {
      const int cols = min(64, e.N - nw);
      for (int q = lane; q < cols * R; q += 64) {
        const int nl = q / R, j = q - nl * R;
        dst[q] = x0[nl * GF_XS + j] + x1[nl * GF_XS + j];
      }
    }
  }
#if defined(GSL_DEV) && defined(GSL_MG_ABL)
  GSL_MG_PHASE(4)
  if (e.stamps && !e.stamps_all && blockIdx.x < 64 * 256 && (blockIdx.x % 64) == 0 && threadIdx.x == 0) {
    unsigned long long* d = e.stamps + 1024 + (blockIdx.x / 64) * 8;
    for (int i = 0; i < 5; ++i) d[i] = ph[i];
  }
#endif
#undef GSL_MG_PHASE
}
// BIAS_RES_F32 epilogue (out-proj / FFN2 forward: x + drop(acc + bias), f32 stream): same staging, the residual is loaded and the
// result stored as full 256-byte rows (16 lanes x 16 B per row, 4 rows per instruction) instead of 64-byte fragment rows.
// Arithmetic identical to epi_math<BIAS_RES_F32>: (alpha * acc + bias) * mask + res.
template <int NI, int EPI = GSL_EPI_BIAS_RES_F32>
__device__ __forceinline__ void epilogue_staged_res_f32(const EpiArgs& e, f32x4_t (&acc)[NI][4], float* cst, int mw, int nw, int lane) {
  const int fr = lane & 15, fc = lane >> 4;
  const int crow = lane >> 4, cq = lane & 15;      // copy phase: row-in-group, 4-float column group
  float* out = reinterpret_cast<float*>(e.out);
  const int n = nw + cq * 4;
  f32x4_t b4 = f32x4_t{0.f, 0.f, 0.f, 0.f}, c4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (n < e.N) b4 = *reinterpret_cast<const f32x4_t*>(e.bias + n);
  if constexpr (EPI == GSL_EPI_PATCH) { if (n < e.N) c4 = *reinterpret_cast<const f32x4_t*>(e.cls + n); }
  // 32 rows per round, 8 residual loads (8 KB per wave) each; the loads of round q + 1 are issued before round q is processed, so one
  // memory latency is exposed per tile instead of one per round (the operand-fragment registers of the K loop are free by now)
  constexpr int NQ = NI / 2;
  auto fetch = [&](int q, f32x4_t (&rs)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = min(mw + q * 32 + r * 4 + crow, e.M - 1);
      if constexpr (EPI == GSL_EPI_PATCH) rs[r] = *reinterpret_cast<const f32x4_t*>(e.pos + (size_t)(m % e.T) * e.N + min(n, e.N - 4));
      else rs[r] = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(e.res) + (size_t)m * e.ldo + min(n, e.N - 4));
    }
  };
  f32x4_t rsa[8], rsb[8];      // BIAS_RES_F32: the residual rows; PATCH: the position-embedding rows of the tokens (vit_face.py:531-537)
  fetch(0, rsa);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int ib = (q >> 1) * 4, half = q & 1;
    f32x4_t (&rs)[8] = (q & 1) ? rsb : rsa;
    if (q + 1 < NQ) fetch(q + 1, (q & 1) ? rsa : rsb);
    if (half == 0) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4_t v = acc[ib + ii][j];
          v *= e.alpha;      // (x * 1.0f is exact: cheaper than a wave-uniform branch in front of every fragment)
          *reinterpret_cast<f32x4_t*>(cst + (ii * 16 + fr) * CLF + j * 16 + fc * 4) = v;
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = half * 32 + r * 4 + crow;
      const int m = mw + ib * 16 + row;
      const f32x4_t c = *reinterpret_cast<const f32x4_t*>(cst + row * CLF + cq * 4);
      float dm[4];
      drop_mul4(e.drop, (uint64_t)(m + e.mbase) * (uint64_t)e.N + (uint64_t)n, dm);
      f32x4_t o;
      if constexpr (EPI == GSL_EPI_PATCH) {       // (tok == 0 ? cls : acc + bias) + pos, then dropout
        const bool is_cls = (m % e.T) == 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = ((is_cls ? c4[k] : c[k] + b4[k]) + rs[r][k]) * dm[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (c[k] + b4[k]) * dm[k] + rs[r][k];
      }
      if (m < e.M && n < e.N) store_stream16(out + (size_t)m * e.ldo + n, make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])), GSL_STMODE_OF(e));
    }
  }
}
// BIAS_RES_BF16 / PATCH_BF16 epilogue (bf16 speed mode with the FORWARD residual stream in bf16: out-proj, FFN2 forward, patch
// embedding): x_out = bf16( dropout(acc + bias) + f32(x_in) ) — f32 arithmetic on the f32 accumulator, ONE rounding on store. Same
// staging as the f32-stream epilogue above; the residual is loaded and the result stored as full 128-byte rows (8 lanes x 16 B per
// row, 8 rows per instruction): half the epilogue bytes of the f32 stream. 64 rows per round, the residual rows of round q + 1 are
// requested before round q is processed. The dropout mask is the one of the f32-stream epilogue (same element index, same hash).
// F16 (the stream format) and DROP (a dropout mask applies) are wave-uniform launch arguments, decided ONCE by the dispatcher below: as run-time
// tests inside pack2s / unpack2s / drop_mul4_w they were ~10 uniform branches (each with an s_waitcnt vmcnt(0) behind it) per output row.
// Round 0's residual rows (64 rows x 128 B of the wave's piece) are requested by the KERNEL right after its K loop — in front of the LoRA tail and the
// barrier(s) that separate the K loop from the epilogue — so that their HBM round trip passes under those instead of in front of the first row (the K
// loop's operand-fragment registers are dead by then). Clamped addresses: the piece may be ragged.
struct ResRows { uint4 r[8]; };
__device__ __forceinline__ void res_rows_request0(const EpiArgs& e, ResRows& rs, int mw, int nw, int lane) {
  const int crow = lane >> 3, cch = lane & 7;
  const int ncl = min(nw + cch * 8, e.N - 8);
  const bf16_t* res = reinterpret_cast<const bf16_t*>(e.res);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int m = min(mw + r * 8 + crow, e.M - 1);
    rs.r[r] = gf_ld<uint4>(res + (size_t)m * e.ldo + ncl);
  }
}
// FULL (the wave's 128 x 64 piece lies inside M x N — every tile of the step's shapes): no row predicate, so the rows are straight-line code. With
// the `if (m < M && n < N)` around every row's store each row was its own basic block, and the compiler's wait insertion gave up on counting across
// them: every row opened with s_waitcnt vmcnt(0) — it waited for the NEXT round's eight row requests (issued just before, to fly under this
// round) and for the previous row's store, 16 times per wave.
template <int NI, int EPI, bool F16, bool DROP, bool FULL>
__device__ __forceinline__ void epilogue_staged_res_bf16_impl(const EpiArgs& e, f32x4_t (&acc)[NI][4], float* cst, int mw, int nw, int lane, ResRows& rs0) {
  auto unpack = [](uint32_t u, float& lo, float& hi) {
    if constexpr (F16) unpack2h(u, lo, hi);
    else { lo = __uint_as_float(u << 16); hi = __uint_as_float(u & 0xffff0000u); }
  };
  auto pack = [](float lo, float hi) -> uint32_t { if constexpr (F16) return pack2h(lo, hi); else return pack2bf(lo, hi); };
  const int fr = lane & 15, fc = lane >> 4;
  const int crow = lane >> 3, cch = lane & 7;
  const bf16_t* res = reinterpret_cast<const bf16_t*>(e.res);
  bf16_t* out = reinterpret_cast<bf16_t*>(e.out);
  const int n = nw + cch * 8, ncl = FULL ? n : min(n, e.N - 8);
  float b8[8], c8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { b8[k] = e.bias[ncl + k]; c8[k] = 0.f; }
  if constexpr (EPI == GSL_EPI_PATCH_BF16) {
#pragma unroll
    for (int k = 0; k < 8; ++k) c8[k] = e.cls[ncl + k];
  }
  constexpr int NQ = NI / 4;
  auto fetch = [&](int q, uint4 (&rs)[8]) {
    if constexpr (EPI == GSL_EPI_BIAS_RES_BF16) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int m = FULL ? mw + q * 64 + r * 8 + crow : min(mw + q * 64 + r * 8 + crow, e.M - 1);
        rs[r] = gf_ld<uint4>(res + (size_t)m * e.ldo + ncl);      // the residual stream is read once, by this workgroup (see gf_request)
      }
    }
  };
  uint4 (&rsa)[8] = rs0.r;      // round 0: requested by the kernel (res_rows_request0)
  uint4 rsb[8];
  const uint32_t rowstep = DROP ? (uint32_t)((4u * (uint32_t)e.N) * DROP_PHI) : 0u;      // 8 rows further = 4 N element pairs
  uint32_t w0 = DROP ? drop_w0(e.drop.key, ((uint64_t)(mw + e.mbase + crow) * (uint64_t)e.N + (uint64_t)n) >> 1) : 0u;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    uint4 (&rs)[8] = (q & 1) ? rsb : rsa;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4_t v = acc[q * 4 + ii][j];
        *reinterpret_cast<f32x4_t*>(cst + (ii * 16 + fr) * CLF + j * 16 + fc * 4) = v;      // (alpha is 1 for this epilogue: the host rejects anything else)
      }
    __builtin_amdgcn_sched_barrier(0);      // the next round's rows are requested AFTER this round's 64 accumulator registers are staged (free)
    if (q + 1 < NQ) fetch(q + 1, (q & 1) ? rsa : rsb);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = r * 8 + crow;
      const int m = mw + q * 64 + row;
      const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(cst + row * CLF + cch * 8);
      const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(cst + row * CLF + cch * 8 + 4);
      const float c[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      float dm[8], o[8];
      if constexpr (DROP) {
        drop_mul4_on(e.drop, w0, dm);
        drop_mul4_on(e.drop, w0 + 2u * DROP_PHI, dm + 4);
        w0 += rowstep;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) dm[k] = 1.0f;
      }
      if constexpr (EPI == GSL_EPI_PATCH_BF16) {       // (tok == 0 ? cls : acc + bias) + pos, then dropout (vit_face.py:531-537)
        const int tok = min(m, e.M - 1) % e.T;
        const float* pr = e.pos + (size_t)tok * e.N + ncl;
        const f32x4_t p0 = *reinterpret_cast<const f32x4_t*>(pr), p1 = *reinterpret_cast<const f32x4_t*>(pr + 4);
        const float pp[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = ((tok == 0 ? c8[k] : c[k] + b8[k]) + pp[k]) * dm[k];
      } else {
        const uint32_t a[4] = {rs[r].x, rs[r].y, rs[r].z, rs[r].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float r0, r1;
          unpack(a[k], r0, r1);
          o[2 * k] = (c[2 * k] + b8[2 * k]) * dm[2 * k] + r0;
          o[2 * k + 1] = (c[2 * k + 1] + b8[2 * k + 1]) * dm[2 * k + 1] + r1;
        }
      }
      if (FULL || (m < e.M && n < e.N))
        store_stream16(out + (size_t)m * e.ldo + n, make_uint4(pack(o[0], o[1]), pack(o[2], o[3]), pack(o[4], o[5]), pack(o[6], o[7])), GSL_STMODE_OF(e));
      // rows stay in program order: without the branches that used to separate them the scheduler hoisted the masks and unpacked residuals of
      // several rows above each other, next to 128 accumulators and two sets of prefetched rows — 22 - 29 spilled dwords, +91 MB of scratch traffic
      // per FFN2-forward launch (PMC WRITE_SIZE 232 -> 323 MB) on the in-order memory pipe
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
template <int NI, int EPI>
__device__ __forceinline__ void epilogue_staged_res_bf16(const EpiArgs& e, f32x4_t (&acc)[NI][4], float* cst, int mw, int nw, int lane, ResRows& rs0) {
  const bool full = mw + NI * 16 <= e.M && nw + 64 <= e.N;      // wave-uniform
  auto go = [&](auto f16, auto drop) {
    if (full) epilogue_staged_res_bf16_impl<NI, EPI, decltype(f16)::value, decltype(drop)::value, true>(e, acc, cst, mw, nw, lane, rs0);
    else epilogue_staged_res_bf16_impl<NI, EPI, decltype(f16)::value, decltype(drop)::value, false>(e, acc, cst, mw, nw, lane, rs0);
  };
  if (e.f16) {
    if (e.drop.thr) go(std::true_type{}, std::true_type{});
    else go(std::true_type{}, std::false_type{});
  } else {
    if (e.drop.thr) go(std::false_type{}, std::true_type{});
    else go(std::false_type{}, std::false_type{});
  }
}
constexpr int CST_WAVE = 2 * 64 * CLD;            // bf16 elements of staging per wave (two outputs)
constexpr int CST_BLOCK8 = 8 * CST_WAVE;          // 8 waves: 147 456 bytes
constexpr int CST_WAVE_G8 = 64 * CLD + 64 * 80 / 2;  // BIAS_GELU_G8: 64 bf16 rows (144 B) + 64 code rows (80 B) = 14 336 bytes per wave

// bijective XCD-aware remap: hardware places block b on XCD b % 8; give each XCD a contiguous
// range of logical tile ids so neighbouring tiles (same A row-panel) share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// ------------------------------------------------------------------ bf16 MFMA kernel, direct-to-LDS staging
constexpr int BM = 128, BN = 128, BK = 64;
// Same tile / fragment / epilogue structure, but the global->LDS copy is the gfx950 LDS-DMA
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass). The DMA destination is
// wave-uniform base + lane*16 B, so the XOR swizzle is applied on the SOURCE address (lane -> (row, cpos)
// reads global chunk cpos ^ (row & 7)) and again on the fragment read — the same involution.
// NBUF = 1: one 32 KB stage, two barriers per K tile, relies on >= 3 resident blocks per CU for overlap.
// NBUF = 2: 64 KB, next tile's DMA in flight during the MFMAs, one barrier per K tile.
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int EPI, int NBUF>
__global__ __launch_bounds__(256) void gemm_bf16_glds_kernel(const bf16_t* __restrict__ A1, int lda1,
                                                             const bf16_t* __restrict__ W1, int ldw1, int K1,
                                                             const bf16_t* __restrict__ A2, int lda2,
                                                             const bf16_t* __restrict__ W2, int ldw2, int K2, EpiArgs e) {
  GSL_OP16_KERNEL_ENTRY();
  resolve_drop(e.drop);
  __shared__ __attribute__((aligned(16))) bf16_t smem[NBUF][2][BM * BK];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (e.N + BN - 1) / BN;
  const int tile = e.remap ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
  const int nk1 = K1 / BK, nk = nk1 + K2 / BK;
  const int lrow = lane >> 3, lc = lane & 7;

  auto issue = [&](int kt, int buf) {
    const bf16_t* Ab; const bf16_t* Wb; int lda, ldw, k0;
    if (kt < nk1) { Ab = A1; Wb = W1; lda = lda1; ldw = ldw1; k0 = kt * BK; }
    else { Ab = A2; Wb = W2; lda = lda2; ldw = ldw2; k0 = (kt - nk1) * BK; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rb = wave * 4 + i;                 // 8-row block: one 1 KB DMA per wave-instruction
      const int row = rb * 8 + lrow;
      const int c = lc ^ (row & 7);
      const int gm = min(m0 + row, e.M - 1), gn = min(n0 + row, e.N - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Ab + (size_t)gm * lda + k0 + c * 8), (lptr_t)(&smem[buf][0][rb * 8 * BK]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)gn * ldw + k0 + c * 8), (lptr_t)(&smem[buf][1][rb * 8 * BK]), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fc = lane >> 4;

  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + fr;
        af[i] = *reinterpret_cast<const bf16x8_t*>(&smem[buf][0][row * BK + (((ks * 4 + fc) ^ (row & 7)) << 3)]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        wf[j] = *reinterpret_cast<const bf16x8_t*>(&smem[buf][1][row * BK + (((ks * 4 + fc) ^ (row & 7)) << 3)]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = GSL_MFMA16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  if constexpr (NBUF == 1) {
    for (int kt = 0; kt < nk; ++kt) {
      issue(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0);
      __syncthreads();
    }
  } else {
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kt has landed (only its DMAs are outstanding here)
      __syncthreads();                                     // ... for every wave; and compute(kt-1) is finished everywhere
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
      compute(kt & 1);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epilogue4<EPI, bf16_t>(e, m0 + wm * 64 + i * 16 + fr, n0 + wn * 64 + j * 16 + fc * 4, v);
    }
}

// operands of the in-kernel LoRA forms: P [16, K] (rows j < r = the adapter's down-projection), Q [N, >= 32] (columns j < r = its
// up-projection), t = s * A P^T is written to tout [M, ldt >= 64] (zero padded) by the N-tile 0 workgroups
#ifndef GSL_SMALL_KSPLIT
#define GSL_SMALL_KSPLIT 1      // ring kernel: two wave groups split the K tiles when K >= GSL_SMALL_KSPLIT_MINK and the grid is at most one workgroup per CU
#endif
#ifndef GSL_SMALL_KSPLIT_MINK
#define GSL_SMALL_KSPLIT_MINK 1024      // (512 is 1.4 % faster on the few-shot step, but FFN1 (N = 2048, K = 512) would then split at 512 rows and not at 1024: the fused two-batch forward must stay bit-identical to two forwards, tests/test_hip_fullsize.py)
#endif
#ifndef GSL_SMALL_WIDE
#define GSL_SMALL_WIDE 1      // 64x128 tiles on the ring kernel when the 64x64 grid exceeds the resident workgroups (0: never)
#endif
#ifndef GSL_SMALL_NSTS
#define GSL_SMALL_NSTS 3      // stages of the 64x64 ring kernel: 3 x 16 KB = three workgroups per CU (4: two; measured, r03_notes.md)
#endif
// Workgroup barrier that PUBLISHES this wave's LDS stores: s_barrier alone only lines the waves up — a ds_write issued in front of it may still be
// in the LDS queue when another wave, past the barrier, reads the location (gfx950 has the back-off barrier: the compiler inserts no wait in front of a
// raw s_barrier, and the raw builtin carries no fence). The K loops use the raw barrier on purpose (their LDS-DMA stream must stay in flight; what they
// publish is retired by a counted vmcnt wait); every hand-over through plain LDS stores uses this one (or __syncthreads()).
__device__ __forceinline__ void wg_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct LoraInk {
  const bf16_t* P; int ldp;
  const bf16_t* Q; int ldq;
  float s;
  bf16_t* tout; int ldt;
};

// ------------------------------------------------------------------ bf16 MFMA kernel, 64x64 tile, 4-stage LDS-DMA ring: few rows
// The launch-bound regime (few-shot batches: M = 1 576 rows; the cls-row tail of the last block: M = batch) has too few 128x128 tiles
// to occupy the chip — M = 1 576, N = 512 are 52 workgroups, and their K = 2048 loop then runs tile after tile with two barriers
// each: 32 us where the vendor library needs 11 (tools/probes/small_m_gemm.py). Here: 64x64 tiles (4x the workgroups), four waves
// (2 x 2, 32x32 each), BK = 64, a ring of three 16 KB stages (three workgroups per CU) with the LDS-DMA running two K tiles ahead, counted vmcnt and ONE raw
// barrier per K tile. Same swizzle / fragment layout / epilogues as the kernels above (fragment-path stores: the outputs are small).
// LORA = true: the in-kernel LoRA form of the 8-phase kernel on this tile — out = epilogue(A W^T + t Q^T) with t = s * A P^T computed
// here: the 16 rows of P ride along in every stage (one more DMA instruction for waves 0 and 1), wave (wm, wn) owns the row fragment
// wm * 32 + wn * 16 of t (two extra MFMAs per K tile), and the rank-r update is one more k-step from LDS at the end. In the launch-bound
// regime this removes the separate skinny GEMM (K = 2048: 12 us on 25 workgroups) per adapted layer and direction.
// NJ = column fragments per wave: 2 = 64x64 tiles, 4 = 64x128 tiles (wide N: M = 1 576, N = 2048 are 800 tiles of 64x64 on 768 resident
// workgroups — 32 of them start a second round; 400 tiles of 64x128 fit one).
constexpr int BMS = 64, BNS = 64, NSTS = GSL_SMALL_NSTS;
#ifndef GSL_SMALL_SLOTS
#define GSL_SMALL_SLOTS (256L * (NSTS <= 3 ? 3 : 2))      // resident 64x64 workgroups on the chip
#endif
constexpr long SMALL_SLOTS = GSL_SMALL_WIDE ? GSL_SMALL_SLOTS : (1L << 40);
// KS = 2: TWO groups of four waves split the K tiles of one output tile between them (group g takes tiles g, g + 2, ...: own stage ring,
// own accumulators, same barriers) and group 1's accumulators are added to group 0's through LDS at the end, in that fixed order. With
// at most one workgroup per CU anyway (M = 1 576, N = 512: 200 tiles) the K = 1536 / 2048 loops are a serial latency chain of 24 / 32
// tile steps: this halves it.
template <int EPI, bool LORA = false, int NJ = 2, int KS = 1>
__global__ __launch_bounds__(256 * KS) void gemm_bf16_small_kernel(const bf16_t* __restrict__ A1, int lda1,
                                                              const bf16_t* __restrict__ W1, int ldw1, int K1,
                                                              const bf16_t* __restrict__ A2, int lda2,
                                                              const bf16_t* __restrict__ W2, int ldw2, int K2, LoraInk lk, EpiArgs e) {
  GSL_OP16_KERNEL_ENTRY();
  resolve_drop(e.drop);
  constexpr int BNT = 32 * NJ;                                   // tile columns
  constexpr int STS = (BMS + BNT + (LORA ? 16 : 0)) * BK;
  __shared__ __attribute__((aligned(16))) bf16_t smem[KS * NSTS * STS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KS == 2 ? wave8 >> 2 : 0, wave = wave8 & 3;      // K group, wave inside the group
  bf16_t* const gsm = smem + grp * (NSTS * STS);                   // this group's stage ring
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (e.N + BNT - 1) / BNT;
  const int tile = e.remap ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int m0 = (tile / nbn) * BMS, n0 = (tile % nbn) * BNT;
  const int nk1 = K1 / BK, nk = nk1 + K2 / BK;
  const int nkg = (nk - grp + KS - 1) / KS, nit = (nk + KS - 1) / KS;      // K tiles of this group, loop steps of the workgroup
  const int lrow = lane >> 3, lc = lane & 7;

  auto issue = [&](int it) {      // step `it` of this group = K tile it * KS + grp; 2 + NJ DMA instructions per wave
    if (it >= nkg) return;
    const int kt = it * KS + grp;
    bf16_t* st = gsm + (it % NSTS) * STS;
    const bf16_t* Ab; const bf16_t* Wb; int lda, ldw, k0;
    if (kt < nk1) { Ab = A1; Wb = W1; lda = lda1; ldw = ldw1; k0 = kt * BK; }
    else { Ab = A2; Wb = W2; lda = lda2; ldw = ldw2; k0 = (kt - nk1) * BK; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rb = wave * 2 + i, row = rb * 8 + lrow, c = lc ^ (row & 7);
      const int gm = min(m0 + row, e.M - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Ab + (size_t)gm * lda + k0 + c * 8), (lptr_t)(st + rb * 8 * BK), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int rb = wave * NJ + i, row = rb * 8 + lrow, c = lc ^ (row & 7);
      const int gn = min(n0 + row, e.N - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)gn * ldw + k0 + c * 8), (lptr_t)(st + BMS * BK + rb * 8 * BK), 16, 0, 0);
    }
    if constexpr (LORA) {
      if (wave < 2) {
        const int row = wave * 8 + lrow, c = lc ^ (row & 7);
        __builtin_amdgcn_global_load_lds((gptr_t)(lk.P + (size_t)row * lk.ldp + k0 + c * 8),
                                         (lptr_t)(st + (BMS + BNT) * BK + wave * 8 * BK), 16, 0, 0);
      }
    }
  };

  f32x4_t acc[2][NJ];
  f32x4_t accp = f32x4_t{0.f, 0.f, 0.f, 0.f};      // LORA: t[row wm*32 + wn*16 + fr][j = fc*4 + reg]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fc = lane >> 4;
  int aoff[2][2], boff[NJ][2];      // fragment read offsets inside a stage (bf16 elements), per k-step
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 32 + i * 16 + fr;
      aoff[i][ks] = ra * BK + (((ks * 4 + fc) ^ (ra & 7)) << 3);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int rb = wn * 16 * NJ + j * 16 + fr;
      boff[j][ks] = BMS * BK + rb * BK + (((ks * 4 + fc) ^ (rb & 7)) << 3);
    }
  }

  issue(0);
  issue(1);
  if (NSTS > 3) issue(2);
  for (int kt = 0; kt < nit; ++kt) {      // (kt counts this group's steps)
    const bool live = kt < nkg;             // the last step of an odd tile count belongs to group 0 alone: group 1 only keeps the barrier
    const int ahead = min(NSTS - 2, nkg - 1 - kt);      // K tiles that may still be in flight behind this step's tile
    // counted wait: 2 + NJ DMA instructions per wave and K tile (+ 1 for the two waves that also fetch P)
#define GSL_SW(N2, N1) { if (ahead >= 2) asm volatile("s_waitcnt vmcnt(" #N2 ")" ::: "memory"); else if (ahead == 1) asm volatile("s_waitcnt vmcnt(" #N1 ")" ::: "memory"); \
                         else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    if (live) {
      if constexpr (NJ == 2) { if (LORA && wave < 2) GSL_SW(10, 5) else GSL_SW(8, 4) }
      else { if (LORA && wave < 2) GSL_SW(14, 7) else GSL_SW(12, 6) }
    }
#undef GSL_SW
    __builtin_amdgcn_s_barrier();      // every wave's share of tile kt is in LDS; compute(kt - 1) finished everywhere
    issue(kt + NSTS - 1);    // overwrites the stage of step kt - 1
    if (!live) continue;
    const bf16_t* st = gsm + (kt % NSTS) * STS;
    bf16x8_t af[2][2], wf[NJ][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i][ks] = *reinterpret_cast<const bf16x8_t*>(st + aoff[i][ks]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) wf[j][ks] = *reinterpret_cast<const bf16x8_t*>(st + boff[j][ks]);
    }
    if constexpr (LORA) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // the wave's own row fragment of A, read again from LDS: selecting af[wn] at run time would index a register array (scratch)
        const int rt = wm * 32 + wn * 16 + fr;
        const bf16x8_t ta = *reinterpret_cast<const bf16x8_t*>(st + rt * BK + (((ks * 4 + fc) ^ (rt & 7)) << 3));
        const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(st + (BMS + BNT) * BK + fr * BK + (((ks * 4 + fc) ^ (fr & 7)) << 3));
        accp = GSL_MFMA16(pf, ta, accp, 0, 0, 0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = GSL_MFMA16(wf[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
  }
  if constexpr (KS == 2) {      // group 1 hands its accumulators to group 0: f32, [wave][fragment][lane] behind the t buffer, fixed order 0 + 1
    __builtin_amdgcn_s_barrier();                      // the stages are free
    float* red = reinterpret_cast<float*>(smem + 64 * 32);
    float* mine = red + (wave * (2 * NJ + 1)) * 256 + lane * 4;
    if (grp == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) *reinterpret_cast<float4*>(mine + (i * NJ + j) * 256) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if constexpr (LORA) *reinterpret_cast<float4*>(mine + 2 * NJ * 256) = make_float4(accp[0], accp[1], accp[2], accp[3]);
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(mine + (i * NJ + j) * 256);
          acc[i][j][0] += v.x; acc[i][j][1] += v.y; acc[i][j][2] += v.z; acc[i][j][3] += v.w;
        }
      if constexpr (LORA) {
        const float4 v = *reinterpret_cast<const float4*>(mine + 2 * NJ * 256);
        accp[0] += v.x; accp[1] += v.y; accp[2] += v.z; accp[3] += v.w;
      }
    }
  }
  const bool lead = grp == 0;      // the group that finishes the tile (barriers below are still taken by every wave)
  if constexpr (LORA) {
    // t -> LDS [64][32] bf16 (columns 16..31 zero), stored once (N-tile 0) for the gradient reductions, then the rank-r update
    __builtin_amdgcn_s_barrier();                      // the stages are free
    bf16_t* tbuf = smem;
    if (lead) {
      bf16_t* d = tbuf + (wm * 32 + wn * 16 + fr) * 32 + fc * 4;
      *reinterpret_cast<uint2*>(d) = make_uint2(pack2o(lk.s * accp[0], lk.s * accp[1]), pack2o(lk.s * accp[2], lk.s * accp[3]));
      *reinterpret_cast<uint2*>(d + 16) = make_uint2(0u, 0u);
    }
    wg_barrier_lds();      // (a raw s_barrier here let a wave read t rows whose ds_write was still queued: a wrong 16-row fragment once in ~10 process runs)
    if (lead && n0 == 0 && lk.tout) {                  // 64 rows x 64 columns = 512 16-byte pieces, two per thread of group 0
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int piece = tid * 2 + q, row = piece >> 3, c = piece & 7;
        if (m0 + row < e.M) {
          const uint4 v = c < 2 ? *reinterpret_cast<const uint4*>(tbuf + row * 32 + c * 8) : make_uint4(0u, 0u, 0u, 0u);
          *reinterpret_cast<uint4*>(lk.tout + (size_t)(m0 + row) * lk.ldt + c * 8) = v;
        }
      }
    }
    if (lead) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = min(n0 + wn * 16 * NJ + j * 16 + fr, e.N - 1);
      const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(lk.Q + (size_t)n * lk.ldq + fc * 8);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8_t tf = *reinterpret_cast<const bf16x8_t*>(tbuf + (wm * 32 + i * 16 + fr) * 32 + fc * 8);
        acc[i][j] = GSL_MFMA16(qf, tf, acc[i][j], 0, 0, 0);
      }
    }
    }
  }
  if (!lead) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epilogue4<EPI, bf16_t>(e, m0 + wm * 32 + i * 16 + fr, n0 + wn * 16 * NJ + j * 16 + fc * 4, v);
    }
}

// ------------------------------------------------------------------ bf16 MFMA kernel, 256x128 tile, 3-stage LDS-DMA ring
// 8 waves (4 x 2) of 64x64, BK = 64, three 48 KB stages (144 KB of the CU's 160 KB LDS): the DMA of tile
// kt+2 is issued while tile kt is multiplied and tile kt+1 is still in flight. Waits are COUNTED
// (s_waitcnt vmcnt(6): each wave has 6 DMA instructions per tile) and the barrier is the raw s_barrier,
// because __syncthreads() would drain the in-flight stage (vmcnt(0)). One barrier per K tile.
// L2->LDS traffic per flop is 25 % lower than the 128x128 tile's.
constexpr int BM3 = 256, BN3 = 128, ST3 = (BM3 + BN3) * BK;   // elements per stage

// ABL (development only): bit0 skip DMA, bit1 skip LDS fragment reads, bit2 skip MFMA, bit3 skip barrier
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_bf16_ring3_kernel(const bf16_t* __restrict__ A1, int lda1,
                                                              const bf16_t* __restrict__ W1, int ldw1, int K1,
                                                              const bf16_t* __restrict__ A2, int lda2,
                                                              const bf16_t* __restrict__ W2, int ldw2, int K2, EpiArgs e) {
  GSL_OP16_KERNEL_ENTRY();
  resolve_drop(e.drop);
  __shared__ __attribute__((aligned(16))) bf16_t smem[3 * ST3];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (e.N + BN3 - 1) / BN3;
  const int tile = e.remap ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int m0 = (tile / nbn) * BM3, n0 = (tile % nbn) * BN3;
  const int nk1 = K1 / BK, nk = nk1 + K2 / BK;
  const int lrow = lane >> 3, lc = lane & 7;

  auto issue = [&](int kt) {
    bf16_t* st = smem + (kt % 3) * ST3;
    const bf16_t* Ab; const bf16_t* Wb; int lda, ldw, k0;
    if (kt < nk1) { Ab = A1; Wb = W1; lda = lda1; ldw = ldw1; k0 = kt * BK; }
    else { Ab = A2; Wb = W2; lda = lda2; ldw = ldw2; k0 = (kt - nk1) * BK; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rb = wave * 4 + i, row = rb * 8 + lrow, c = lc ^ (row & 7);
      const int gm = min(m0 + row, e.M - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Ab + (size_t)gm * lda + k0 + c * 8), (lptr_t)(st + rb * 8 * BK), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rb = wave * 2 + i, row = rb * 8 + lrow, c = lc ^ (row & 7);
      const int gn = min(n0 + row, e.N - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)gn * ldw + k0 + c * 8), (lptr_t)(st + BM3 * BK + rb * 8 * BK), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fc = lane >> 4;

  if (!(ABL & 1)) { issue(0); if (nk > 1) issue(1); }
  bf16x8_t af[4], wf[4];
  if (ABL & 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i] = *reinterpret_cast<const bf16x8_t*>(smem + (wm * 64 + i * 16 + fr) * BK + fc * 8); wf[i] = af[i]; }
  }
  for (int kt = 0; kt < nk; ++kt) {
    if (!(ABL & 1)) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // tile kt landed, tile kt+1 may still fly
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!(ABL & 8)) __builtin_amdgcn_s_barrier();   // every wave's share of tile kt is in LDS; compute(kt-1) finished everywhere
    if (!(ABL & 1)) { if (kt + 2 < nk) issue(kt + 2); }   // overwrites the stage of tile kt-1
    const bf16_t* As = smem + (kt % 3) * ST3;
    const bf16_t* Ws = As + BM3 * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (!(ABL & 2)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wm * 64 + i * 16 + fr;
          af[i] = *reinterpret_cast<const bf16x8_t*>(As + row * BK + (((ks * 4 + fc) ^ (row & 7)) << 3));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = wn * 64 + j * 16 + fr;
          wf[j] = *reinterpret_cast<const bf16x8_t*>(Ws + row * BK + (((ks * 4 + fc) ^ (row & 7)) << 3));
        }
      }
      if (!(ABL & 4)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = GSL_MFMA16(wf[j], af[i], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" :: "v"(af[i]), "v"(wf[i])); }
      }
    }
  }
  if constexpr (EPI == GSL_EPI_STORE) {
    // STORE with out2: a COMPACT [M, 16] copy (row stride 16) of output columns 0 .. 15 — the LoRA down-projection u1 is a [M, 64] K segment
    // for the forward GEMM and a 16-column operand for the gradient-fused FFN2-dX epilogue, which then reads 32 contiguous rows with one
    // 1 KB instruction instead of 32 rows x 32 B of 128-byte lines (FFN2-dX 0.80 -> 0.78 ms). A lane's 4 columns of 16 rows: 512 contiguous bytes.
    if (e.out2 && n0 == 0 && wn == 0) {
      bf16_t* o2 = reinterpret_cast<bf16_t*>(e.out2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        const f32x4_t v = acc[i][0] * e.alpha;
        if (m < e.M) *reinterpret_cast<uint2*>(o2 + (size_t)m * 16 + fc * 4) = make_uint2(pack2o(v[0], v[1]), pack2o(v[2], v[3]));
      }
    }
  }
  if constexpr (!epi_out_is_f32<EPI>()) {
    static_assert(3 * ST3 >= CST_BLOCK8, "C staging must fit in the stage ring");
    if (ABL == 0 && (e.N % 8) == 0 && (e.ldo % 8) == 0 && (EPI != GSL_EPI_BIAS_GELU_G8 || ((e.N % 16) == 0 && (e.ldo % 16) == 0))) {
      __builtin_amdgcn_s_barrier();            // every wave is done with the stage ring: reuse it for C staging
      epilogue_staged_bf16<EPI, 4>(e, acc, smem + wave * CST_WAVE, m0 + wm * 64, n0 + wn * 64, lane);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epilogue4<EPI, bf16_t>(e, m0 + wm * 64 + i * 16 + fr, n0 + wn * 64 + j * 16 + fc * 4, v);
    }
}

// 256x256 output tile of the 8-phase kernel (and of the development variants in gemm_dev_*.inc)
constexpr int BM4 = 256, BN4 = 256, ST4 = (BM4 + BN4) * BK;
#ifdef GSL_DEV
#include "gemm_dev_a.inc"
#endif

constexpr int ST4L = (BM4 + BN4 + 16) * BK;

// ------------------------------------------------------------------ bf16 MFMA kernel, 256x256 tile, 8-phase ping-pong schedule
// Same tile, wave grid (2 x 4, 128x64 per wave) and epilogue as gemm_bf16_t256_kernel; the K loop is re-scheduled:
//  * a K tile (BK = 64) is computed in 4 phases = the 4 quadrants (64 rows x 32 cols x K 64 = 16 MFMAs) of the wave's 128x64 tile;
//  * the LDS stage of a K tile is 4 half-tiles of 16 KB: B-h0 / B-h1 hold the ch = 0 / 1 column halves of all 4 wave columns,
//    A-h0 / A-h1 the rh = 0 / 1 row halves of both wave rows, so that a half-tile is read in exactly one phase
//    (q0: B-h0 + A-h0, q1: B-h1, q2: A-h1, q3: nothing) and can be re-staged for K tile t+2 right after: one LDS-DMA half-tile is
//    issued per phase, the stream runs 7 half-tiles ahead, and the counted wait (vmcnt(6), once per K tile, in q3) leaves three
//    half-tiles in flight;
//  * the two wave rows run staggered by one barrier (wm == 1 takes one extra s_barrier up front): while one group issues its
//    16 MFMAs at raised priority the other one issues ds_reads and the LDS-DMA — the matrix pipe and the LDS/TA pipes overlap.
// Hazards (stagger included): a half-tile staged in phase p is read in phase >= p+4; the wait that retires it sits before the
// first barrier of the phase before the read; a half-tile is re-staged >= 2 phases after its last ds_read, or 1 phase after for
// B-h0 whose reads are retired (lgkmcnt(8)) before the first barrier of q0.
// LORA = true: the in-kernel LoRA form of gemm_bf16_t256_lora_kernel (below) on this schedule — the 16 rows of P ride along
// with piece B-h0 (one extra DMA for waves 0 and 1, so their counted wait is vmcnt(7)), the P fragment is read in q0 next to B-h0,
// and a wave issues its 4 extra MFMAs (2 of its row-half's 8 row fragments x 2 k-steps) in q0 (wn < 2) or q2 (wn >= 2).
// GRAD = true (MUL epilogue + LORA only): the LoRA-gradient reductions of the tile are fused into the epilogue (epilogue_staged_mulgrad);
// t is then kept as [256][16] behind the staging regions so that it survives the epilogue.
template <int EPI, bool LORA, bool GRAD = false>
__global__ __launch_bounds__(512) void gemm_bf16_p8_kernel(const bf16_t* __restrict__ A1, int lda1,
                                                           const bf16_t* __restrict__ W1, int ldw1, int K1,
                                                           const bf16_t* __restrict__ A2, int lda2,
                                                           const bf16_t* __restrict__ W2, int ldw2, int K2, LoraInk lk, EpiArgs e) {
  GSL_OP16_KERNEL_ENTRY();
  resolve_drop(e.drop);
  constexpr int STG = LORA ? ST4L : ST4;
  static_assert(!GRAD || (LORA && epi_is_mul<EPI>()), "GRAD is the gradient-fused form of the in-kernel-LoRA MUL GEMM");
  static_assert(!GRAD || 2 * STG * 2 <= 8 * GF_WAVE_B, "t16 must lie behind the K-loop stages");
  // BIAS_GELU_G8: the GELU table (GT_N x 4 B) sits behind the K-loop stages, and the C staging of this epilogue needs 64 x (144 + 80) B per
  // wave only (bf16 rows + code rows), so the block does not grow: 2 x 64 KB of stages + 16 KB of table = the 144 KB of the other epilogues
  constexpr bool TAB = (EPI == GSL_EPI_BIAS_GELU_G8);
  constexpr int CSTW = TAB ? CST_WAVE_G8 : CST_WAVE;
  constexpr int SMEM_E = GRAD ? GF_BLOCK_B / 2 : (TAB ? 2 * STG + GT_N * 2 : ((2 * STG > CST_BLOCK8) ? 2 * STG : CST_BLOCK8));
  static_assert(!TAB || 8 * CST_WAVE_G8 <= 2 * STG, "the G8 staging must end before the table");
  __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM_E];   // stages, then C staging
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nbn = (e.N + BN4 - 1) / BN4;
  const int tile0 = e.remap ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tile = e.mrev ? (int)gridDim.x - 1 - tile0 : tile0;
  const int m0 = (tile / nbn) * BM4, n0 = (tile % nbn) * BN4;
  const int nk1 = K1 / BK, nk = nk1 + K2 / BK;
  const int krot = e.krot ? (tile % nbn) % nk : 0;
  const int lrow = lane >> 3, lc = lane & 7;
  constexpr int HT = 128 * BK;     // half-tile, bf16 elements

  // per-lane source rows of the two DMA instructions a lane contributes to an A / B half-tile (half added at issue time)
  int arow[2], brow[2], csw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + lrow;
    arow[i] = m0 + (r >> 6) * 128 + (r & 63);
    brow[i] = n0 + (r >> 5) * 64 + (r & 31);
    csw[i] = (lc ^ (r & 7)) * 8;
  }
  // PIECE: 0 = B-h0, 1 = A-h0, 2 = B-h1, 3 = A-h1 (stream order within a K tile)
  auto stage = [&](int kt, auto piece_c) {
    constexpr int PIECE = decltype(piece_c)::value;
    if (kt >= nk) return;
    constexpr bool isA = PIECE & 1;
    constexpr int half = PIECE >> 1;
    bf16_t* dst = smem + (kt & 1) * STG + (isA ? 0 : BM4 * BK) + half * HT;
    int kk = kt + krot;                 // K tile actually fetched in loop step kt
    if (kk >= nk) kk -= nk;
    const bf16_t* base; int ld, k0;
    if (kk < nk1) { base = isA ? A1 : W1; ld = isA ? lda1 : ldw1; k0 = kk * BK; }
    else { base = isA ? A2 : W2; ld = isA ? lda2 : ldw2; k0 = (kk - nk1) * BK; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = isA ? min(arow[i] + half * 64, e.M - 1) : min(brow[i] + half * 32, e.N - 1);
      // (cache-policy bits on the two operand streams — sc0 / nt / sc1 — were measured neutral to negative: profiles/r03_notes.md)
      if constexpr (isA) __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)g * ld + k0 + csw[i]), (lptr_t)(dst + (wave * 2 + i) * 8 * BK), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)g * ld + k0 + csw[i]), (lptr_t)(dst + (wave * 2 + i) * 8 * BK), 16, 0, 0);
    }
    if constexpr (LORA && PIECE == 0) {
      if (wave < 2) {
        const int row = wave * 8 + lrow, c = lc ^ (row & 7);
        __builtin_amdgcn_global_load_lds((gptr_t)(lk.P + (size_t)row * lk.ldp + k0 + c * 8),
                                         (lptr_t)(smem + (kt & 1) * STG + (BM4 + BN4) * BK + wave * 8 * BK), 16, 0, 0);
      }
    }
  };
  using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>; using P3 = std::integral_constant<int, 3>;

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fc = lane >> 4;
  // fragment read offsets inside a half-tile (bf16 elements): A rows wm*64 + i*16 + fr, B rows wn*32 + j*16 + fr; chunk (ks*4+fc)^(row&7)
  int aoff[4][2], boff[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int row = wm * 64 + i * 16 + fr; aoff[i][ks] = row * BK + (((ks * 4 + fc) ^ (row & 7)) << 3); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int row = wn * 32 + j * 16 + fr; boff[j][ks] = row * BK + (((ks * 4 + fc) ^ (row & 7)) << 3); }
  }

  // development: cycle stamps (kernel start, prologue landed, K loop done, epilogue done) of every 64th workgroup
#ifdef GSL_DEV
  unsigned long long* dbg8 = (e.stamps && blockIdx.x < 64 * 256 && (blockIdx.x % 64) == 0 && tid == 0) ? e.stamps + (blockIdx.x / 64) * 4 : nullptr;
  // GSL_P8_STAMPS_ALL: every workgroup records (buffer of gridDim.x x 4 u64), the start stamp carries XCC_ID / HW_ID in its top 16 bits
  if (e.stamps && e.stamps_all) dbg8 = tid == 0 ? e.stamps + (size_t)blockIdx.x * 4 : nullptr;
  if (dbg8 && e.stamps_all) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long id = ((unsigned long long)(xcc & 15u) << 12) | ((hw >> 8) & 0xfffu);   // cu_id[11:8], sh_id[12], se_id[15:13]...
    dbg8[0] = (__builtin_readcyclecounter() & 0xffffffffffffull) | (id << 48);
  } else
#else
  constexpr unsigned long long* dbg8 = nullptr;
#endif
  if (dbg8) dbg8[0] = __builtin_readcyclecounter();
  // Every kernel argument the K loop's requests read is consumed once HERE: the compiler loads kernel arguments lazily (s_load) and waits for them at
  // their first use — inside the loop that was an s_waitcnt lgkmcnt(0) in phase q0 of every K tile, i.e. a wait for the eight fragment reads just issued
  // in front of the LDS-DMA request (scalar loads and LDS reads share the counter).
  asm volatile("" ::"s"(A1), "s"(W1), "s"(A2), "s"(W2), "s"(lda1), "s"(ldw1), "s"(lda2), "s"(ldw2), "s"(K1), "s"(K2), "s"(e.M), "s"(e.N));
  // prologue: K tile 0 complete + three half-tiles of K tile 1 in flight
  stage(0, P0{}); stage(0, P1{}); stage(0, P2{}); stage(0, P3{});
  stage(1, P0{}); stage(1, P1{}); stage(1, P2{});
  // (one 16 KB half-tile request per phase, 2 DMA instructions per wave; batching a K tile's requests is neutral in this loop although the
  //  stream ALONE runs faster that way: profiles/r04_g_hybrid_stream.md)
  constexpr int YOUNG = 6;          // DMA instructions of a wave younger than K tile kt+1's last piece at the counted wait
  if constexpr (TAB) {      // the GELU table: 16 pieces of 1 KB, two per wave, BEHIND the prologue's requests (K tile 0 is not delayed by it); the
#pragma unroll             // counted wait below leaves them in flight, the K loop's first counted wait (step 0, q3) retires them in order
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i;
      __builtin_amdgcn_global_load_lds((gptr_t)(GELU_G8_TAB + piece * 256 + lane * 4), (lptr_t)(smem + 2 * STG + piece * 512), 16, 0, 0);
    }
  }
  constexpr int TABP = TAB ? 2 : 0;
  if (nk < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (LORA && wave < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG + 1 + TABP) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG + TABP) : "memory");
  __builtin_amdgcn_s_barrier();
  if (dbg8) dbg8[1] = __builtin_readcyclecounter();
  if (wm == 1) __builtin_amdgcn_s_barrier();     // stagger the second wave row by one barrier

  bf16x8_t af[4][2], bf0[2][2], bf1[2][2], pf[2];
  f32x4_t accp[2];
  accp[0] = accp[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int poff0 = fr * BK + (((0 * 4 + fc) ^ (fr & 7)) << 3), poff1 = fr * BK + (((1 * 4 + fc) ^ (fr & 7)) << 3);
  const int pi = (wn & 1) * 2;      // this wave's two row fragments (within its row half wn >> 1) of the 16 extra columns
  // In-kernel LoRA: the 16 extra output columns t = A P^T need 16 MFMAs per wave row and K tile (8 row fragments x 2 k-steps), shared by
  // the row's 4 waves: a wave issues its 4 extra MFMAs in ONE phase (20 + 16 + 20 + 16 on the barrier-paced path). One extra MFMA in every
  // phase instead was measured slower (profiles/r03_notes.md; the wave-uniform branch ladder costs more than the uneven phases).
#define GSL_P8_PEXTRA(PH)                                                                                   \
  if constexpr (LORA && ((PH) == 0 || (PH) == 2)) {                                                         \
    if ((wn >> 1) == ((PH) >> 1)) {                                                                         \
      if (pi) {                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                  \
          accp[0] = GSL_MFMA16(pf[ks], af[2][ks], accp[0], 0, 0, 0);           \
          accp[1] = GSL_MFMA16(pf[ks], af[3][ks], accp[1], 0, 0, 0);           \
        }                                                                                                   \
      } else {                                                                                              \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                  \
          accp[0] = GSL_MFMA16(pf[ks], af[0][ks], accp[0], 0, 0, 0);           \
          accp[1] = GSL_MFMA16(pf[ks], af[1][ks], accp[1], 0, 0, 0);           \
        }                                                                                                   \
      }                                                                                                     \
    }                                                                                                       \
  }
  // row fragment (of the wave row's 8) that accp[t] holds
#define GSL_P8_TFRAG(t) (2 * wn + (t))
#define GSL_P8_MFMA(RH, CH, BF, PH)                                                                          \
  __builtin_amdgcn_s_barrier();                                                                             \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
  __builtin_amdgcn_sched_barrier(0);                                                                        \
  __builtin_amdgcn_s_setprio(1);                                                                            \
  GSL_P8_PEXTRA(PH)                                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
        acc[(RH) * 4 + i][(CH) * 2 + j] = GSL_MFMA16(BF[j][ks], af[i][ks], acc[(RH) * 4 + i][(CH) * 2 + j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                        \
  __builtin_amdgcn_s_barrier();                                                                             \
  __builtin_amdgcn_sched_barrier(0);

  // (an L2 warm-up for the NEXT round's first two K tiles, issued from inside this loop, was measured useless: profiles/r03_notes.md; the code left with commit 15030e9)
  // (a 4-phase cut of this loop — two phases of 32 MFMAs per K tile, half the barriers — is bit-identical and 0.7 % slower per step:
  //  profiles/r04_notes.md; commit 52d4782 has the code)
  for (int kt = 0; kt < nk; ++kt) {
    const bf16_t* As0 = smem + (kt & 1) * STG;
    const bf16_t* As1 = As0 + HT;
    const bf16_t* Bs0 = As0 + BM4 * BK;
    const bf16_t* Bs1 = Bs0 + HT;
    // ---- q0: quadrant (rh0, ch0); reads B-h0 (first, retired before the barrier) and A-h0; stages A-h1(kt+1)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) bf0[j][ks] = *reinterpret_cast<const bf16x8_t*>(Bs0 + boff[j][ks]);
    if constexpr (LORA) {
      pf[0] = *reinterpret_cast<const bf16x8_t*>(Bs0 + 2 * HT + poff0);
      pf[1] = *reinterpret_cast<const bf16x8_t*>(Bs0 + 2 * HT + poff1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i][ks] = *reinterpret_cast<const bf16x8_t*>(As0 + aoff[i][ks]);
    __builtin_amdgcn_sched_barrier(0);
    stage(kt + 1, P3{});
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    GSL_P8_MFMA(0, 0, bf0, 0)
    // ---- q1: (rh0, ch1); reads B-h1; stages B-h0(kt+2)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) bf1[j][ks] = *reinterpret_cast<const bf16x8_t*>(Bs1 + boff[j][ks]);
    __builtin_amdgcn_sched_barrier(0);
    stage(kt + 2, P0{});
    GSL_P8_MFMA(0, 1, bf1, 1)
    // ---- q2: (rh1, ch1); reads A-h1; stages A-h0(kt+2)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i][ks] = *reinterpret_cast<const bf16x8_t*>(As1 + aoff[i][ks]);
    __builtin_amdgcn_sched_barrier(0);
    stage(kt + 2, P1{});
    GSL_P8_MFMA(1, 1, bf1, 2)
    // ---- q3: (rh1, ch0); no reads; stages B-h1(kt+2); the once-per-K-tile counted wait: K tile kt+1 has landed
    stage(kt + 2, P2{});
    if (kt + 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (LORA && wave < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG) : "memory");
    GSL_P8_MFMA(1, 0, bf0, 3)
  }
#undef GSL_P8_MFMA
#undef GSL_P8_PEXTRA
  if (wm == 0) __builtin_amdgcn_s_barrier();     // re-balance the barrier count of the stagger
  if (dbg8) dbg8[2] = __builtin_readcyclecounter();
  ResRows rs0;
  constexpr bool RES16 = EPI == GSL_EPI_BIAS_RES_BF16;
  const bool res16_staged = RES16 && (e.N % 8) == 0 && (e.ldo % 8) == 0 && e.N >= 8;      // (the condition of the staged epilogue below)
  if constexpr (RES16) {
    if (res16_staged) res_rows_request0(e, rs0, m0 + wm * 128, n0 + wn * 64, lane);
    asm volatile("" ::: "memory");
  }
  if constexpr (LORA && GRAD) {
    // same as below with t kept as [256][16] behind the staging regions (the K-loop stages end before it: no barrier needed first)
    bf16_t* t16 = smem + (8 * GF_WAVE_B) / 2;
    // the Q fragments of the rank-r tail are requested FIRST: their round trip passes under the t hand-over and its barrier (requested behind
    // the barrier they cost the tail ~2 k exposed cycles per tile; the K loop's operand-fragment registers are dead from here on)
    bf16x8_t qf[4];
    if (GSL_MG_ON(e, 32))
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = min(n0 + wn * 64 + j * 16 + fr, e.N - 1);
      qf[j] = *reinterpret_cast<const bf16x8_t*>(lk.Q + (size_t)n * lk.ldq + fc * 8);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16_t* d = t16 + (wm * 128 + GSL_P8_TFRAG(t) * 16 + fr) * 16 + fc * 4;
      *reinterpret_cast<uint2*>(d) = make_uint2(pack2o(lk.s * accp[t][0], lk.s * accp[t][1]), pack2o(lk.s * accp[t][2], lk.s * accp[t][3]));
    }
    __syncthreads();
    if (n0 == 0 && lk.tout && GSL_MG_ON(e, 64)) {
      const int row = tid >> 1, half = tid & 1;
      if (m0 + row < e.M) {
        bf16_t* dst = lk.tout + (size_t)(m0 + row) * lk.ldt + half * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint4 v = (half || c >= 2) ? make_uint4(0u, 0u, 0u, 0u) : *reinterpret_cast<const uint4*>(t16 + row * 16 + c * 8);
          *reinterpret_cast<uint4*>(dst + c * 8) = v;
        }
      }
    }
    // round 0 of the epilogue's global operands flies under the rank-r tail (the operand-fragment registers of the K loop are dead).
    // Placement: behind the t16 hand-over (the compiler closes the K loop's LDS-DMA stream with s_waitcnt vmcnt(0) in front of the
    // first LDS store) and behind the Q fragments (vmcnt retires in order: the MFMAs below must not wait for these loads).
    asm volatile("" ::: "memory");
    GfOperands go;
    gf_request<EPI == GSL_EPI_MUL_G8>(e, go, m0 + wm * 128, n0 + wn * 64, 0, lane);
    if (GSL_MG_ON(e, 32))
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bf16x8_t tf = *reinterpret_cast<const bf16x8_t*>(t16 + (wm * 128 + i * 16 + fr) * 16 + (fc & 1) * 8);
      if (fc >= 2) tf = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};       // k slots 16..31 of the rank-r k-step
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = GSL_MFMA16(qf[j], tf, acc[i][j], 0, 0, 0);
    }
    __syncthreads();            // every wave is done with the stages: reuse them for the staging regions
    epilogue_staged_mulgrad<8, EPI == GSL_EPI_MUL_G8>(e, acc, reinterpret_cast<char*>(smem) + wave * GF_WAVE_B, reinterpret_cast<char*>(smem) + (wave ^ 4) * GF_WAVE_B,
                               t16 + wm * 128 * 16, m0 + wm * 128, n0 + wn * 64, lane, wm, m0 / BM4, go);
    if (dbg8) dbg8[3] = __builtin_readcyclecounter();
    return;
  } else if constexpr (LORA) {
    // t = s * (A P^T): accp[t][reg] = T[row = wm*128 + GSL_P8_TFRAG(t)*16 + fr][j = fc*4 + reg] -> LDS [256][32] bf16 (cols 16..31 = 0)
    bf16x8_t qf[4];                                    // rank-r update: one more k-step (k = 32: r live columns); its Q fragments are requested
#pragma unroll                                         // here, so that their round trip passes under the two barriers of the t hand-over
    for (int j = 0; j < 4; ++j) {
      const int n = min(n0 + wn * 64 + j * 16 + fr, e.N - 1);
      qf[j] = *reinterpret_cast<const bf16x8_t*>(lk.Q + (size_t)n * lk.ldq + fc * 8);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // stages are free
    bf16_t* tbuf = smem;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16_t* d = tbuf + (wm * 128 + GSL_P8_TFRAG(t) * 16 + fr) * 32 + fc * 4;
      *reinterpret_cast<uint2*>(d) = make_uint2(pack2o(lk.s * accp[t][0], lk.s * accp[t][1]), pack2o(lk.s * accp[t][2], lk.s * accp[t][3]));
      *reinterpret_cast<uint2*>(d + 16) = make_uint2(0u, 0u);
    }
    wg_barrier_lds();
    if (n0 == 0 && lk.tout) {                          // one N-tile stores t for the gradient reductions: [M, 64], zero padded
      const int row = tid >> 1, half = tid & 1;
      if (m0 + row < e.M) {
        bf16_t* dst = lk.tout + (size_t)(m0 + row) * lk.ldt + half * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint4 v = half ? make_uint4(0u, 0u, 0u, 0u) : *reinterpret_cast<const uint4*>(tbuf + row * 32 + c * 8);
          *reinterpret_cast<uint4*>(dst + c * 8) = v;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bf16x8_t tf = *reinterpret_cast<const bf16x8_t*>(tbuf + (wm * 128 + i * 16 + fr) * 32 + fc * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = GSL_MFMA16(qf[j], tf, acc[i][j], 0, 0, 0);
    }
  }
  if constexpr (EPI == GSL_EPI_BIAS_RES_F32 || EPI == GSL_EPI_PATCH) {
    if ((e.N % 4) == 0 && (e.ldo % 4) == 0 && e.N >= 4 && (EPI != GSL_EPI_PATCH || e.ldo == e.N)) {
      __builtin_amdgcn_s_barrier();            // every wave is done with the stages (and tbuf): reuse them for C staging
      epilogue_staged_res_f32<8, EPI>(e, acc, reinterpret_cast<float*>(smem + wave * CST_WAVE), m0 + wm * 128, n0 + wn * 64, lane);
      if (dbg8) dbg8[3] = __builtin_readcyclecounter();
      return;
    }
  }
  if constexpr (EPI == GSL_EPI_BIAS_RES_BF16 || EPI == GSL_EPI_PATCH_BF16) {
    if ((e.N % 8) == 0 && (e.ldo % 8) == 0 && e.N >= 8 && (EPI != GSL_EPI_PATCH_BF16 || e.ldo == e.N)) {
      __builtin_amdgcn_s_barrier();            // every wave is done with the stages (and tbuf): reuse them for C staging
      epilogue_staged_res_bf16<8, EPI>(e, acc, reinterpret_cast<float*>(smem + wave * CST_WAVE), m0 + wm * 128, n0 + wn * 64, lane, rs0);
      if (dbg8) dbg8[3] = __builtin_readcyclecounter();
      return;
    }
  }
  if constexpr (!epi_out_is_f32<EPI>()) {
    if ((e.N % 8) == 0 && (e.ldo % 8) == 0 && (EPI != GSL_EPI_BIAS_GELU_G8 || ((e.N % 16) == 0 && (e.ldo % 16) == 0))) {
      __builtin_amdgcn_s_barrier();            // every wave is done with the stages: reuse them for C staging
      if constexpr (epi_is_mul<EPI>()) {
        if (e.N >= 8) {
          epilogue_staged_mul<8, EPI>(e, acc, reinterpret_cast<float*>(smem + wave * CST_WAVE), m0 + wm * 128, n0 + wn * 64, lane);
          if (dbg8) dbg8[3] = __builtin_readcyclecounter();
          return;
        }
      }
      epilogue_staged_bf16<EPI, 8, false, TAB>(e, acc, smem + wave * CSTW, m0 + wm * 128, n0 + wn * 64, lane, nullptr,
                                               (uint32_t)(uintptr_t)(lptr_t)(smem + 2 * STG));
      if (dbg8) dbg8[3] = __builtin_readcyclecounter();
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epilogue4<EPI, bf16_t>(e, m0 + wm * 128 + i * 16 + fr, n0 + wn * 64 + j * 16 + fc * 4, v);
    }
}

// (the OVERLAP GEMM of round 6 — 4-wave workgroups on 256 x 128 tiles, two resident per CU so that one's epilogue runs under the other's K loop,
//  csrc/gemm_o4.inc, development build only, GSL_O4=1 — ties this kernel on every shape of the step and loses 1 % on the step: two 256 x 128 tiles
//  need 1.5x the operand bytes of one 256 x 256 tile, and the L2 -> LDS stream is what bounds these K loops: profiles/r06_e_overlap_gemm.md)
#ifdef GSL_DEV
#include "gemm_o4.inc"
#endif

// (a 4-wave, one-wave-per-SIMD 32x32x16 form of the plain-store GEMM — csrc/gemm_w4.inc, development build only — has the faster K loop at
//  K = 512 and loses the tile on its epilogue: profiles/r05_e_kloop_4w32.md, r05_h_w4_stamps.md)
#ifdef GSL_DEV
#include "gemm_w4.inc"
#endif

// (a persistent form of this kernel — tiles walked per CU, the next tile's first K tile requested under the current epilogue — is
//  bit-identical and time-neutral: csrc/gemm_dev_c.inc, development build only; profiles/r04_notes.md)
#ifdef GSL_DEV
#include "gemm_dev_c.inc"
#else
#define GSL_P8_PERSISTENT 0
#endif

#ifdef GSL_DEV
#include "gemm_dev_b.inc"
#endif

// ------------------------------------------------------------------ f32 kernel (parity mode)
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A1, int lda1,
                                                       const float* __restrict__ W1, int ldw1, int K1,
                                                       const float* __restrict__ A2, int lda2,
                                                       const float* __restrict__ W2, int ldw2, int K2, EpiArgs e) {
  GSL_OP16_KERNEL_ENTRY();
  resolve_drop(e.drop);
  __shared__ __attribute__((aligned(16))) float As[16][68];
  __shared__ __attribute__((aligned(16))) float Ws[16][68];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int nbn = (e.N + 63) / 64;
  const int m0 = (blockIdx.x / nbn) * 64, n0 = (blockIdx.x % nbn) * 64;
  const int nk1 = K1 / 16, nk = nk1 + K2 / 16;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int lrow = tid >> 2, lk = (tid & 3) * 4;
  for (int kt = 0; kt < nk; ++kt) {
    const float* Ab; const float* Wb; int lda, ldw, k0;
    if (kt < nk1) { Ab = A1; Wb = W1; lda = lda1; ldw = ldw1; k0 = kt * 16; }
    else { Ab = A2; Wb = W2; lda = lda2; ldw = ldw2; k0 = (kt - nk1) * 16; }
    const int gm = min(m0 + lrow, e.M - 1), gn = min(n0 + lrow, e.N - 1);
    const float4 a4 = *reinterpret_cast<const float4*>(Ab + (size_t)gm * lda + k0 + lk);
    const float4 w4 = *reinterpret_cast<const float4*>(Wb + (size_t)gn * ldw + k0 + lk);
    __syncthreads();
    As[lk + 0][lrow] = a4.x; As[lk + 1][lrow] = a4.y; As[lk + 2][lrow] = a4.z; As[lk + 3][lrow] = a4.w;
    Ws[lk + 0][lrow] = w4.x; Ws[lk + 1][lrow] = w4.y; Ws[lk + 2][lrow] = w4.z; Ws[lk + 3][lrow] = w4.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 w = *reinterpret_cast<const float4*>(&Ws[k][tx * 4]);
      const float av[4] = {a.x, a.y, a.z, a.w}, wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], wv[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) epilogue4<EPI, float>(e, m0 + ty * 4 + i, n0 + tx * 4, acc[i]);
}

// ------------------------------------------------------------------ f32 kernel on the matrix cores (parity mode, round 4)
// v_mfma_f32_16x16x4_f32: f32 operands, f32 accumulate, and — MI355X_MICROARCH.md / cdna_hip_programming.md section 3 — bit for bit a
// k-ordered fmaf chain, i.e. exactly what gemm_f32_kernel computes per output element (acc = 0; k ascending over [A1 | A2]): the two
// kernels are interchangeable bit for bit (tests/test_hip_ops.py::test_f32_gemm_mfma_equals_valu_bitwise), so every f32 golden test holds
// on either. What changes is the rate: 64 FLOP/clk/SIMD without occupying the VALU (the VALU kernel ran at 57 TF/s on the FFN1 shape).
// 128x128 tile, 4 waves of 64x64 (4 x 4 fragments), BK = 32 floats (128-byte rows: the LDS-DMA staging, 16-byte-chunk XOR swizzle and
// double buffering of gemm_bf16_glds_kernel, byte for byte), operands swapped (mfma(W, A)) so that a lane owns 4 consecutive output
// columns of one row — the epilogue code of the bf16 kernels (epi_math / epilogue4) serves unchanged. Fragment reads are ds_read_b32
// (lane (fr, fc): row fr, k = 4 ks + fc — ascending k inside every MFMA): 8 reads per 16 MFMAs of 32 cycles, nowhere near a limit.
constexpr int FBK = 32;                 // floats per K tile (= BK bf16 in bytes)
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const float* __restrict__ A1, int lda1,
                                                            const float* __restrict__ W1, int ldw1, int K1,
                                                            const float* __restrict__ A2, int lda2,
                                                            const float* __restrict__ W2, int ldw2, int K2, EpiArgs e) {
  GSL_OP16_KERNEL_ENTRY();
  resolve_drop(e.drop);
  __shared__ __attribute__((aligned(16))) float smem[2][2][BM * FBK];      // [buffer][A | W][128 rows x 32 floats] = 64 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (e.N + BN - 1) / BN;
  const int tile = e.remap ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
  const int nk1 = K1 / FBK, nk = nk1 + K2 / FBK;
  const int lrow = lane >> 3, lc = lane & 7;
  auto issue = [&](int kt, int buf) {
    const float* Ab; const float* Wb; int lda, ldw, k0;
    if (kt < nk1) { Ab = A1; Wb = W1; lda = lda1; ldw = ldw1; k0 = kt * FBK; }
    else { Ab = A2; Wb = W2; lda = lda2; ldw = ldw2; k0 = (kt - nk1) * FBK; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rb = wave * 4 + i;                 // 8-row block: one 1 KB DMA per wave-instruction
      const int row = rb * 8 + lrow;
      const int c = lc ^ (row & 7);
      const int gm = min(m0 + row, e.M - 1), gn = min(n0 + row, e.N - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Ab + (size_t)gm * lda + k0 + c * 4), (lptr_t)(&smem[buf][0][rb * 8 * FBK]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)gn * ldw + k0 + c * 4), (lptr_t)(&smem[buf][1][rb * 8 * FBK]), 16, 0, 0);
    }
  };
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fc = lane >> 4;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) { issue(kt + 1, buf ^ 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float* As = smem[buf][0];
    const float* Ws = smem[buf][1];
    // (requesting the fragments of k-step ks + 1 before the MFMAs of step ks — two register sets — was measured neutral: 114 - 129 TF/s on the
    //  step's shapes either way, tools/probes/f32_gemm_bench.py; two waves per SIMD already cover the LDS round trip)
#pragma unroll
    for (int ks = 0; ks < FBK / 4; ++ks) {
      float af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int row = wm * 64 + i * 16 + fr; af[i] = As[row * FBK + ((ks ^ (row & 7)) << 2) + fc]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int row = wn * 64 + j * 16 + fr; wf[j] = Ws[row * FBK + ((ks ^ (row & 7)) << 2) + fc]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();      // every wave is done with this buffer before the next iteration's DMA overwrites it
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epilogue4<EPI, float>(e, m0 + wm * 64 + i * 16 + fr, n0 + wn * 64 + j * 16 + fc * 4, v);
    }
}

// Launch knobs. The product library has none: block-id remap on, K rotation off, non-temporal output stores, no stamps, and the
// tile is chosen from the shape alone. The development build (-DGSL_DEV -> libgslora_hip_dev.so, selected with GSLORA_HIP_LIB) reads
// the ablation / variant knobs of tools/bench_gemm*.py and tools/probes/ from the environment.
static inline void set_launch_knobs(EpiArgs& e, bool allow_krot) {
  e.mbase = 0; e.remap = 1; e.krot = 0; e.stmode = GSL_STMODE; e.f16 = 0; e.stamps = nullptr; e.stamps_all = 0; e.mrev = 0; e.pf = 0; e.o4_delay = 0;
  e.ln_mean = nullptr; e.ln_rstd = nullptr; e.ln_c = nullptr; e.ln_d = nullptr; e.ln_rs = 1;
#ifdef GSL_DEV
  { const char* rm = getenv("GSL_XCD_REMAP"); if (rm) e.remap = atoi(rm); }
  { const char* kr = getenv("GSL_KROT"); if (kr && allow_krot) e.krot = atoi(kr); }
  { const char* sm = getenv("GSL_STORE_MODE"); if (sm) e.stmode = atoi(sm); }
  { const char* sp = getenv("GSL_P8_STAMPS"); if (sp) e.stamps = reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 0)); }
  { const char* sp = getenv("GSL_P8_STAMPS_ALL"); e.stamps_all = sp && atoi(sp); }
  { const char* pf = getenv("GSL_PF"); if (pf) e.pf = atoi(pf); }
  { const char* od = getenv("GSL_O4_DELAY"); if (od) e.o4_delay = atoi(od); }
#else
  (void)allow_krot;
#endif
}

// development: GSL_MREV = bit mask of the 8-phase launches that walk their tiles in reverse order (key: epilogue id; STORE: 0 N > K,
// 11 N < K, 12 N == K; 16 + epilogue id for the in-kernel-LoRA form; 30 the gradient-fused FFN2-dX)
static inline int mrev_for(int key) {
#ifdef GSL_DEV
  static const long mask = [] { const char* m = getenv("GSL_MREV"); return m ? strtol(m, nullptr, 0) : 0L; }();
  return (int)((mask >> key) & 1);
#else
  (void)key;
  return 0;
#endif
}

// workgroups of the persistent kernels: one per CU
#if GSL_P8_PERSISTENT
static inline int p8p_grid() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t pr;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    n -= n % 8;      // xcd_remap of the sequence numbers assumes seq % 8 = blockIdx % 8
    if (n < 8) n = 8;
  }
  return n;
}
#endif

template <int EPI>
static int launch_gemm(int dtype, const void* A1, int lda1, const void* W1, int ldw1, int K1, const void* A2,
                       int lda2, const void* W2, int ldw2, int K2, const EpiArgs& e_in, hipStream_t st) {
  const EpiArgs& e = e_in;
  if (dtype == GSL_OP16) {
    const int nblk = ((e.M + BM - 1) / BM) * ((e.N + BN - 1) / BN);
    // Tile choice, measured on MI355X at M = 201 728 (profiles/r01_gemm_ab.md): N >= 512 wants the 256x256 8-phase tile (also for the
    // VALU-heavy BIAS_GELU epilogue), skinny N the 256x128 ring. Fewer than 128 tiles of 256x256 cannot fill the 256 CUs: the 128x128
    // kernel (4x the workgroups) wins there (measured at M = 1576: 15-44 us vs 19-58 us per GEMM); from ~150 tiles on the 8-phase
    // kernel is ahead.   1 = 128x128 single stage, 3 = 256x128 three-stage ring, 8 = 256x256 8-phase ping-pong.
    const long tiles256 = (long)((e.M + 255) / 256) * ((e.N + 255) / 256);
    int variant = (e.M < 1024 || tiles256 < 128) ? 1 : (e.N >= 512 ? 8 : 3);
    // up to ~256 tiles of 128x128 (one per CU) leave CUs idle and serialise the K loop: 64x64 tiles with a 4-stage ring (12). Measured at
    // M = 1 576 (tools/probes/small_m_gemm.py, profiles/r03_c_small_m.md).
    if (variant == 1 && (long)nblk <= 256) variant = 12;
    if (EPI == GSL_EPI_STORE && e.out2) variant = 3;      // the compact second output lives on the ring kernel (checked by the caller: N <= 128, no bias)
#define GSL_LAUNCH(KERNEL, NB, NT) hipLaunchKernelGGL(KERNEL, dim3(NB), dim3(NT), 0, st, (const bf16_t*)A1, lda1, (const bf16_t*)W1, \
                                                      ldw1, K1, (const bf16_t*)A2, lda2, (const bf16_t*)W2, ldw2, K2, e)
#ifdef GSL_DEV
    // development knobs: GSL_GEMM_VARIANT = 1 / 3 / 8 or the lab kernels 4 (256x256 two-stage), 9 (256x128x32, two workgroups per CU),
    // 10 (persistent ping-pong), 11 (in-wave pipelined); GSL_GEMM_ABL = main-loop ablation of the ring kernel (tools/bench_gemm_abl.py)
    const char* ev = getenv("GSL_GEMM_VARIANT");
    if (ev) variant = atoi(ev);
    if (EPI == GSL_EPI_BIAS_GELU && !ev && variant == 8) { const char* gv = getenv("GSL_GELU_VARIANT"); if (gv) variant = atoi(gv); }
    if (variant == 10 && (K1 + K2) >= 192 && (e.N % 8) == 0 && (e.ldo % 8) == 0 && e.N >= 8) {
      if constexpr (EPI == GSL_EPI_STORE || EPI == GSL_EPI_BIAS_GELU) {
        EpiArgs e = e_in;
        { const char* ab = getenv("GSL_PP_ABL"); e.T = ab ? atoi(ab) : 0; }
        const int ntm = (e.M + PP_TM - 1) / PP_TM, ntn = (e.N + PP_TN - 1) / PP_TN;
        const int nunits = ntm * ((ntn % 2 == 0) ? 2 : 1);
        int grid = nunits < 256 ? nunits : 256;
        hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI>), dim3(grid), dim3(512), 0, st, (const bf16_t*)A1, lda1, (const bf16_t*)W1, ldw1, K1,
                           (const bf16_t*)A2, lda2, (const bf16_t*)W2, ldw2, K2, e);
        return check_launch("gsl_gemm_nt(pp)");
      }
    }
    if (variant == 11 && (e.M % IW_TM) == 0 && (e.N % IW_TN) == 0 && (e.ldo % 8) == 0 && (K1 % 64) == 0 && (K2 % 64) == 0 &&
        ((K1 + K2) == 512 || (K1 + K2) == 576) && (EPI != GSL_EPI_BIAS_GELU || e.out2)) {
      if constexpr (EPI == GSL_EPI_STORE || EPI == GSL_EPI_BIAS_GELU) {
        EpiArgs e = e_in;
        { const char* ab = getenv("GSL_PP_ABL"); e.T = ab ? atoi(ab) : 0; }
        const int ntm = e.M / IW_TM, ntn = e.N / IW_TN;
        const int nunits = ntm * ((ntn % 2 == 0) ? 2 : 1);
        const int grid = nunits < 256 ? nunits : 256;
        if ((K1 + K2) == 576)
          hipLaunchKernelGGL((gemm_bf16_iw_kernel<EPI, 9>), dim3(grid), dim3(512), 0, st, (const bf16_t*)A1, lda1, (const bf16_t*)W1, ldw1, K1,
                             (const bf16_t*)A2, lda2, (const bf16_t*)W2, ldw2, K2, e);
        else
          hipLaunchKernelGGL((gemm_bf16_iw_kernel<EPI, 8>), dim3(grid), dim3(512), 0, st, (const bf16_t*)A1, lda1, (const bf16_t*)W1, ldw1, K1,
                             (const bf16_t*)A2, lda2, (const bf16_t*)W2, ldw2, K2, e);
        return check_launch("gsl_gemm_nt(iw)");
      }
    }
    if constexpr (EPI != GSL_EPI_BIAS_RES_BF16 && EPI != GSL_EPI_PATCH_BF16 && EPI != GSL_EPI_MUL_G8 && EPI != GSL_EPI_BIAS_GELU_G8) {
      if (variant == 9) {
        EpiArgs e9 = e;
        if (EPI != GSL_EPI_PATCH) { const char* sg = getenv("GSL_STAGGER"); e9.T = sg ? atoi(sg) : 0; }
        hipLaunchKernelGGL(gemm_bf16_k32x2_kernel<EPI>, dim3(((e.M + 255) / 256) * ((e.N + 127) / 128)), dim3(512), 0, st, (const bf16_t*)A1, lda1,
                           (const bf16_t*)W1, ldw1, K1, (const bf16_t*)A2, lda2, (const bf16_t*)W2, ldw2, K2, e9);
        return check_launch("gsl_gemm_nt(k32x2)");
      }
      if (variant == 4) {
        GSL_LAUNCH(gemm_bf16_t256_kernel<EPI>, ((e.M + BM4 - 1) / BM4) * ((e.N + BN4 - 1) / BN4), 512);
        return check_launch("gsl_gemm_nt(t256)");
      }
    }
    if constexpr (EPI == GSL_EPI_STORE) {      // probe: W fragments straight from L2 into registers (13 full, 14 A-DMA stream alone, 15 A-DMA + W loads alone)
      if (variant >= 13 && variant <= 15) {
        const int nb3 = ((e.M + BM3 - 1) / BM3) * ((e.N + BN3 - 1) / BN3);
        if (variant == 13) GSL_LAUNCH((gemm_bf16_ring3w_kernel<EPI, 0>), nb3, 512);
        else if (variant == 14) GSL_LAUNCH((gemm_bf16_ring3w_kernel<EPI, 1>), nb3, 512);
        else GSL_LAUNCH((gemm_bf16_ring3w_kernel<EPI, 2>), nb3, 512);
        return check_launch("gsl_gemm_nt(ring3w probe)");
      }
    }
    if constexpr (EPI == GSL_EPI_STORE) {
      const char* ab = getenv("GSL_GEMM_ABL");
      const int abl = ab ? atoi(ab) : 0;
      if (variant == 3 && abl) {
        const int nb3 = ((e.M + BM3 - 1) / BM3) * ((e.N + BN3 - 1) / BN3);
        switch (abl) {
          case 1: GSL_LAUNCH((gemm_bf16_ring3_kernel<EPI, 1>), nb3, 512); break;
          case 2: GSL_LAUNCH((gemm_bf16_ring3_kernel<EPI, 2>), nb3, 512); break;
          case 3: GSL_LAUNCH((gemm_bf16_ring3_kernel<EPI, 3>), nb3, 512); break;
          case 4: GSL_LAUNCH((gemm_bf16_ring3_kernel<EPI, 4>), nb3, 512); break;
          case 5: GSL_LAUNCH((gemm_bf16_ring3_kernel<EPI, 5>), nb3, 512); break;
          case 6: GSL_LAUNCH((gemm_bf16_ring3_kernel<EPI, 6>), nb3, 512); break;
          case 9: GSL_LAUNCH((gemm_bf16_ring3_kernel<EPI, 9>), nb3, 512); break;
          default: GSL_LAUNCH((gemm_bf16_ring3_kernel<EPI, 11>), nb3, 512);
        }
        return check_launch("gsl_gemm_nt(ring3 ablation)");
      }
    }
#endif
#if GSL_P8_PERSISTENT
    if constexpr (EPI == GSL_EPI_STORE) {
      // plain-store GEMMs (QKV, out-proj dX, QKV dX, LoRA-free dX) with at least two rounds of tiles: the persistent form (see the kernel)
      const int nt8 = ((e.M + BM4 - 1) / BM4) * ((e.N + BN4 - 1) / BN4);
      if (variant == 8 && (e.N % 8) == 0 && (e.ldo % 8) == 0 && nt8 >= 2 * p8p_grid()) {
        hipLaunchKernelGGL((gemm_bf16_p8p_kernel<EPI>), dim3(p8p_grid()), dim3(512), 0, st, (const bf16_t*)A1, lda1, (const bf16_t*)W1, ldw1, K1,
                           (const bf16_t*)A2, lda2, (const bf16_t*)W2, ldw2, K2, nt8, e);
        return check_launch("gsl_gemm_nt(p8p)");
      }
    }
#endif
#ifdef GSL_DEV
    if constexpr (EPI == GSL_EPI_STORE) {
      // development (GSL_W4=1): plain-store GEMMs on the 4-wave 32x32x16 kernel where its shape rules hold (gemm_w4.inc; measured alternative)
      const char* w = getenv("GSL_W4");
      if (w && atoi(w) == 1 && variant == 8 && w4_usable(e.M, e.N, K1, K2, lda1, ldw1, e.ldo)) {
        hipLaunchKernelGGL((gemm_op16_w4_kernel<EPI>), dim3(((e.M + 255) / 256) * (e.N / 256)), dim3(256), 0, st, (const op16_t*)A1, lda1,
                           (const op16_t*)W1, ldw1, K1, e);
        return check_launch("gsl_gemm_nt(w4)");
      }
    }
#endif
#ifdef GSL_DEV
    if constexpr (EPI == GSL_EPI_STORE || EPI == GSL_EPI_BIAS_GELU_G8) {
      // development (GSL_O4=1): the plain-store and fused-FFN1 GEMMs of the 8-phase class on the overlap kernel where its shape rules hold
      // (gemm_o4.inc; a measured alternative). GSL_O4_ONE_PER_CU=1: 16 KB of dynamic LDS on top = one workgroup per CU.
      const char* o = getenv("GSL_O4");
      if (o && atoi(o) != 0 && variant == 8 && !(EPI == GSL_EPI_STORE && e.out2) && o4_usable(e.M, e.N, K1, K2, lda1, ldw1, lda2, ldw2, e.ldo)) {
        const char* o1 = getenv("GSL_O4_ONE_PER_CU");
        const int o4_dyn = (o1 && atoi(o1)) ? 16384 : 0;
        hipLaunchKernelGGL((gemm_op16_o4_kernel<EPI>), dim3(((e.M + O4_BM - 1) / O4_BM) * (e.N / O4_BN)), dim3(256), o4_dyn, st, (const op16_t*)A1, lda1,
                           (const op16_t*)W1, ldw1, K1, (const op16_t*)A2, lda2, (const op16_t*)W2, ldw2, K2, e);
        return check_launch("gsl_gemm_nt(o4)");
      }
    }
#endif
    if (variant == 8) {
      EpiArgs e8 = e;
      e8.mrev = mrev_for(EPI == GSL_EPI_STORE ? (e.N > K1 ? 0 : (e.N < K1 ? 11 : 12)) : EPI);
      hipLaunchKernelGGL((gemm_bf16_p8_kernel<EPI, false>), dim3(((e.M + BM4 - 1) / BM4) * ((e.N + BN4 - 1) / BN4)), dim3(512), 0, st,
                         (const bf16_t*)A1, lda1, (const bf16_t*)W1, ldw1, K1, (const bf16_t*)A2, lda2, (const bf16_t*)W2, ldw2, K2, LoraInk{}, e8);
    } else if (variant == 3) {
      GSL_LAUNCH((gemm_bf16_ring3_kernel<EPI, 0>), ((e.M + BM3 - 1) / BM3) * ((e.N + BN3 - 1) / BN3), 512);
    } else if (variant == 12) {
      // more 64x64 tiles than resident workgroups (3 per CU): 64x128 tiles, one round
      if ((long)((e.M + BMS - 1) / BMS) * ((e.N + BNS - 1) / BNS) > SMALL_SLOTS)
        hipLaunchKernelGGL((gemm_bf16_small_kernel<EPI, false, 4>), dim3(((e.M + BMS - 1) / BMS) * ((e.N + 2 * BNS - 1) / (2 * BNS))), dim3(256), 0, st,
                           (const bf16_t*)A1, lda1, (const bf16_t*)W1, ldw1, K1, (const bf16_t*)A2, lda2, (const bf16_t*)W2, ldw2, K2, LoraInk{}, e);
      else if (GSL_SMALL_KSPLIT && (K1 + K2) >= GSL_SMALL_KSPLIT_MINK && (long)((e.M + BMS - 1) / BMS) * ((e.N + BNS - 1) / BNS) <= 256)      // a serial K chain on <= one workgroup per CU
        hipLaunchKernelGGL((gemm_bf16_small_kernel<EPI, false, 2, 2>), dim3(((e.M + BMS - 1) / BMS) * ((e.N + BNS - 1) / BNS)), dim3(512), 0, st,
                           (const bf16_t*)A1, lda1, (const bf16_t*)W1, ldw1, K1, (const bf16_t*)A2, lda2, (const bf16_t*)W2, ldw2, K2, LoraInk{}, e);
      else
        hipLaunchKernelGGL((gemm_bf16_small_kernel<EPI, false, 2>), dim3(((e.M + BMS - 1) / BMS) * ((e.N + BNS - 1) / BNS)), dim3(256), 0, st,
                           (const bf16_t*)A1, lda1, (const bf16_t*)W1, ldw1, K1, (const bf16_t*)A2, lda2, (const bf16_t*)W2, ldw2, K2, LoraInk{}, e);
    } else {
      GSL_LAUNCH((gemm_bf16_glds_kernel<EPI, 1>), nblk, 256);
    }
#undef GSL_LAUNCH
  }
#if GSL_HAS_F32
  else {
    // parity mode: the matrix-core kernel wherever its 128x128 tiles are not mostly padding (skinny N = 64 LoRA projections, a handful of
    // rows: the 64x64 VALU kernel); the two are bit-identical, the choice is speed only
    // (development build: GSL_F32_VALU=1 forces the VALU kernel at every shape — the switch for parity debugging should the MFMA's 4-term
    //  accumulation ever stop being a k-ordered fmaf chain on another part or compiler; tests/test_hip_ops.py compares the two directly)
#ifdef GSL_DEV
    const char* fv = getenv("GSL_F32_VALU");
    const bool force_valu = fv && atoi(fv) != 0;
#else
    constexpr bool force_valu = false;
#endif
    if (e.N >= 128 && e.M >= 64 && !force_valu) {
      EpiArgs ef = e;
      ef.remap = 1;
      const int nblk = ((e.M + BM - 1) / BM) * ((e.N + BN - 1) / BN);
      hipLaunchKernelGGL(gemm_f32_mfma_kernel<EPI>, dim3(nblk), dim3(256), 0, st, (const float*)A1, lda1, (const float*)W1,
                         ldw1, K1, (const float*)A2, lda2, (const float*)W2, ldw2, K2, ef);
    } else {
      const int nblk = ((e.M + 63) / 64) * ((e.N + 63) / 64);
      hipLaunchKernelGGL(gemm_f32_kernel<EPI>, dim3(nblk), dim3(256), 0, st, (const float*)A1, lda1, (const float*)W1,
                         ldw1, K1, (const float*)A2, lda2, (const float*)W2, ldw2, K2, e);
    }
  }
#endif
  return check_launch("gsl_gemm_nt");
}

// ---- tail split (round 6). The 8-phase kernel runs one 256 x 256 tile per CU at a time, so a launch takes ceil(tiles / CUs) tile times: the
// N = 512 GEMMs of the step (out-proj forward and dX, QKV dX, FFN1-dX, FFN2 forward: 1 576 tiles on 256 CUs = 6.16 rounds) pay SEVEN — 12 % of
// five GEMMs per layer. When the last round would be at most 30 % full, the launch is split by rows: whole rounds of 256 x 256 tiles, then the
// remaining row panels as a second launch, which the tile rule puts on the 64 x 64 ring kernel (640 small workgroups, three per CU: ~0.25 tile
// times instead of one). Same MFMA instruction, same k order, same epilogue arithmetic: the rows of the tail are bit-identical to what the
// 8-phase kernel writes (tests/test_hip_ops.py::test_gemm_tail_split_*); their dropout counters continue at row `mbase` (EpiArgs::mbase).
static inline int gemm_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t pr;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  return n;
}
#ifndef GSL_TAIL_SPLIT
#define GSL_TAIL_SPLIT 1
#endif
// rows of the tail launch (0: no split): 16-bit operands on the 8-phase class, whole tiles, a last round of at most 30 % of the CUs
static inline int tail_split_rows(int M, int N, int dtype) {
  bool on = GSL_TAIL_SPLIT != 0;
#ifdef GSL_DEV
  { const char* t = getenv("GSL_TAIL_SPLIT"); if (t) on = atoi(t) != 0; }
#endif
  if (!on || dtype != GSL_OP16 || (M % 256) || (N % 256) || N < 512) return 0;
  const long nt = N / 256, tiles = (long)(M / 256) * nt, ncu = gemm_num_cus();
  if (tiles <= ncu) return 0;
  const long r = tiles % ncu;
  if (r == 0 || r * 10 > ncu * 3 || (r % nt)) return 0;
  return (int)(r / nt) * 256;
}
static inline const void* rows_after(const void* p, long rows, long ld, int esize) { return p ? static_cast<const char*>(p) + rows * ld * esize : nullptr; }
static inline void* rows_after(void* p, long rows, long ld, int esize) { return p ? static_cast<char*>(p) + rows * ld * esize : nullptr; }

static int gemm_nt_rows(const void* A1, int lda1, const void* W1, int ldw1, int K1, const void* A2, int lda2,
                        const void* W2, int ldw2, int K2, int M, int N, int dtype, int epilogue, float alpha,
                        const float* bias, const void* res, const void* aux, void* out, void* out2, int ldo,
                        const float* pos, const float* cls, int T, float p_drop, uint64_t seed, uint32_t site,
                        gsl_stream_t s, int mbase);

extern "C" int GSL_ENTRY(gsl_gemm_nt)(const void* A1, int lda1, const void* W1, int ldw1, int K1, const void* A2, int lda2,
                           const void* W2, int ldw2, int K2, int M, int N, int dtype, int epilogue, float alpha,
                           const float* bias, const void* res, const void* aux, void* out, void* out2, int ldo,
                           const float* pos, const float* cls, int T, float p_drop, uint64_t seed, uint32_t site,
                           gsl_stream_t s) {
  GSL_FORWARD_H16(dtype, h16_gsl_gemm_nt(A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, M, N, dtype, epilogue, alpha, bias, res, aux, out, out2,
                                         ldo, pos, cls, T, p_drop, seed, site, s));
  // plain STORE and the 16-bit residual epilogues split (operands, residual and output are 2-byte row-major tensors: the tail is a pointer offset)
  const bool splittable = (epilogue == GSL_EPI_STORE && !out2) || epilogue == GSL_EPI_BIAS_RES_BF16 || epilogue == GSL_EPI_BIAS_RES_F16;
  const int tail = (splittable && A1 && W1 && out && M >= 1024) ? tail_split_rows(M, N, dtype) : 0;
  if (tail > 0) {
    const int head = M - tail;
    const int rc = gemm_nt_rows(A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, head, N, dtype, epilogue, alpha, bias, res, aux, out, out2, ldo, pos, cls,
                                T, p_drop, seed, site, s, 0);
    if (rc) return rc;
    return gemm_nt_rows(rows_after(A1, head, lda1, 2), lda1, W1, ldw1, K1, K2 ? rows_after(A2, head, lda2, 2) : A2, lda2, W2, ldw2, K2, tail, N, dtype,
                        epilogue, alpha, bias, rows_after(res, head, ldo, 2), aux, rows_after(out, head, ldo, 2), out2, ldo, pos, cls, T, p_drop, seed, site, s,
                        head);
  }
  return gemm_nt_rows(A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, M, N, dtype, epilogue, alpha, bias, res, aux, out, out2, ldo, pos, cls, T, p_drop, seed,
                      site, s, 0);
}

static int gemm_nt_rows(const void* A1, int lda1, const void* W1, int ldw1, int K1, const void* A2, int lda2,
                        const void* W2, int ldw2, int K2, int M, int N, int dtype, int epilogue, float alpha,
                        const float* bias, const void* res, const void* aux, void* out, void* out2, int ldo,
                        const float* pos, const float* cls, int T, float p_drop, uint64_t seed, uint32_t site,
                        gsl_stream_t s, int mbase) {
  GSL_CHECK_ARG((GSL_HAS_F32 && dtype == GSL_F32) || dtype == GSL_OP16, "dtype");
  GSL_CHECK_ARG(M > 0 && N > 0 && (N % 4) == 0, "M>0, N>0, N%4==0");
  GSL_CHECK_ARG(K1 > 0 && (K1 % 64) == 0 && K2 >= 0 && (K2 % 64) == 0, "K1,K2 multiples of 64");
  GSL_CHECK_ARG(A1 && W1 && out && (K2 == 0 || (A2 && W2)), "null operand");
  GSL_CHECK_ARG((lda1 % 8) == 0 && (ldw1 % 8) == 0 && (K2 == 0 || ((lda2 % 8) == 0 && (ldw2 % 8) == 0)) && (ldo % 4) == 0,
                "leading dimensions must keep 16-byte alignment");
  GSL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "p_drop");
  EpiArgs e;
  e.alpha = alpha; e.bias = bias; e.res = res; e.aux = aux; e.out = out; e.out2 = out2; e.ldo = ldo;
  e.pos = pos; e.cls = cls; e.T = T; e.drop = make_drop(p_drop, seed, site); e.M = M; e.N = N;
  set_launch_knobs(e, true);
  e.mbase = mbase;
  e.hmT = 0; e.hmH = 0;
  hipStream_t st = as_stream(s);
  switch (epilogue) {
    case GSL_EPI_STORE:
      if (out2) GSL_CHECK_ARG(dtype != GSL_F32 && N >= 16 && N <= 128 && !bias, "STORE with out2 (compact [M,16] copy of columns 0..15): 16-bit operands, 16 <= N <= 128, no bias");
      return launch_gemm<GSL_EPI_STORE>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    case GSL_EPI_STORE_F32: return launch_gemm<GSL_EPI_STORE_F32>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    case GSL_EPI_STORE_QKV_HM_LN:
    case GSL_EPI_STORE_LN:          // consumer-side LayerNorm: pos = mean [M], cls = rstd [M], aux = c [N] (f32), bias = d [N] (required)
      GSL_CHECK_ARG(pos && cls && aux && bias && !out2 && alpha == 1.0f, "STORE_LN: pos = mean[M], cls = rstd[M], aux = c[N], bias = d[N] (all f32), alpha 1, no out2");
      e.ln_mean = pos; e.ln_rstd = cls; e.ln_c = reinterpret_cast<const float*>(aux); e.ln_d = bias; e.bias = nullptr; e.aux = nullptr; e.pos = nullptr; e.cls = nullptr;
      if (epilogue == GSL_EPI_STORE_LN) {
        GSL_CHECK_ARG(T >= 0, "STORE_LN: T = 0, or the row stride of mean / rstd (row m reads element m * T: the cls rows of a [B * T] tensor)");
        e.ln_rs = T > 0 ? T : 1;
        return launch_gemm<GSL_EPI_STORE>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
      }
      [[fallthrough]];
    case GSL_EPI_STORE_QKV_HM:      // the STORE kernels with a permuting copy-out: out is [B][H][3][T][64], M = B * T rows, N = 3 * H * 64
      GSL_CHECK_ARG(dtype == GSL_OP16 && T >= 8 && (M % T) == 0 && (N % 192) == 0 && ldo == N, "STORE_QKV_HM: bf16, M = B*T (T >= 8), N = 3*H*64, ldo = N");
      e.hmT = T; e.hmH = N / 192;
      return launch_gemm<GSL_EPI_STORE>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    case GSL_EPI_BIAS_RES_F32:
      GSL_CHECK_ARG(bias && res, "bias/res required");
      return launch_gemm<GSL_EPI_BIAS_RES_F32>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    case GSL_EPI_BIAS_RES_F16:      // the same kernels with the stream element type switched at run time (EpiArgs::f16)
      e.f16 = 1;
      [[fallthrough]];
    case GSL_EPI_BIAS_RES_BF16:
      GSL_CHECK_ARG(bias && res && dtype == GSL_OP16 && (ldo % 8) == 0, "bias/res required, bf16 only");
      return launch_gemm<GSL_EPI_BIAS_RES_BF16>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    case GSL_EPI_PATCH_F16:
      e.f16 = 1;
      [[fallthrough]];
    case GSL_EPI_PATCH_BF16:
      GSL_CHECK_ARG(bias && pos && cls && T > 0 && dtype == GSL_OP16 && (ldo % 8) == 0, "bias/pos/cls/T required, bf16 only");
      return launch_gemm<GSL_EPI_PATCH_BF16>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    case GSL_EPI_BIAS_GELU:
      GSL_CHECK_ARG(bias, "bias required");
      return launch_gemm<GSL_EPI_BIAS_GELU>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    case GSL_EPI_MUL:
      GSL_CHECK_ARG(aux, "aux required");
      return launch_gemm<GSL_EPI_MUL>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    case GSL_EPI_BIAS_GELU_G8:
      GSL_CHECK_ARG(bias && dtype == GSL_OP16 && (N % 64) == 0, "bias required, bf16 only, N % 64 == 0 (slab-major code tensor)");
      return launch_gemm<GSL_EPI_BIAS_GELU_G8>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    case GSL_EPI_MUL_G8:      // aux = 8-bit GELU' codes [M, ldo bytes]; p_drop = the dropout rate of the forward that wrote them (no mask is applied here)
      GSL_CHECK_ARG(aux && dtype == GSL_OP16 && (N % 64) == 0, "aux required, bf16 only, N % 64 == 0 (slab-major code tensor)");
      return launch_gemm<GSL_EPI_MUL_G8>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    case GSL_EPI_PATCH:
      GSL_CHECK_ARG(bias && pos && cls && T > 0, "bias/pos/cls/T required");
      return launch_gemm<GSL_EPI_PATCH>(dtype, A1, lda1, W1, ldw1, K1, A2, lda2, W2, ldw2, K2, e, st);
    default: return fail(GSL_ERR_ARG, "gsl_gemm_nt: unknown epilogue%s %ld", "", epilogue);
  }
}

static int gemm_nt_lora_rows(const void* A, int lda, const void* W, int ldw, int K, const void* P, int ldp, const void* Q,
                             int ldq, float lora_scale, void* tout, int ldt, int M, int N, int dtype, int epilogue,
                             const float* bias, const void* res, const void* aux, void* out, void* out2, int ldo,
                             float p_drop, uint64_t seed, uint32_t site, gsl_stream_t s, int mbase);

extern "C" int GSL_ENTRY(gsl_gemm_nt_lora)(const void* A, int lda, const void* W, int ldw, int K, const void* P, int ldp, const void* Q,
                                int ldq, float lora_scale, void* tout, int ldt, int M, int N, int dtype, int epilogue,
                                const float* bias, const void* res, const void* aux, void* out, void* out2, int ldo,
                                float p_drop, uint64_t seed, uint32_t site, gsl_stream_t s) {
  GSL_FORWARD_H16(dtype, h16_gsl_gemm_nt_lora(A, lda, W, ldw, K, P, ldp, Q, ldq, lora_scale, tout, ldt, M, N, dtype, epilogue, bias, res, aux, out,
                                              out2, ldo, p_drop, seed, site, s));
  // tail split (see gsl_gemm_nt): the in-kernel-LoRA form exists on the 64 x 64 ring kernel too; t = s A P^T of the tail rows goes to tout's tail rows
  const bool splittable = (epilogue == GSL_EPI_STORE && !out2) || epilogue == GSL_EPI_BIAS_RES_BF16 || epilogue == GSL_EPI_BIAS_RES_F16;
  const int tail = (splittable && A && W && out && M >= 1024) ? tail_split_rows(M, N, dtype) : 0;
  if (tail > 0) {
    const int head = M - tail;
    const int rc = gemm_nt_lora_rows(A, lda, W, ldw, K, P, ldp, Q, ldq, lora_scale, tout, ldt, head, N, dtype, epilogue, bias, res, aux, out, out2, ldo,
                                     p_drop, seed, site, s, 0);
    if (rc) return rc;
    return gemm_nt_lora_rows(rows_after(A, head, lda, 2), lda, W, ldw, K, P, ldp, Q, ldq, lora_scale, rows_after(tout, head, ldt, 2), ldt, tail, N, dtype,
                             epilogue, bias, rows_after(res, head, ldo, 2), aux, rows_after(out, head, ldo, 2), out2, ldo, p_drop, seed, site, s, head);
  }
  return gemm_nt_lora_rows(A, lda, W, ldw, K, P, ldp, Q, ldq, lora_scale, tout, ldt, M, N, dtype, epilogue, bias, res, aux, out, out2, ldo, p_drop, seed,
                           site, s, 0);
}

static int gemm_nt_lora_rows(const void* A, int lda, const void* W, int ldw, int K, const void* P, int ldp, const void* Q,
                             int ldq, float lora_scale, void* tout, int ldt, int M, int N, int dtype, int epilogue,
                             const float* bias, const void* res, const void* aux, void* out, void* out2, int ldo,
                             float p_drop, uint64_t seed, uint32_t site, gsl_stream_t s, int mbase) {
  if (dtype != GSL_OP16) return fail(GSL_ERR_UNSUPPORTED, "gsl_gemm_nt_lora: bf16 / fp16 operands only (f32 parity mode uses gsl_gemm_nt with a K segment)%s %ld", "", dtype);
  GSL_CHECK_ARG(M > 0 && N > 0 && (N % 4) == 0 && K > 0 && (K % 64) == 0, "M,N>0, N%4==0, K%64==0");
  GSL_CHECK_ARG(A && W && P && Q && out, "null operand");
  GSL_CHECK_ARG((lda % 8) == 0 && (ldw % 8) == 0 && (ldp % 8) == 0 && (ldq % 8) == 0 && ldq >= 32 && (ldo % 4) == 0 &&
                (!tout || ((ldt % 8) == 0 && ldt >= 64)), "leading dimensions (P [16,K], Q [N,>=32], tout [M,>=64])");
  GSL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "p_drop");
  EpiArgs e;
  e.alpha = 1.0f; e.bias = bias; e.res = res; e.aux = aux; e.out = out; e.out2 = out2; e.ldo = ldo;
  e.pos = nullptr; e.cls = nullptr; e.T = 0; e.drop = make_drop(p_drop, seed, site); e.M = M; e.N = N;
  set_launch_knobs(e, true);
  e.mbase = mbase;
  e.hmT = 0; e.hmH = 0;
  LoraInk lk;
  lk.P = (const bf16_t*)P; lk.ldp = ldp; lk.Q = (const bf16_t*)Q; lk.ldq = ldq; lk.s = lora_scale; lk.tout = (bf16_t*)tout; lk.ldt = ldt;
  const int nb = ((M + BM4 - 1) / BM4) * ((N + BN4 - 1) / BN4);
  hipStream_t st = as_stream(s);
#ifdef GSL_DEV
  const char* ev = getenv("GSL_GEMM_VARIANT");      // development knob: 4 = single-phase 256x256 kernel, default = 8-phase schedule
  const bool old_sched = ev && atoi(ev) == 4;
#endif
#ifdef GSL_DEV
#define GSL_LL_DEV(EPIV)                                                                                                          \
    if (old_sched) { hipLaunchKernelGGL(gemm_bf16_t256_lora_kernel<EPIV>, dim3(nb), dim3(512), 0, st, (const bf16_t*)A, lda,       \
                                        (const bf16_t*)W, ldw, K, lk, e); break; }
#else
#define GSL_LL_DEV(EPIV)
#endif
  // few rows (the launch-bound regime): the 64x64 ring kernel, same rule as gsl_gemm_nt's tile choice
  const long tiles256 = (long)nb, nblk128 = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const bool small = (M < 1024 || tiles256 < 128) && nblk128 <= 256;
#define GSL_LL(EPIV)                                                                                                              \
  do {                                                                                                                            \
    GSL_LL_DEV(EPIV)                                                                                                              \
    if (small) {                                                                                                                  \
      if ((long)((M + BMS - 1) / BMS) * ((N + BNS - 1) / BNS) > SMALL_SLOTS)                                                      \
        hipLaunchKernelGGL((gemm_bf16_small_kernel<EPIV, true, 4>), dim3(((M + BMS - 1) / BMS) * ((N + 2 * BNS - 1) / (2 * BNS))), dim3(256), 0, st, \
                           (const bf16_t*)A, lda, (const bf16_t*)W, ldw, K, (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, 0, 0, lk, e); \
      else if (GSL_SMALL_KSPLIT && K >= GSL_SMALL_KSPLIT_MINK && (long)((M + BMS - 1) / BMS) * ((N + BNS - 1) / BNS) <= 256)                         \
        hipLaunchKernelGGL((gemm_bf16_small_kernel<EPIV, true, 2, 2>), dim3(((M + BMS - 1) / BMS) * ((N + BNS - 1) / BNS)), dim3(512), 0, st, \
                           (const bf16_t*)A, lda, (const bf16_t*)W, ldw, K, (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, 0, 0, lk, e); \
      else                                                                                                                        \
        hipLaunchKernelGGL((gemm_bf16_small_kernel<EPIV, true, 2>), dim3(((M + BMS - 1) / BMS) * ((N + BNS - 1) / BNS)), dim3(256), 0, st, \
                           (const bf16_t*)A, lda, (const bf16_t*)W, ldw, K, (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, 0, 0, lk, e); \
      break;                                                                                                                      \
    }                                                                                                                             \
    e.mrev = mrev_for(16 + EPIV);                                                                                                 \
    hipLaunchKernelGGL((gemm_bf16_p8_kernel<EPIV, true>), dim3(nb), dim3(512), 0, st, (const bf16_t*)A, lda, (const bf16_t*)W,     \
                       ldw, K, (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, 0, 0, lk, e);                                    \
  } while (0)
  switch (epilogue) {
    case GSL_EPI_STORE: GSL_LL(GSL_EPI_STORE); break;
    case GSL_EPI_BIAS_RES_F32: GSL_CHECK_ARG(bias && res, "bias/res required"); GSL_LL(GSL_EPI_BIAS_RES_F32); break;
    case GSL_EPI_BIAS_RES_F16: e.f16 = 1; [[fallthrough]];
    case GSL_EPI_BIAS_RES_BF16: GSL_CHECK_ARG(bias && res && (ldo % 8) == 0, "bias/res required"); GSL_LL(GSL_EPI_BIAS_RES_BF16); break;
    case GSL_EPI_BIAS_GELU: GSL_CHECK_ARG(bias, "bias required"); GSL_LL(GSL_EPI_BIAS_GELU); break;
    case GSL_EPI_MUL: GSL_CHECK_ARG(aux, "aux required"); GSL_LL(GSL_EPI_MUL); break;
    case GSL_EPI_BIAS_GELU_G8: GSL_CHECK_ARG(bias && (N % 64) == 0, "bias required, N % 64 == 0"); GSL_LL(GSL_EPI_BIAS_GELU_G8); break;
    case GSL_EPI_MUL_G8: GSL_CHECK_ARG(aux && (N % 64) == 0, "aux required, N % 64 == 0"); GSL_LL(GSL_EPI_MUL_G8); break;
    default: return fail(GSL_ERR_ARG, "gsl_gemm_nt_lora: unsupported epilogue%s %ld", "", epilogue);
  }
#undef GSL_LL
  return check_launch("gsl_gemm_nt_lora");
}

// =====================================================================================
// FFN2-dX with its two LoRA-gradient reductions fused into the epilogue (see epilogue_staged_mulgrad).
// Partials [M tiles][N][R] of both gradients, then a two-level fixed-order reduction into the strided gradient views.
// =====================================================================================
constexpr int GF_FAN = 32;
// level 1: thread (4 consecutive outputs, slab) sums GF_FAN consecutive M-tile partials
__global__ __launch_bounds__(256) void mulgrad_reduce1_kernel(const float4* __restrict__ part, float4* __restrict__ part2, int NR4, int ntile,
                                                              long which_stride4, long which_stride4_out) {
  GSL_OP16_KERNEL_ENTRY();
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NR4) return;
  const float4* p = part + (size_t)blockIdx.z * which_stride4;
  const int s0 = blockIdx.y * GF_FAN, s1 = min(ntile, s0 + GF_FAN);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sp = s0; sp < s1; ++sp) {
    const float4 v = p[(size_t)sp * NR4 + idx];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  part2[(size_t)blockIdx.z * which_stride4_out + (size_t)blockIdx.y * NR4 + idx] = a;
}
// level 2: fixed-order sum of the slabs, output strides, optional accumulate; blockIdx.y selects the gradient
__global__ __launch_bounds__(256) void mulgrad_reduce2_kernel(const float* __restrict__ part2, long which_stride, float* G1, long g1sn, long g1sj,
                                                              float* G2, long g2sn, long g2sj, int N, int R, int r, int nslab,
                                                              int accumulate, const float* __restrict__ gscale) {
  GSL_OP16_KERNEL_ENTRY();
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * R) return;
  const int n = idx / R, j = idx % R;
  if (j >= r) return;
  const float* p = part2 + (size_t)blockIdx.y * which_stride;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += p[(size_t)k * N * R + idx];
  if (gscale) s *= gscale[1];      // fp16 operands: the backward ran on gradients multiplied by gscale[0] (a power of two): exact un-scaling
  float* g = blockIdx.y ? (G2 + (size_t)n * g2sn + (size_t)j * g2sj) : (G1 + (size_t)n * g1sn + (size_t)j * g1sj);
  *g = accumulate ? (*g + s) : s;
}
#if GSL_HAS_F32
extern "C" long gsl_gemm_mulgrad_ws_elems(int M, int N, int r) {
  const long R = (r <= 8) ? 8 : 16;
  const long ntile = (M + BM4 - 1) / BM4, nslab = (ntile + GF_FAN - 1) / GF_FAN;
  return 2 * (ntile + nslab) * (long)N * R;
}
#endif
extern "C" int GSL_ENTRY(gsl_gemm_nt_lora_mulgrad)(const void* A, int lda, const void* W, int ldw, int K, const void* P, int ldp, const void* Q,
                                        int ldq, float lora_scale, void* tout, int ldt, int M, int N, const void* aux, void* out,
                                        int ldo, const void* U1, int ldu1, float* G1, long g1sn, long g1sj, const void* Y2, float* G2,
                                        long g2sn, long g2sj, int r, int accumulate, float* ws, int aux_u8, float p_drop, int dtype,
                                        const float* gscale, gsl_stream_t s) {
  GSL_FORWARD_H16(dtype, h16_gsl_gemm_nt_lora_mulgrad(A, lda, W, ldw, K, P, ldp, Q, ldq, lora_scale, tout, ldt, M, N, aux, out, ldo, U1, ldu1, G1, g1sn,
                                                      g1sj, Y2, G2, g2sn, g2sj, r, accumulate, ws, aux_u8, p_drop, dtype, gscale, s));
  GSL_CHECK_ARG(dtype == GSL_OP16, "dtype: bf16 or fp16 operands");
  GSL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "p_drop");
  GSL_CHECK_ARG(M > 0 && N >= 8 && (N % 8) == 0 && K > 0 && (K % 64) == 0, "M>0, N%8==0, K%64==0");
  GSL_CHECK_ARG(A && W && P && Q && out && aux && U1 && G1 && Y2 && G2 && ws, "null operand");
  GSL_CHECK_ARG((lda % 8) == 0 && (ldw % 8) == 0 && (ldp % 8) == 0 && (ldq % 8) == 0 && ldq >= 32 && (ldo % 8) == 0 && ldo >= N &&
                (!tout || ((ldt % 8) == 0 && ldt >= 64)) && (ldu1 % 8) == 0 && ldu1 >= 16,
                "leading dimensions (P [16,K], Q [N,>=32], tout [M,>=64], U1 [M,>=16], out/aux/Y2 [M,ldo])");
  GSL_CHECK_ARG(r >= 1 && r <= 16, "r in [1,16]");
  EpiArgs e;
  e.alpha = 1.0f; e.bias = nullptr; e.res = nullptr; e.aux = aux; e.out = out; e.out2 = nullptr; e.ldo = ldo;
  e.pos = nullptr; e.cls = nullptr; e.T = 0; e.drop = make_drop(0.f, 0, 0); e.M = M; e.N = N;
  e.drop.scale = 1.0f / (1.0f - p_drop);      // only the decode of the 8-bit GELU' codes reads it: no mask is applied in this kernel
  set_launch_knobs(e, false);   // K rotation stays off: every N tile must accumulate t = s A P^T in the same K order (G2 contracts the tile-local t, which has to equal tout bit for bit)
  e.hmT = 0; e.hmH = 0;
  const int R = (r <= 8) ? 8 : 16;
  const int ntile = (M + BM4 - 1) / BM4, nslab = (ntile + GF_FAN - 1) / GF_FAN;
  const size_t NR = (size_t)N * R;
  e.gu1 = (const bf16_t*)U1; e.ldgu1 = ldu1; e.gy2 = (const bf16_t*)Y2; e.gR = R;
  e.gpart1 = ws; e.gpart2 = ws + (size_t)ntile * NR;
  float* part2 = ws + 2 * (size_t)ntile * NR;
  LoraInk lk;
  lk.P = (const bf16_t*)P; lk.ldp = ldp; lk.Q = (const bf16_t*)Q; lk.ldq = ldq; lk.s = lora_scale; lk.tout = (bf16_t*)tout; lk.ldt = ldt;
  const int nb = ntile * ((N + BN4 - 1) / BN4);
  hipStream_t st = as_stream(s);
  e.mrev = mrev_for(30);
  if (aux_u8)
    hipLaunchKernelGGL((gemm_bf16_p8_kernel<GSL_EPI_MUL_G8, true, true>), dim3(nb), dim3(512), 0, st, (const bf16_t*)A, lda, (const bf16_t*)W,
                       ldw, K, (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, 0, 0, lk, e);
  else
    hipLaunchKernelGGL((gemm_bf16_p8_kernel<GSL_EPI_MUL, true, true>), dim3(nb), dim3(512), 0, st, (const bf16_t*)A, lda, (const bf16_t*)W,
                       ldw, K, (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, 0, 0, lk, e);
  int rc = check_launch("gsl_gemm_nt_lora_mulgrad");
  if (rc) return rc;
  const int NR4 = (int)(NR / 4);
  if (nslab == 1) {      // one slab: the second level sums the M tiles itself, in the same order (one launch less in the launch-bound regime)
    hipLaunchKernelGGL(mulgrad_reduce2_kernel, dim3(((int)NR + 255) / 256, 2), dim3(256), 0, st, ws, (long)((size_t)ntile * NR), G1, g1sn,
                       g1sj, G2, g2sn, g2sj, N, R, r, ntile, accumulate, gscale);
    return check_launch("gsl_gemm_nt_lora_mulgrad(reduce)");
  }
  hipLaunchKernelGGL(mulgrad_reduce1_kernel, dim3((NR4 + 255) / 256, nslab, 2), dim3(256), 0, st, (const float4*)ws, (float4*)part2, NR4,
                     ntile, (long)((size_t)ntile * NR / 4), (long)((size_t)nslab * NR / 4));
  hipLaunchKernelGGL(mulgrad_reduce2_kernel, dim3(((int)NR + 255) / 256, 2), dim3(256), 0, st, part2, (long)((size_t)nslab * NR), G1, g1sn,
                     g1sj, G2, g2sn, g2sj, N, R, r, nslab, accumulate, gscale);
  return check_launch("gsl_gemm_nt_lora_mulgrad(reduce)");
}
GSL_OPNS_END
