/* gslora_hip.h — C ABI of libgslora_hip.so: the MI355X (gfx950) kernels behind the GS-LoRA
 * forgetting train step.
 *
 * Boundary contract:
 *  - plain C: pointers + sizes, no C++/torch types. All pointers are DEVICE pointers owned by
 *    the caller (torch tensors' data_ptr()); the library allocates nothing.
 *  - every call is asynchronous on the given hipStream_t (passed as void*; NULL = default stream).
 *  - return 0 on success, negative gsl_status on error; message via gsl_last_error()
 *    (thread-local). Never aborts.
 *  - dtype selects the operand/activation element type: GSL_F32 (parity mode, exact-f32 kernels),
 *    GSL_BF16 or GSL_F16 (speed mode: 16-bit operands, f32 accumulate on MFMA; "bf16" in the comments below means either 16-bit
 *    format unless a comment says otherwise). LayerNorm statistics, biases,
 *    LoRA master weights, losses and optimizer state are always f32; the residual stream and its
 *    gradient are f32 or — in speed mode, per call (x_dtype / stream_dtype) — bf16.
 *
 * Each entry point names the reference code it replaces (paths relative to bjzhb666/GS-LoRA).
 * The reference has no FFI of its own (it is pure PyTorch); the "binding a maintainer would add"
 * is the ctypes stub in gs-lora_amd/gslora_hip/_lib.py, shown in INTEGRATION.md.
 */
#ifndef GSLORA_HIP_H
#define GSLORA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the entry points declared here are exported. */
#define GSL_API __attribute__((visibility("default")))

typedef void* gsl_stream_t;

/* GSL_F16 (round 5): the speed mode with IEEE fp16 operands — the same kernels, bytes and MFMA rate as GSL_BF16 (v_mfma_f32_16x16x32_f16),
 * an 11-bit significand instead of 8. Activations saturate at +-65504 on store (NaN / Inf stay visible); the backward runs on gradients
 * multiplied by a power-of-two loss scale chosen on the device by gsl_head_bwd (gscale = {S, 1/S}) which the LoRA-gradient reductions
 * divide out again, exactly. As an x_dtype it is also the forward residual stream format of the GSL_BF16 mode (round 4). */
enum gsl_dtype { GSL_F32 = 0, GSL_BF16 = 1, GSL_F16 = 2 };

enum gsl_status {
  GSL_OK = 0,
  GSL_ERR_ARG = -1,      /* bad shape / alignment / unsupported size */
  GSL_ERR_LAUNCH = -2,   /* hipLaunch / runtime error */
  GSL_ERR_UNSUPPORTED = -3
};

/* GEMM epilogues (gsl_gemm_nt). acc = alpha * (A1*W1^T + A2*W2^T)  */
enum gsl_epilogue {
  GSL_EPI_STORE = 0,        /* out[dtype]  = acc (+ bias); out2 (nullable; 16-bit operands, 16 <= N <= 128, no bias): a COMPACT [M,16] copy
                               (row stride 16) of output columns 0..15 — the 16-column operand form of a LoRA down-projection for
                               gsl_gemm_nt_lora_mulgrad's U1 (32 rows = one contiguous 1 KB read)                                   */
  GSL_EPI_BIAS_RES_F32 = 1, /* outf32      = dropout(acc + bias) + res                          */
  GSL_EPI_BIAS_GELU = 2,    /* out[dtype]  = dropout(gelu(acc+bias)); out2[dtype] = gelu'(acc+bias)*dropmask */
  GSL_EPI_MUL = 3,          /* out[dtype]  = acc * aux[dtype]                                   */
  GSL_EPI_PATCH = 4,        /* outf32      = dropout((tok==0 ? cls : acc + bias) + pos[tok]),  tok = m % T */
  GSL_EPI_STORE_F32 = 5,    /* outf32      = acc (+ bias)                                       */
  GSL_EPI_STORE_QKV_HM = 6, /* bf16 only: STORE of a QKV projection (N = 3*H*64, rows m = b*T + t, T = tokens per image) into the
                               head-major layout [B][H][3][T][64] that the attention entry points read with qkv_layout = 1 */
  GSL_EPI_BIAS_RES_BF16 = 7,/* bf16 only, the forward residual stream carried in bf16: out[bf16] = bf16(dropout(acc + bias) + f32(res[bf16]))
                               — f32 arithmetic on the f32 accumulator, one rounding on store; same dropout mask as BIAS_RES_F32 */
  GSL_EPI_PATCH_BF16 = 8,   /* bf16 only: PATCH with a bf16 output */
  GSL_EPI_MUL_G8 = 9,       /* bf16 only: MUL with aux = the 8-bit GELU' code tensor (slab-major, see below) written by BIAS_GELU_G8;
                               p_drop = the dropout rate of the forward that wrote it (decode scale 1/(1-p); no mask is applied here) */
  GSL_EPI_BIAS_RES_F16 = 11,/* bf16 only: BIAS_RES_BF16 with the forward residual stream (res in, out) in IEEE fp16 (clamped to +-65504 on store) */
  GSL_EPI_PATCH_F16 = 12,   /* bf16 only: PATCH with an fp16 output */
  GSL_EPI_STORE_LN = 13,    /* STORE with a CONSUMER-SIDE LayerNorm (reference vit_face.py:316-323 PreNorm + :358-360 to_qkv; modified_VIT.py: ln_1 -> in_proj): A1 is the RAW
                               residual stream x (not LN(x)), W1 the weight with gamma folded in (W'[n,k] = W[n,k] gamma[k], operand format), and
                               out[dtype] = rstd[m] * (acc - mean[m] * c[n]) + d[n]   with pos = mean [M], cls = rstd [M] (gsl_layernorm_fwd with
                               y = NULL), aux = c [N] = rowsum_k W' (f32, of the ROUNDED W'), bias = d [N] = W beta (+ the layer's bias) (f32).
                               One pass over x replaces LayerNorm's read + write and the GEMM reads the stream itself. alpha = 1, no out2. T > 0: row m takes
                               mean[m*T], rstd[m*T] — the statistics of every T-th row of a larger tensor (A1 = its cls rows at lda1 = T*D). */
  GSL_EPI_STORE_QKV_HM_LN = 14, /* the same with the head-major copy-out of GSL_EPI_STORE_QKV_HM */
  GSL_EPI_BIAS_GELU_G8 = 10 /* bf16 only: BIAS_GELU whose second output is the 8-bit fixed-point code of gelu'(acc+bias)*dropmask:
                               q = round(gelu' * keep * 200 + 26), decoded as (q - 26) * 0.005 / (1 - p); gelu' lies in [-0.129, 1.129]:
                               absolute error <= 0.0025/(1-p), a dropped element decodes to exactly 0. out2 is M*N bytes in SLAB-MAJOR
                               order [N/64][M][64] (element (m, n) at ((n/64)*M + m)*64 + n%64; N % 64 == 0): a tensor private to this
                               epilogue and its reader, laid out so that both touch consecutive memory */
};

GSL_API int gsl_version(void);
GSL_API const char* gsl_last_error(void);

/* ---- K1 patch gather: einops 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (vit_face.py:530).
 * img f32 [B,C,H,W] -> out[dtype] [B*T, p*p*C], T = 1 + (H/p)*(W/p); row b*T (cls slot) is zero. */
GSL_API int gsl_patchify(const float* img, void* out, int B, int C, int H, int W, int p, int dtype, gsl_stream_t s);

/* ---- K3/K5/K6/K7/K8 dense NT GEMM with an optional second K segment (the LoRA rank-r term)
 * and a fused epilogue. Replaces F.linear + loralib.Linear.forward (vit_face.py:330-334,349-356)
 * and their autograd dX.
 *   A1 [M,K1] (lda1), W1 [N,K1] (ldw1); A2 [M,K2] (lda2), W2 [N,K2] (ldw2)  — all `dtype`;
 *   K1 % 64 == 0, K2 % 64 == 0 (K2 may be 0). bias/pos/cls f32; res f32 (bf16 for GSL_EPI_BIAS_RES_BF16). out/out2/aux per epilogue.
 *   dropout: p_drop in [0,1); mask = hash(seed, site, m*N+n) (see gsl_dropout_keep in DESIGN.md).
 *   Every (seed, site) pair of this ABI: when bit 31 of `site` (GSL_SEED_ON_DEVICE) is set, `seed` is not the value but a device
 *   pointer to a uint64 holding it — the kernels load it, so a captured HIP graph replays with the value current at replay time. */
#define GSL_SEED_ON_DEVICE 0x80000000u
GSL_API int gsl_gemm_nt(const void* A1, int lda1, const void* W1, int ldw1, int K1,
                const void* A2, int lda2, const void* W2, int ldw2, int K2,
                int M, int N, int dtype, int epilogue, float alpha,
                const float* bias, const void* res, const void* aux,
                void* out, void* out2, int ldo,
                const float* pos, const float* cls, int T,
                float p_drop, uint64_t seed, uint32_t site, gsl_stream_t s);

/* Same GEMM with the LoRA rank-r term produced INSIDE the kernel (bf16 only; replaces the two-launch form
 * t = s*(A P^T) [gsl_gemm_nt, N=64] ; out = A W^T + t Q^T [gsl_gemm_nt with a K segment], and saves one full read of A):
 *   out = epilogue(A*W^T + t*Q^T),  t = lora_scale * (A*P^T)
 *   A [M,K] (lda), W [N,K] (ldw), P [16,K] (ldp; rows >= r zero), Q [N,32] (ldq >= 32; cols >= r zero), K % 64 == 0.
 *   tout (nullable) [M, ldt >= 64] receives t in bf16, zero padded to 64 columns (input of gsl_lora_grad).
 * Epilogues: STORE, BIAS_RES_F32, BIAS_RES_BF16, BIAS_GELU, BIAS_GELU_G8, MUL, MUL_G8. */
GSL_API int gsl_gemm_nt_lora(const void* A, int lda, const void* W, int ldw, int K,
                     const void* P, int ldp, const void* Q, int ldq, float lora_scale, void* tout, int ldt,
                     int M, int N, int dtype, int epilogue,
                     const float* bias, const void* res, const void* aux, void* out, void* out2, int ldo,
                     float p_drop, uint64_t seed, uint32_t site, gsl_stream_t s);

/* The MUL form of gsl_gemm_nt_lora (FFN2-dX: out = (A W^T + t Q^T) * aux, bf16) with the two LoRA-gradient reductions that consume
 * its tiles fused into the epilogue instead of two gsl_lora_grad launches that re-read [M,N] tensors from HBM:
 *   G1[n*g1sn + j*g1sj] (+)= sum_m out[m,n] * U1[m,j]      (dB of the up-projection adapter: loralib autograd of vit_face.py:330)
 *   G2[n*g2sn + j*g2sj] (+)= sum_m Y2[m,n] * t[m,j]        (dA of the down-projection adapter; Y2 = the saved FFN hidden activation)
 * U1 [M, ldu1 >= 16] bf16 (columns r..15 zero or discarded), Y2 / aux / out [M,N] bf16 with row stride ldo, N % 8 == 0.
 * ws f32 >= gsl_gemm_mulgrad_ws_elems(M, N, r). Reductions are fixed-order (bit-reproducible).
 * aux_u8 != 0: aux is the 8-bit GELU' code tensor of GSL_EPI_BIAS_GELU_G8 (slab-major [N/64][M][64], N % 64 == 0) and p_drop the
 * dropout rate of that forward. dtype: GSL_BF16 or GSL_F16 (the operand format of every 16-bit tensor of the call).
 * gscale (nullable): device pointer to {S, 1/S} written by gsl_head_bwd — the reductions are multiplied by 1/S on the way out (the
 * operands carry loss-scaled gradients; a power of two, so the un-scaling is exact). */
GSL_API long gsl_gemm_mulgrad_ws_elems(int M, int N, int r);
GSL_API int gsl_gemm_nt_lora_mulgrad(const void* A, int lda, const void* W, int ldw, int K,
                             const void* P, int ldp, const void* Q, int ldq, float lora_scale, void* tout, int ldt,
                             int M, int N, const void* aux, void* out, int ldo,
                             const void* U1, int ldu1, float* G1, long g1sn, long g1sj,
                             const void* Y2, float* G2, long g2sn, long g2sj,
                             int r, int accumulate, float* ws, int aux_u8, float p_drop, int dtype, const float* gscale, gsl_stream_t s);

/* ---- K2 LayerNorm (nn.LayerNorm, vit_face.py:316-323, 498-500). x — the residual stream — is `x_dtype` (f32; bf16 when the bf16
 * speed mode carries the forward stream in bf16), rows of length D at stride x_row_stride (elements); y[dtype] [M,D]; mean/rstd f32 [M].
 * D in {64,128,256,512,768,1024}. y may be NULL: row statistics only (the input of GSL_EPI_STORE_LN, whose GEMM normalises in its epilogue). */
GSL_API int gsl_layernorm_fwd(const void* x, long x_row_stride, const float* gamma, const float* beta, float eps,
                      void* y, float* mean, float* rstd, int M, int D, int dtype, int x_dtype, gsl_stream_t s);
/* LayerNorm forward + the LoRA down-projection that reads its output, in one pass (bf16 mode): y = LN(x) [M,D] bf16 and
 * u = alpha * y P^T for the rank-r adapter of the layer that consumes y (FFN1's lora_A, vit_face.py:330 + loralib Linear.forward):
 * P [>= 16, D] bf16 at row stride ldp (rows j < r = lora_A[j,:], the rest zero), u [M,64] bf16 dense with columns >= 16 written as zero —
 * the LoRA K segment of gsl_gemm_nt. Replaces gsl_layernorm_fwd + a skinny gsl_gemm_nt that re-read y. x bf16 at x_row_stride; D in {512, 768}. */
GSL_API int gsl_layernorm_fwd_lora(const void* x, long x_row_stride, const float* gamma, const float* beta, float eps, void* y,
                           float* mean, float* rstd, int M, int D, const void* P, int ldp, float alpha, void* u, gsl_stream_t s);
/* dx = dres + LN'(dy) ; dxb[dtype] = dx * dropmask(site) (nullable). dy is `dtype` [M,D] (dense). dres / dx — the residual-GRADIENT
 * stream — are `stream_dtype`: f32, or bf16 when dtype is bf16 (speed mode: the stream is re-read and re-written by every LayerNorm
 * backward of the chain). x (the saved forward stream) is `x_dtype`.
 * dres/dx rows are io_row_stride elements apart (0 -> D; dres may alias dx: in-place update); the dropout counter of element (row, d) is
 * row*drop_row_stride + d (0 -> D). dxb is always dense [M,D].
 * dres_cls_T > 0: dres is COMPACT — it holds only the rows of the cls tokens ([M / dres_cls_T] rows, io_row_stride apart); row m
 * receives dres[m / dres_cls_T] when m % dres_cls_T == 0 and nothing otherwise, and dx is written dense [M,D]. (The stream gradient
 * leaving the cls-row-only backward of the last block is exactly zero off the cls rows: no zero-filled tensor is written or read.)
 * gmax (nullable; the overflow guard of the loss-scaled fp16 backward, round 6): device f32, raised (atomic max) to the largest |value| this call
 * read in dy or stored in dx — gscale + 2 of gsl_head_bwd. Every gradient of the chain passes a LayerNorm backward; a saturated 16-bit store
 * anywhere upstream (an operand of +-65504) or in the stream shows here as gmax >= 65504. */
GSL_API int gsl_layernorm_bwd(const void* dy, const void* x, long x_row_stride, const float* gamma,
                      const float* mean, const float* rstd, const void* dres,
                      void* dx, long io_row_stride, void* dxb, int M, int D, int dtype, int stream_dtype, int x_dtype,
                      float p_drop, uint64_t seed, uint32_t site, long drop_row_stride, int dres_cls_T, float* gmax, gsl_stream_t s);

/* ---- K4 attention, head_dim 64, no mask, softmax(QK^T*scale)V (vit_face.py:358-376).
 * qkv_layout (the INPUT qkv): 0 = token-major qkv[dtype] [B*T, 3*H*64] (q|k|v, each 'b n (h d)', as the reference's to_qkv output),
 * 1 = head-major [B][H][3][T][64] (bf16 kernels only; written by gsl_gemm_nt's GSL_EPI_STORE_QKV_HM): every panel row is a full
 * 128-byte line next to its neighbours. Outputs are token-major in both cases: o[dtype] [B*T, H*64], lse f32 [B,H,T]. */
GSL_API int gsl_attention_fwd(const void* qkv, void* o, float* lse, int B, int T, int H, float scale, int dtype, int qkv_layout, gsl_stream_t s);
/* dqkv[dtype] [B*T,3*H*64] (token-major, always); delta_ws f32 [B,H,T] scratch. */
GSL_API int gsl_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv,
                      float* delta_ws, int B, int T, int H, float scale, int dtype, int qkv_layout, gsl_stream_t s);
/* The last transformer block when the head pools the cls token (vit_face.py:540): everything after its attention is token-wise, so
 * only the cls query's attention output is ever consumed. Forward of that one query row per (image, head) against the full K / V
 * panels: o_cls[dtype] [B, H*64], lse_cls f32 [B, H].
 * qkv_layout 0 / 1 as above (q = token 0 of the qkv tensor, q_cls ignored); 2: the block's projection computed Q for the cls rows only —
 * `qkv` is kv[dtype] token-major [B*T, 2*H*64] (k | v) and q_cls[dtype] [B, H*64] holds the queries. */
GSL_API int gsl_attention_fwd_cls(const void* qkv, const void* q_cls, void* o_cls, float* lse_cls, int B, int T, int H, float scale,
                                  int dtype, int qkv_layout, gsl_stream_t s);
/* Its backward: only the cls query carries an output gradient. d_o_cls[dtype] [B, H*64] is dO of the cls rows;
 * cls_compact != 0: o / lse are the [B, H*64] / [B, H] outputs of gsl_attention_fwd_cls, 0: the full tensors of gsl_attention_fwd
 * ([B*T, H*64] / [B, H, T]; the cls rows are read). qkv_layout 0 / 1: writes the full dqkv [B*T,3*H*64] (dQ rows of the other tokens
 * = 0; dq_cls unused); 2: writes dkv [B*T, 2*H*64] into `dqkv` and the query gradients into dq_cls [B, H*64]. */
GSL_API int gsl_attention_bwd_cls(const void* qkv, const void* q_cls, const void* o, const void* d_o_cls, const float* lse, void* dqkv,
                                  void* dq_cls, int B, int T, int H, float scale, int dtype, int qkv_layout, int cls_compact, gsl_stream_t s);

/* ---- K9 LoRA gradient (skinny, reduction over M rows): G[n*gsn + j*gsj] (+)= sum_m Y[m,n] * U[m,j]
 * Y[dtype] [M,N] with row stride ldy >= N elements (a column block of a wider tensor is allowed), U[dtype] [M,ldu] (first r columns
 * used, r <= 16; columns r..15 must be readable zeros or belong to other adapters whose products are discarded).
 * ws f32 >= gsl_lora_grad_ws_elems(). gscale (nullable): device {S, 1/S} of a loss-scaled backward (see gsl_head_bwd): the sums are
 * multiplied by 1/S before they are stored / accumulated. */
GSL_API long gsl_lora_grad_ws_elems(int M, int N, int r);
GSL_API int gsl_lora_grad(const void* Y, long ldy, const void* U, int ldu, float* G, long gsn, long gsj,
                  int M, int N, int r, int dtype, int accumulate, float* ws, const float* gscale, gsl_stream_t s);
/* The same for n reductions in two launches per 24 descriptors (the launch-bound regime: a few-shot step runs 24 of them). 16-bit operands
 * only (dtype GSL_BF16 or GSL_F16, one format per call; gscale as above), N % 256 == 0, 16-byte aligned rows; any M >= 1. `descs` is a HOST array — the launches carry the descriptors by value, so a captured
 * HIP graph keeps them. The G of one call must not overlap. ws: gsl_lora_grad_batch_ws_elems(descs, n) floats (-1: invalid descriptor). */
typedef struct gsl_lgrad_desc {
  const void* Y; long ldy; const void* U; int ldu; int M; int N; int r; int accumulate; int pad_; float* G; long gsn, gsj;
} gsl_lgrad_desc;
GSL_API long gsl_lora_grad_batch_ws_elems(const gsl_lgrad_desc* descs, int n);
GSL_API int gsl_lora_grad_batch(const gsl_lgrad_desc* descs, int n, float* ws, int dtype, const float* gscale, gsl_stream_t s);


/* ---- K10 head: cls pool + LayerNorm + CosFace (vit_face.py:540-546, 171-208; s=64, m=0.35).
 * linear_head != 0 selects the ViT-B/16 path instead (modified_VIT.py:32-38): logits = emb * W^T + head_bias, with Wn = W
 * un-normalised and cos_s = 1 on the backward side. */
GSL_API int gsl_cosface_prep(const float* W, float* Wn, int C, int D, gsl_stream_t s);   /* Wn = F.normalize(W) */
GSL_API int gsl_head_fwd(const void* x, int x_dtype, int T, const float* gamma, const float* beta, float eps,
                 const float* Wn, const int64_t* label, float* emb, float* mean, float* rstd,
                 float* logits, int B, int D, int C, float cos_s, float cos_m,
                 const float* head_bias, int linear_head, int pool_mean, gsl_stream_t s);
/* pool_mean = 0: the head pools token 0 (pool='cls'); 1: the mean over the T tokens (pool='mean', vit_face.py:540).
 * x is `x_dtype` (the residual stream, see gsl_layernorm_fwd).
 * dlogits [B,C] / demb [B,D] nullable. dx [B*T,D]: with pool='cls' the cls rows get the gradient and the others are zeroed,
 * with pool='mean' every token row gets d pooled / T. dxb[dtype] = dx * dropmask(site) (nullable). dx is `stream_dtype` (see
 * gsl_layernorm_bwd). compact != 0 (pool='cls' only): dx / dxb are [B,D], the cls rows alone — nothing is zero-filled; the dropout
 * counters stay those of the dense tensor.
 * gscale != NULL (fp16 operands): LOSS-SCALED backward. The kernel runs twice: pass 1 writes max|d loss / d stream| of every image to
 * amax_ws [B], pass 2 picks the power of two S with S * max in [2^10, 2^11), stores dx / dxb multiplied by S and publishes
 * gscale[0..1] = {S, 1/S} on the device (no host sync; HIP-graph safe). Every kernel downstream is linear in the gradient; the
 * LoRA-gradient reductions take gscale and divide S out.
 * Overflow guard (round 6; GradScaler semantics without a host sync). gscale is f32 [4] that PERSISTS across steps (zero it once):
 *   [2] = the largest |scaled gradient| the LayerNorm backwards of this backward saw (gsl_layernorm_bwd's gmax; reset to 0 here),
 *   [3] = the exponent E in use: S * max lands in [2^(E-1), 2^E).
 * Pass 1 reads the PREVIOUS backward's [2]: >= 65504 (a 16-bit store saturated) or non-finite -> E drops by 2 (floor 4); below 2^9 with E
 * under target_exp -> E grows by 1; an E outside [4, 15] (the zeroed buffer) -> target_exp. target_exp: 0 = the default 11 (32x headroom at the head;
 * the largest gradient operand of a depth-6 chain measured 1.2x the head's). gsl_adamw_flat / _dev skip the update of a step whose [2] saturated. */
GSL_API int gsl_head_bwd(const float* dlogits, const float* demb, const void* x, int x_dtype, int T, const float* gamma,
                 const float* mean, const float* rstd, const float* emb, const float* Wn,
                 void* dx, void* dxb, int B, int D, int C, float cos_s, int dtype, int stream_dtype,
                 float p_drop, uint64_t seed, uint32_t site, int linear_head, int pool_mean, int compact,
                 float* gscale, float* amax_ws, int target_exp, gsl_stream_t s);

/* ---- K11 cross entropy (mean) + top-1 (engine_cl.py:65-78, util/utils.py:354-368).
 * out2 f32 [2] = { sum_i CE_i , #correct }; row_ws f32 [2*B] scratch (per-row loss / hit, summed in a fixed order). */
GSL_API int gsl_ce_fwd(const float* logits, const int64_t* labels, float* out2, float* row_ws, int B, int C, gsl_stream_t s);
/* dlogits (+)= coef[0] * scale * (softmax - onehot) ; coef is a DEVICE scalar (no host sync). */
GSL_API int gsl_ce_bwd(const float* logits, const int64_t* labels, const float* coef, float scale,
               float* dlogits, int B, int C, int accumulate, gsl_stream_t s);

/* ---- K13 prototype KL (engine_cl.py:571-603): out1[0] = sum_i KL(softmax(proto[y_i]) || softmax(emb_i)). */
GSL_API int gsl_proto_kl_fwd(const float* emb, const int64_t* labels, const float* proto, float* out1, float* row_ws /*[B]*/,
                     int B, int D, int C, gsl_stream_t s);
GSL_API int gsl_proto_kl_bwd(const float* emb, const int64_t* labels, const float* proto, const float* coef,
                     float scale, float* demb, int B, int D, int C, int accumulate, gsl_stream_t s);

/* ---- scalar tail of the step (engine_cl.py:65-125, single process): from the batch SUMS of the kernels above
 *   total = beta*relu(BND - ce_f_sum/n_f) + ce_r_sum/n_r + alpha*structure + w_f*relu(BND_pro - kl_f_sum/n_f) + w_r*kl_r_sum/n_r
 * meters8 = [beta*loss_forget, loss_remain, total, alpha*structure, top1_forget %, top1_remain %, proto_f, proto_r],
 * coefs5 = d total / d {ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure} (relu'(0) = 0). kl_* / structure nullable (term absent).
 * All inputs and outputs are device scalars / small device arrays: no host sync. */
GSL_API int gsl_loss_combine(const float* ce_r_sum, const float* ce_f_sum, const float* kl_f_sum, const float* kl_r_sum,
                     const float* structure, const float* hit_r, const float* hit_f, float n_r, float n_f,
                     float beta, float BND, float alpha, float w_f, float w_r, float BND_pro,
                     float* total, float* meters8, float* coefs5, gsl_stream_t s);

/* Data-parallel form of the scalar tail (SURVEY 8(e) collective C2; reference semantics train_own_forget_cl.py:494-497, engine_cl.py:78,99):
 * pack8 = the sum-all-reduced [ce_r_sum, ce_f_sum, hit_r, hit_f, n_r, n_f, kl_f_sum, kl_r_sum] — global batch sums and sizes, all on the
 * device. Same outputs as gsl_loss_combine; coefs5 are the derivatives with respect to THIS rank's local sums (= those of the global sums). */
GSL_API int gsl_loss_combine_pack(const float* pack8, const float* structure, int has_proto, float beta, float BND, float alpha, float w_f,
                          float w_r, float BND_pro, float* total, float* meters8, float* coefs5, gsl_stream_t s);
/* The loss section of a single-process step in ONE launch (launch-bound batches): per-row CE / top-1 of logits [N,C] and prototype KL of
 * emb [N,D] against proto [Cp,D] (emb NULL: no prototype term), their sums over the remain rows [0,nr) and the forget rows [nr,N), the scalar
 * tail of gsl_loss_combine (out14 = total, meters8, coefs5) and the backward of all of it for an upstream gradient of 1: dlogits [N,C],
 * demb [N,D]. Coefficients and gradients bit-identical to (meters within an ulp of) gsl_ce_fwd + gsl_proto_kl_fwd + gsl_loss_combine + gsl_ce_bwd + gsl_proto_kl_bwd on the two row
 * ranges. 0 < nr < N <= gsl_loss_tail_max_rows(). structure (nullable): device scalar, the group-lasso value. */
GSL_API int gsl_loss_tail_max_rows(void);
GSL_API int gsl_loss_tail(const float* logits, const int64_t* labels, int N, int nr, int C, const float* emb, const float* proto, int D,
                  int Cp, const float* structure, float beta, float BND, float alpha, float w_f, float w_r, float BND_pro,
                  float* out14, float* dlogits, float* demb, gsl_stream_t s);



/* ---- K12 group-lasso norms over a flat LoRA buffer (engine_cl.py:349-432, util/cal_norm.py:4-146).
 * tensor t = flat[toff[t] .. +tnumel[t]) belongs to group tgroup[t] (tables on device, int64/int64/int32).
 * Outputs (f32 unless noted): tensor_sumsq[ntensors], group_norm[ngroups] = sqrt(sum sumsq),
 * cal_norm[ngroups] = sum sqrt(sumsq) (cal_norm.py 'L2'), loss[1] = sum_g group_norm,
 * mask u8[ngroups] = group_norm > tau.  partial_ws f32 [ntensors*GSL_NORM_SPLIT]. */
#define GSL_NORM_SPLIT 8
GSL_API int gsl_group_norms_fwd(const float* flat, const int64_t* toff, const int64_t* tnumel, const int32_t* tgroup,
                        int ntensors, int ngroups, float tau, float* partial_ws, float* tensor_sumsq,
                        float* group_norm, float* cal_norm, float* loss, uint8_t* mask, gsl_stream_t s);
/* gradflat[i] += coef[0]*scale * flat[i] / group_norm[g(i)]  (0 where group_norm == 0). */
GSL_API int gsl_group_norms_bwd(const float* flat, const int64_t* toff, const int64_t* tnumel, const int32_t* tgroup,
                        int ntensors, const float* group_norm, const float* coef, float scale,
                        float* gradflat, gsl_stream_t s);

/* ---- K14 fused AdamW over a flat buffer (torch.optim.AdamW as timm.create_optimizer builds it,
 * train_own_forget_cl.py:811-813): decoupled wd, bias correction, step >= 1. */
GSL_API int gsl_adamw_flat(float* p, const float* g, float* m, float* v, long n,
                   float lr, float beta1, float beta2, float eps, float wd, int step, const float* guard, gsl_stream_t s);
/* guard (nullable): device f32 — the overflow guard of a loss-scaled fp16 backward (gscale + 2 of gsl_head_bwd). The update is SKIPPED
 * (p, m, v untouched; torch.cuda.amp.GradScaler.step semantics) when *guard >= 65504 or is not finite: a gradient of this step was clipped. */
/* HIP-graph form of the same update: the step count t (>= 1, int64) and the learning rate (f32) are read from device memory,
 * so a captured graph of the whole forgetting step replays with fresh values (bias corrections 1 - beta^t in f64 in-kernel). */
GSL_API int gsl_adamw_flat_dev(float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float beta1, float beta2,
                       float eps, float wd, const int64_t* step_dev, const float* guard, gsl_stream_t s);

/* ---- helpers: f32 -> dtype casts for the frozen-weight caches and padded LoRA operands. */
GSL_API int gsl_cast(const float* in, void* out, long n, int dtype, gsl_stream_t s);
/* out[dtype] [C,R] = in[R,C]^T */
GSL_API int gsl_transpose_cast(const float* in, void* out, int R, int C, int dtype, gsl_stream_t s);
/* out[dtype] [rows_out, ld_out] zero-padded copy: out[i, j] = scale * in[i*si + j*sj] for i<rows, j<cols. */
GSL_API int gsl_pack_pad(const float* in, long si, long sj, int rows, int cols, float scale,
                 void* out, int rows_out, int ld_out, int dtype, gsl_stream_t s);
/* The same for n packs in one launch. descs_dev: device array of n descriptors (all outputs share `dtype`); max_elems =
 * max over descriptors of rows_out * ld_out. The table is built once by the host and reused every step (HIP-graph friendly). */
typedef struct gsl_pack_desc {
  const float* in; long si, sj; int rows, cols; float scale; int pad_; void* out; int rows_out, ld_out;
} gsl_pack_desc;
GSL_API int gsl_pack_pad_batch(const gsl_pack_desc* descs_dev, int n, long max_elems, int dtype, gsl_stream_t s);

/* dropout keep-mask as the kernels compute it (for tests): keep[i] = 1/0 for element index i. */
GSL_API int gsl_dropout_mask(uint8_t* keep, long n, float p_drop, uint64_t seed, uint32_t site, gsl_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* GSLORA_HIP_H */
