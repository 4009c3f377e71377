// gsl_h16.h — internal: the fp16-operand builds of the double-compiled translation units (gemm.hip, attention.hip; see gsl_common.h,
// "the 16-bit matrix-core operand format"). The exported entry point of include/gslora_hip.h forwards to h16_<name> when
// dtype == GSL_F16; the h16_ symbols are hidden (-fvisibility=hidden: linkable inside libgslora_hip.so, not exported).
#pragma once
#include "gsl_common.h"

extern "C" {
int h16_gsl_gemm_nt(const void* A1, int lda1, const void* W1, int ldw1, int K1, const void* A2, int lda2, const void* W2, int ldw2, int K2,
                    int M, int N, int dtype, int epilogue, float alpha, const float* bias, const void* res, const void* aux, void* out,
                    void* out2, int ldo, const float* pos, const float* cls, int T, float p_drop, uint64_t seed, uint32_t site, gsl_stream_t s);
int h16_gsl_gemm_nt_lora(const void* A, int lda, const void* W, int ldw, int K, const void* P, int ldp, const void* Q, int ldq,
                         float lora_scale, void* tout, int ldt, int M, int N, int dtype, int epilogue, const float* bias, const void* res,
                         const void* aux, void* out, void* out2, int ldo, float p_drop, uint64_t seed, uint32_t site, gsl_stream_t s);
int h16_gsl_gemm_nt_lora_mulgrad(const void* A, int lda, const void* W, int ldw, int K, const void* P, int ldp, const void* Q, int ldq,
                                 float lora_scale, void* tout, int ldt, int M, int N, const void* aux, void* out, int ldo, const void* U1,
                                 int ldu1, float* G1, long g1sn, long g1sj, const void* Y2, float* G2, long g2sn, long g2sj, int r,
                                 int accumulate, float* ws, int aux_u8, float p_drop, int dtype, const float* gscale, gsl_stream_t s);
int h16_gsl_attention_fwd(const void* qkv, void* o, float* lse, int B, int T, int H, float scale, int dtype, int qkv_layout, gsl_stream_t s);
int h16_gsl_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, float* delta_ws, int B, int T,
                          int H, float scale, int dtype, int qkv_layout, gsl_stream_t s);
int h16_gsl_attention_fwd_cls(const void* qkv, const void* q_cls, void* o_cls, float* lse_cls, int B, int T, int H, float scale, int dtype,
                              int qkv_layout, gsl_stream_t s);
int h16_gsl_attention_bwd_cls(const void* qkv, const void* q_cls, const void* o, const void* d_o_cls, const float* lse, void* dqkv,
                              void* dq_cls, int B, int T, int H, float scale, int dtype, int qkv_layout, int cls_compact, gsl_stream_t s);
}

// first statement of a dtype-dispatching entry point (plain compile only: the fp16 build IS the target)
#ifdef GSL_OP_F16
#define GSL_FORWARD_H16(dtype, call)
#define GSL_HAS_F32 0
#else
#define GSL_FORWARD_H16(dtype, call) do { if ((dtype) == GSL_F16) return call; } while (0)
#define GSL_HAS_F32 1
#endif
