// Entry points that are NOT part of the public C ABI (include/ddmp_hip.h) any more (round 6): the typed float32 / bfloat16 forms behind
// the dtype-generic dispatchers of csrc/dispatch.hip (the host mirror binds the generic forms only) and the per-thread "next call"
// records that the ABI-3 `_o` entry points set and clear around their own call (csrc/opts.hip).  Built with hidden visibility:
// none of these is exported from libddmp_hip.so.
#pragma once
#include "../../include/ddmp_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
int ddmp_bf16_to_f32(const uint16_t* in, float* out, int64_t n, ddmp_stream stream);
int ddmp_graph_tables(const ddmp_graph* g, const int32_t** rowptr, const int32_t** col, const float** dinv);
int ddmp_bn_bwd_apply_f32(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, float* dY, int64_t lddy,
                          int64_t n_rows, int C, const float* scale, const float* shift, const float* c1,
                          const float* c0, float slope, double* dbias_sums /*[C]*/, void* workspace,
                          size_t workspace_bytes, ddmp_stream stream);
int ddmp_gemm_nn_bf16(const uint16_t* A, int64_t lda, const float* W, int64_t ldw, uint16_t* Y, int64_t ldy,
                      int64_t n_rows, int M, int K, void* workspace, size_t workspace_bytes, ddmp_stream stream);
int ddmp_gemm_nn_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy,
                     int64_t n_rows, int M, int K, void* workspace /*nullable*/, size_t workspace_bytes,
                     ddmp_stream stream);
int ddmp_gemm_nt_bf16(const uint16_t* A, int64_t lda, const float* W, int64_t ldw, uint16_t* Y, int64_t ldy,
                      int64_t n_rows, int K, int M, const float* bias, const float* pro_scale, const float* pro_shift,
                      float slope, void* workspace, size_t workspace_bytes, ddmp_stream stream);
size_t ddmp_gemm_rows_bf16_workspace_bytes(int K, int M);
int ddmp_gemm_tn_bf16(const uint16_t* G, int64_t ldg, const uint16_t* Z, int64_t ldz, float* dW, int64_t lddw,
                      int64_t n_rows, int M, int K, const float* pro_scale, const float* pro_shift, float slope,
                      void* workspace, size_t workspace_bytes, ddmp_stream stream);
size_t ddmp_gemm_tn_bf16_workspace_bytes(int64_t n_rows, int M, int K);
int ddmp_gemm_tn_f32(const float* G, int64_t ldg, const float* Z, int64_t ldz, float* dW, int64_t lddw,
                     int64_t n_rows, int M, int K, const float* pro_scale, const float* pro_shift,
                     float slope, void* workspace, size_t workspace_bytes, ddmp_stream stream);
int ddmp_head_bwd_bf16(const uint16_t* Y, int64_t ldy, int64_t n_rows, const float* scale, const float* shift,
                       float slope, const float* W1, const float* b1, const float* W2, const float* b2, int kind,
                       const float* dout, uint16_t* dZ, int64_t lddz, float* dW1, float* db1, float* dW2, float* db2,
                       void* workspace, size_t workspace_bytes, ddmp_stream stream);
int ddmp_head_bwd_f32(const float* Y, int64_t ldy, int64_t n_rows, const float* scale, const float* shift,
                      float slope, const float* W1, const float* b1, const float* W2, const float* b2,
                      int kind, const float* dout, float* dZ, int64_t lddz, float* dW1, float* db1,
                      float* dW2, float* db2, void* workspace, size_t workspace_bytes, ddmp_stream stream);
int ddmp_head_fwd_bf16(const uint16_t* Y, int64_t ldy, int64_t n_rows, const float* scale, const float* shift,
                       float slope, const float* W1, const float* b1, const float* W2, const float* b2, int kind,
                       const float* x_pos /*kind 0*/, float* out, ddmp_stream stream);
int ddmp_head_fwd_f32(const float* Y, int64_t ldy, int64_t n_rows, const float* scale, const float* shift,
                      float slope, const float* W1, const float* b1, const float* W2, const float* b2,
                      int kind, const float* x_pos /*kind 0*/, float* out, ddmp_stream stream);
int ddmp_spmm_bf16(const ddmp_graph* g, const uint16_t* X, int64_t ldx, uint16_t* Y, int64_t ldy, int C,
                   const float* bias, const float* pro_scale, const float* pro_shift, float slope, ddmp_stream stream);
int ddmp_spmm_bnbwd_bf16(const ddmp_graph* g, const uint16_t* dZ, int64_t lddz, const uint16_t* Yb, int64_t ldyb,
                         uint16_t* out, int64_t ld_out, int C, const float* a, const float* b, const float* c1,
                         const float* c0, float slope, ddmp_stream stream);
int ddmp_spmm_bnred_bf16(const ddmp_graph* g, const uint16_t* X, int64_t ldx, uint16_t* Y, int64_t ldy, int C,
                         const uint16_t* Yp, int64_t ldyp, const float* scale, const float* shift, const float* mean,
                         const float* rstd, float slope, double* sums2 /*[2C]*/, void* workspace, size_t workspace_bytes,
                         ddmp_stream stream);
size_t ddmp_spmm_bnred_bf16_workspace_bytes(int64_t n_rows, int C);
int ddmp_spmm_bnred_f32(const ddmp_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy, int C,
                        const float* Yp, int64_t ldyp, const float* scale, const float* shift, const float* mean,
                        const float* rstd, float slope, double* sums2 /*[2C]*/, void* workspace,
                        size_t workspace_bytes, ddmp_stream stream);
size_t ddmp_spmm_bnred_workspace_bytes(int64_t n_rows, int C);
int ddmp_spmm_stats_bf16(const ddmp_graph* g, const uint16_t* X, int64_t ldx, uint16_t* Y, int64_t ldy, int C,
                         const float* bias /*nullable*/, const float* pro_scale /*nullable*/,
                         const float* pro_shift /*nullable*/, float slope, const float* ref /*[C], nullable*/,
                         double* sums2 /*[2C]*/, void* workspace, size_t workspace_bytes, ddmp_stream stream);
int ddmp_bn_bwd_reduce_f32(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t n_rows, int C,
                           const float* scale, const float* shift, const float* mean, const float* rstd,
                           float slope, double* sums2 /*[2C]*/, void* workspace, size_t workspace_bytes,
                           ddmp_stream stream);
int ddmp_bn_next_prepare(double n_total, int C, const float* gamma, const float* beta, float eps, float momentum,
                         float* scale, float* shift, float* mean, float* rstd, float* running_mean /*nullable*/,
                         float* running_var /*nullable*/);
int ddmp_bn_next_bwd_prepare(double n_total, int C, const float* scale, const float* mean, const float* rstd,
                             float* dgamma, float* dbeta, float* c1, float* c0);
int ddmp_bn_next_cancel(void);
int ddmp_next_cancel(void);
int ddmp_gemm_next_scales(float* slot_a, float* slot_b, int prime);
int ddmp_gemm_next_prepared(void);
#ifdef __cplusplus
}
#endif
