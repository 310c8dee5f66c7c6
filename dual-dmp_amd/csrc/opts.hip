// ABI 3: the *_o entry points -- per-call options (include/ddmp_hip.h, ddmp_opts) instead of state armed "for the next call".
// The kernels' entry points still read a per-thread record at their top (finalize.h, gemm.hip: take_scale_ctx); an _o form fills
// that record from its options right before the call it wraps and clears whatever is left of it right after -- nothing an _o
// call is given can reach another call, and nothing armed earlier can reach an _o call.
#include "ddmp_common.h"

namespace {
struct OptScope {
    int err = DDMP_OK;
    explicit OptScope(const ddmp_opts* o) {
        ddmp_next_cancel();
        if (!o) return;
        if (o->struct_size < sizeof(ddmp_opts) || (o->flags & ~(DDMP_OPT_BN_FWD | DDMP_OPT_BN_BWD | DDMP_OPT_SCALES | DDMP_OPT_PREPARED)) ||
            ((o->flags & DDMP_OPT_BN_FWD) && (o->flags & DDMP_OPT_BN_BWD))) {
            err = DDMP_EINVAL;
            return;
        }
        if (o->flags & DDMP_OPT_BN_FWD)
            err = ddmp_bn_next_prepare(o->bn_n_total, o->bn_C, o->bn_in[0], o->bn_in[1], o->bn_eps, o->bn_momentum, o->bn_out[0],
                                       o->bn_out[1], o->bn_out[2], o->bn_out[3], o->bn_out[4], o->bn_out[5]);
        if (err == DDMP_OK && (o->flags & DDMP_OPT_BN_BWD))
            err = ddmp_bn_next_bwd_prepare(o->bn_n_total, o->bn_C, o->bn_in[0], o->bn_in[1], o->bn_in[2], o->bn_out[0], o->bn_out[1],
                                           o->bn_out[2], o->bn_out[3]);
        if (err == DDMP_OK && (o->flags & DDMP_OPT_SCALES)) err = ddmp_gemm_next_scales(o->slot_a, o->slot_b, o->prime);
        if (err == DDMP_OK && (o->flags & DDMP_OPT_PREPARED)) err = ddmp_gemm_next_prepared();
        if (err != DDMP_OK) ddmp_next_cancel();
    }
    ~OptScope() { ddmp_next_cancel(); }
};
}  // namespace

extern "C" int ddmp_bn_stats_o(const void* Y, int64_t ldy, int64_t n_rows, int C, int dtype, double* sums, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_bn_stats(Y, ldy, n_rows, C, dtype, sums, workspace, workspace_bytes, stream);
}

extern "C" int ddmp_bn_bwd_reduce_o(const void* dZ, int64_t lddz, const void* Y, int64_t ldy, int64_t n_rows, int C, int dtype, const float* scale, const float* shift, const float* mean, const float* rstd, float slope, double* sums2, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_bn_bwd_reduce(dZ, lddz, Y, ldy, n_rows, C, dtype, scale, shift, mean, rstd, slope, sums2, workspace, workspace_bytes, stream);
}

extern "C" int ddmp_spmm_stats_o(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype, const float* bias, const float* pro_scale, const float* pro_shift, float slope, const float* ref, double* sums2, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_spmm_stats(g, X, ldx, Y, ldy, C, dtype, bias, pro_scale, pro_shift, slope, ref, sums2, workspace, workspace_bytes, stream);
}

extern "C" int ddmp_spmm_bnred_o(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype, const void* Yp, int64_t ldyp, const float* scale, const float* shift, const float* mean, const float* rstd, float slope, double* sums2, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_spmm_bnred(g, X, ldx, Y, ldy, C, dtype, Yp, ldyp, scale, shift, mean, rstd, slope, sums2, workspace, workspace_bytes, stream);
}

extern "C" int ddmp_gemm_nt_o(const void* A, int64_t lda, const float* W, int64_t ldw, void* Y, int64_t ldy, int64_t n_rows, int K, int M, int dtype, const float* bias, const float* pro_scale, const float* pro_shift, float slope, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_gemm_nt(A, lda, W, ldw, Y, ldy, n_rows, K, M, dtype, bias, pro_scale, pro_shift, slope, workspace, workspace_bytes, stream);
}

extern "C" int ddmp_gemm_nn_o(const void* A, int64_t lda, const float* W, int64_t ldw, void* Y, int64_t ldy, int64_t n_rows, int M, int K, int dtype, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_gemm_nn(A, lda, W, ldw, Y, ldy, n_rows, M, K, dtype, workspace, workspace_bytes, stream);
}

extern "C" int ddmp_gemm_tn_o(const void* G, int64_t ldg, const void* Z, int64_t ldz, float* dW, int64_t lddw, int64_t n_rows, int M, int K, int dtype, const float* pro_scale, const float* pro_shift, float slope, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_gemm_tn(G, ldg, Z, ldz, dW, lddw, n_rows, M, K, dtype, pro_scale, pro_shift, slope, workspace, workspace_bytes, stream);
}

extern "C" int ddmp_gemm_nt_stats_f32_o(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy, int64_t n_rows, int K, int M, const float* bias , const float* pro_scale , const float* pro_shift , float slope, double* sums2 , void* workspace, size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_gemm_nt_stats_f32(A, lda, W, ldw, Y, ldy, n_rows, K, M, bias, pro_scale, pro_shift, slope, sums2, workspace, workspace_bytes, stats_ws, stats_ws_bytes, stream);
}

extern "C" int ddmp_gemm_nt_stats_bf16_o(const uint16_t* A, int64_t lda, const float* W, int64_t ldw, uint16_t* Y, int64_t ldy, int64_t n_rows, int K, int M, const float* bias, const float* pro_scale, const float* pro_shift, float slope, double* sums2, void* workspace, size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_gemm_nt_stats_bf16(A, lda, W, ldw, Y, ldy, n_rows, K, M, bias, pro_scale, pro_shift, slope, sums2, workspace, workspace_bytes, stats_ws, stats_ws_bytes, stream);
}

extern "C" int ddmp_gemm_nn_bnred_f32_o(const float* A, int64_t lda, const float* W, int64_t ldw, float* out, int64_t ld_out, int64_t n_rows, int M, int K, const float* Yp, int64_t ldyp, const float* scale, const float* shift, const float* mean, const float* rstd, float slope, double* sums2, void* workspace, size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_gemm_nn_bnred_f32(A, lda, W, ldw, out, ld_out, n_rows, M, K, Yp, ldyp, scale, shift, mean, rstd, slope, sums2, workspace, workspace_bytes, stats_ws, stats_ws_bytes, stream);
}


extern "C" int ddmp_gemm_nn_bnbwd_f32_o(const float* dZ, int64_t lddz, const float* Yb, int64_t ldyb, const float* W, int64_t ldw, float* out, int64_t ld_out, int64_t n_rows, int M, int K, const float* a, const float* b, const float* c1, const float* c0, float slope, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_gemm_nn_bnbwd_f32(dZ, lddz, Yb, ldyb, W, ldw, out, ld_out, n_rows, M, K, a, b, c1, c0, slope, workspace, workspace_bytes, stream);
}

extern "C" int ddmp_gemm_tn_bnbwd_f32_o(const float* dZ, int64_t lddz, const float* Yb, int64_t ldyb, const float* Z, int64_t ldz, float* dW, int64_t lddw, int64_t n_rows, int M, int K, const float* a, const float* b, const float* c1, const float* c0, const float* pro_scale , const float* pro_shift , float slope, void* workspace, size_t workspace_bytes, ddmp_stream stream, const ddmp_opts* opts) {
    OptScope scope(opts);
    if (scope.err) return scope.err;
    return ddmp_gemm_tn_bnbwd_f32(dZ, lddz, Yb, ldyb, Z, ldz, dW, lddw, n_rows, M, K, a, b, c1, c0, pro_scale, pro_shift, slope, workspace, workspace_bytes, stream);
}
