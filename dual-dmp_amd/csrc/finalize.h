// Tail-fused finalisation of the column reductions (round 3).
//
// Every BatchNorm of the path turns a float64 [2C] result of a column reduction into float32 per-column coefficients:
// forward (sum y, sum y^2) -> scale / shift / mean / rstd (+ running statistics); backward (sum g, sum g*yhat) -> dgamma,
// dbeta, c1, c0.  That used to be a kernel of its own behind the reduction's second stage -- 48 launches of ~3 us per
// iteration, 11 % of the launches of a 13k-face mesh.  ddmp_bn_next_prepare / ddmp_bn_next_bwd_prepare (include/ddmp_hip.h)
// arm the NEXT reducing entry point of this host thread: its second-stage kernel (one thread holds both sums of a column)
// writes the coefficients too, with the arithmetic of bn_prepare_kernel / bn_bwd_prepare_kernel (same device function:
// bitwise the same values).  A reducing entry that takes a route without a second stage falls back to the stand-alone
// kernel by itself (FinalizeScope), so "armed => the coefficients exist when the call returns" holds on every route.
#pragma once
#include "ddmp_common.h"

namespace ddmp {

struct FinalizeArgs {
    int kind = 0;                  // 0 nothing | 1 BatchNorm forward coefficients | 2 BatchNorm backward coefficients
    int C = 0;
    double n_total = 0.0;
    const float* in[3] = {nullptr, nullptr, nullptr};   // 1: gamma, beta, -          2: scale, mean, rstd
    float* out[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    //                                1: scale, shift, mean, rstd, running_mean, running_var      2: dgamma, dbeta, c1, c0
    float eps = 0.f, momentum = 0.f;
};

// column c: s0 = sums[c], s1 = sums[C + c]
__device__ __forceinline__ void finalize_column(const FinalizeArgs& f, int c, double s0, double s1) {
    if (f.kind == 1) {
        const double mu = s0 / f.n_total;
        double var = s1 / f.n_total - mu * mu;                   // biased
        if (var < 0.0) var = 0.0;
        const float muf = (float)mu;
        const float rs = (float)(1.0 / sqrt(var + (double)f.eps));
        const float a = f.in[0][c] * rs;
        f.out[0][c] = a;
        f.out[1][c] = fmaf(-muf, a, f.in[1][c]);
        f.out[2][c] = muf;
        f.out[3][c] = rs;
        if (f.out[4]) {
            const double unb = f.n_total > 1.0 ? var * f.n_total / (f.n_total - 1.0) : var;
            f.out[4][c] = (1.f - f.momentum) * f.out[4][c] + f.momentum * muf;
            f.out[5][c] = (1.f - f.momentum) * f.out[5][c] + f.momentum * (float)unb;
        }
    } else if (f.kind == 2) {
        const double db = s0, dg = s1;
        f.out[1][c] = (float)db;
        f.out[0][c] = (float)dg;
        const double a = f.in[0][c], r = f.in[2][c], mu = f.in[1][c];
        const double k1 = -a * r * dg / f.n_total;
        f.out[2][c] = (float)k1;
        f.out[3][c] = (float)(-a * db / f.n_total - k1 * mu);
    }
}

// host side (dispatch.hip): what ddmp_bn_next_* armed on this thread / what the running reducing entry has to apply
FinalizeArgs& finalize_pending();
FinalizeArgs& finalize_active();
// second-stage helpers: the coefficients to write for a reduction over C columns (kind 0 if none / another width), marking
// them as taken
FinalizeArgs finalize_take(int C);

// At the top of every extern "C" entry that produces a float64 [2C] column reduction.  Nested entries (a fused form falling
// back to GEMM + ddmp_bn_stats) share the outermost scope's request.
// `width` = the columns of THIS call's reduction: a pending request for another width (armed for a call that was never made:
// an error or an exception in the caller in between) is DROPPED here, never run against sums of another size; so is a
// request whose call has no sums buffer (argument error).
struct FinalizeScope {
    const double* sums;
    hipStream_t st;
    bool owns;
    FinalizeScope(const double* sums_, ddmp_stream stream, int width);
    ~FinalizeScope();                                            // request still open: stand-alone prepare kernel
};
// everything armed on this host thread for "the next call" (ddmp_next_pending / ddmp_next_cancel of the C ABI); the GEMM
// side's share lives in gemm.hip
int gemm_next_pending();
void gemm_next_cancel();

}  // namespace ddmp
