// Launcher of the wide f16x3 weight-gradient kernel (gemm_tn_rm.hip) -- called by the tn entry points of gemm.hip.
#pragma once
#include "ddmp_common.h"

namespace ddmp {

struct TnRmArgs {
    const float* G;                  // [n_rows, M]  gradient operand (dZ when g2 != nullptr)
    int64_t ldg;
    const float* G2;                 // nullable: Yb -- the operand is then the BatchNorm+LeakyReLU backward of (G, G2)
    int64_t ldg2;
    const float* Z;                  // [n_rows, K]
    int64_t ldz;
    float* part;                     // [n_splits][M][K] partial sums
    int64_t ld_out, split_stride;
    int n_rows, M, K, rows_per_split, n_tiles_m, n_tiles_k, n_splits;
    const float* pscale;             // nullable: Z is lrelu(Z * pscale + pshift)
    const float* pshift;
    const float *ga, *gb, *gk1, *gk0;        // per column of M (with G2)
    float slope;
    float* gslot;                    // scale slots of the two operands (gemm_f16s.inc)
    float* zslot;
    int target, heal;
    int tm, tk;                      // panel of dW per workgroup: 256 x 256 | 256 x 128 | 128 x 256 (n_tiles_* count these)
};

// grid = cdiv(n_splits, 8) * 8 * n_tiles_m * n_tiles_k workgroups of 512 threads (TnPlan of gemm.hip, T >= 4)
void launch_tn_rm(const TnRmArgs& a, hipStream_t st);

}  // namespace ddmp
