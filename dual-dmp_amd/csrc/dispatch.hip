// dtype-tagged entry points (include/ddmp_hip.h, "dtype-tagged forms"): `void*` features + DDMP_F32 | DDMP_BF16.
// Pure dispatch onto the typed functions; plus the two conversion kernels of the bf16-feature mode.
#include "b16_common.h"
#include "finalize.h"

#include <algorithm>

namespace {
using namespace ddmp;

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const __bf16 b = (__bf16)in[i];
        out[i] = __builtin_bit_cast(unsigned short, b);
    }
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = __uint_as_float((unsigned)in[i] << 16);
}
inline bool dt_ok(int dtype) { return dtype == DDMP_F32 || dtype == DDMP_BF16; }
typedef const float* cf;
typedef const uint16_t* cb;
}  // namespace

// A one-thread kernel with a name of its own: bench.py launches it right before and right after its timed region, so that
// a rocprofv3 kernel trace of the bench command can be cut to exactly the timed iterations (scripts/rocpd_summary.py
// --between-markers): no priming iteration, no set-up copies in the committed statistics.
__global__ void ddmp_trace_marker_kernel(int* p) {
    if (p) *p = 1;
}
extern "C" int ddmp_trace_marker(ddmp_stream stream) {
    hipLaunchKernelGGL(ddmp_trace_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)nullptr);
    LAUNCH_TRY();
    return DDMP_OK;
}

// The yardstick of bench.py's "fraction of what a copy reaches" figures: a streaming copy written like the kernels it is
// compared with (16 bytes per lane, 4 loads in flight per lane before the first store, workgroup b on XCD b % 8 walking
// that XCD's contiguous eighth of the buffer) -- NOT torch's elementwise copy_ (4.7-5.1 TB/s on these boxes, which flattered
// every "x % of a device copy" in round 4; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).  mode 0: plain loads and
// stores; 1: nontemporal loads and stores (what the gather's output stores use).
template <int MODE>
__global__ __launch_bounds__(256) void copy_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16,
                                                         int blocks_per_xcd) {
    typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
    const int64_t per_xcd = (n16 + ddmp::kXcd - 1) / ddmp::kXcd;
    const int64_t lo = (int64_t)(blockIdx.x & (ddmp::kXcd - 1)) * per_xcd, hi = min(n16, lo + per_xcd);
    const int64_t step = (int64_t)blocks_per_xcd * 1024;
    for (int64_t i = lo + (int64_t)(blockIdx.x >> 3) * 1024 + threadIdx.x; i < hi; i += step) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = min(i + 256 * u, hi - 1);
            if (MODE == 1) {
                const nt_u4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(src + j));
                v[u] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                v[u] = src[j];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + 256 * u;
            if (j < hi) {
                if (MODE == 1) {
                    nt_u4 t = {v[u].x, v[u].y, v[u].z, v[u].w};
                    __builtin_nontemporal_store(t, reinterpret_cast<nt_u4*>(dst + j));
                } else {
                    dst[j] = v[u];
                }
            }
        }
    }
}
// The same bytes in the GATHER's access pattern: a workgroup owns 64 consecutive rows of a row-major [n_rows, row_bytes] matrix
// and walks them one 128-byte slab at a time (8 lanes x 16 bytes per row, 8 rows per wave instruction, nontemporal stores) --
// what spmm_lean / spmm_patch do minus the gather itself.  Its rate is the ceiling of that pattern on this box: the distance to
// the streaming copy above is what walking 1-2 KB rows slab by slab costs in HBM efficiency, whatever the graph is.
__global__ __launch_bounds__(256) void copy_probe_rows_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                              int n_rows, int row_bytes, int chunks_per_xcd, int n_chunks) {
    typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
    const int chunk = (blockIdx.x & (ddmp::kXcd - 1)) * chunks_per_xcd + (blockIdx.x >> 3);
    if (chunk >= n_chunks) return;
    const int r0 = chunk * 64, nr = min(64, n_rows - r0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 3, sl = lane & 7;
    for (int b0 = 0; b0 < row_bytes; b0 += 128) {
        uint4 v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int lr = min(wave * 8 + grp + 32 * q, nr - 1);
            v[q] = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + lr) * row_bytes + b0 + sl * 16);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int lr = wave * 8 + grp + 32 * q;
            if (lr < nr) {
                nt_u4 t = {v[q].x, v[q].y, v[q].z, v[q].w};
                __builtin_nontemporal_store(t, reinterpret_cast<nt_u4*>(dst + (int64_t)(r0 + lr) * row_bytes + b0 + sl * 16));
            }
        }
    }
}
extern "C" int ddmp_copy_probe_rows(const void* src, void* dst, int64_t n_rows, int row_bytes, ddmp_stream stream) {
    ARG_TRY(src && dst && n_rows > 0 && n_rows < (int64_t)INT32_MAX && row_bytes >= 128 && row_bytes % 128 == 0);
    ARG_TRY(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0);
    const int n_chunks = (int)((n_rows + 63) / 64), cpx = (n_chunks + ddmp::kXcd - 1) / ddmp::kXcd;
    hipLaunchKernelGGL(copy_probe_rows_kernel, dim3(cpx * ddmp::kXcd), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src,
                       (unsigned char*)dst, (int)n_rows, row_bytes, cpx, n_chunks);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_copy_probe(const void* src, void* dst, int64_t bytes, int mode, ddmp_stream stream) {
    ARG_TRY(src && dst && bytes > 0 && bytes % 16 == 0 && (mode == 0 || mode == 1));
    ARG_TRY(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0);
    const int bpx = 256;                                         // 2048 workgroups: 8 per CU
    const int64_t n16 = bytes / 16;
    if (mode == 1)
        hipLaunchKernelGGL(copy_probe_kernel<1>, dim3(bpx * ddmp::kXcd), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
                           (uint4*)dst, n16, bpx);
    else
        hipLaunchKernelGGL(copy_probe_kernel<0>, dim3(bpx * ddmp::kXcd), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
                           (uint4*)dst, n16, bpx);
    LAUNCH_TRY();
    return DDMP_OK;
}

// The GEMM sources carry compile-time hooks for timing-only A/B builds (scripts/build_ablation*.sh, scripts/rr_ablation.sh:
// -DDDMP_ABLATE / _PANEL_ABLATE / _RR_ABLATE / _TN_ABLATE remove parts of a kernel; RESULTS ARE WRONG in such builds).  A
// library built with any of them says so here, and the Python host side refuses to load it as the product library
// (dual-dmp_amd/_lib.py; tests/test_cli_cpu.py asserts 0 for the in-tree build).
extern "C" int ddmp_build_ablation_flags(void) {
    int f = 0;
#ifdef DDMP_ABLATE
    f |= 1;
#endif
#ifdef DDMP_PANEL_ABLATE
    f |= 2;
#endif
#if defined(DDMP_RR_ABLATE) && DDMP_RR_ABLATE != 0
    f |= 4;
#endif
#ifdef DDMP_TN_ABLATE
    f |= 8;
#endif
#ifdef DDMP_NO_NT
    f |= 16;
#endif
    return f;
}

// ---------------------------------------------------------------- tail-fused finalisation of column reductions (finalize.h)
namespace ddmp {
static thread_local FinalizeArgs g_fin_pending, g_fin_active;
FinalizeArgs& finalize_pending() { return g_fin_pending; }
FinalizeArgs& finalize_active() { return g_fin_active; }
FinalizeArgs finalize_take(int C) {
    FinalizeArgs f = g_fin_active;
    if (f.kind == 0 || f.C != C) return FinalizeArgs();
    g_fin_active = FinalizeArgs();
    return f;
}
FinalizeScope::FinalizeScope(const double* sums_, ddmp_stream stream, int width)
    : sums(sums_), st((hipStream_t)stream), owns(false) {
    if (g_fin_pending.kind != 0) {
        const bool fits = g_fin_pending.C == width && sums_ != nullptr;
        if (fits) g_fin_active = g_fin_pending;
        g_fin_pending = FinalizeArgs();                          // consumed or dropped: never left for a later call
        owns = fits;
    }
}
FinalizeScope::~FinalizeScope() {
    if (!owns) return;
    const FinalizeArgs f = g_fin_active;
    g_fin_active = FinalizeArgs();
    if (f.kind == 1)
        (void)ddmp_bn_prepare_f32(sums, f.n_total, f.C, f.in[0], f.in[1], f.eps, f.momentum, f.out[0], f.out[1], f.out[2],
                                  f.out[3], f.out[4], f.out[5], (ddmp_stream)st);
    else if (f.kind == 2)
        (void)ddmp_bn_bwd_prepare_f32(sums, f.n_total, f.C, f.in[0], f.in[1], f.in[2], f.out[0], f.out[1], f.out[2], f.out[3],
                                      (ddmp_stream)st);
}
}  // namespace ddmp

extern "C" int ddmp_bn_next_prepare(double n_total, int C, const float* gamma, const float* beta, float eps, float momentum,
                                    float* scale, float* shift, float* mean, float* rstd, float* running_mean,
                                    float* running_var) {
    ARG_TRY(n_total > 0 && C > 0 && gamma && beta && scale && shift && mean && rstd);
    ARG_TRY((running_mean == nullptr) == (running_var == nullptr));
    ddmp::FinalizeArgs f;
    f.kind = 1; f.C = C; f.n_total = n_total; f.eps = eps; f.momentum = momentum;
    f.in[0] = gamma; f.in[1] = beta;
    f.out[0] = scale; f.out[1] = shift; f.out[2] = mean; f.out[3] = rstd; f.out[4] = running_mean; f.out[5] = running_var;
    ddmp::finalize_pending() = f;
    return DDMP_OK;
}
extern "C" int ddmp_bn_next_bwd_prepare(double n_total, int C, const float* scale, const float* mean, const float* rstd,
                                        float* dgamma, float* dbeta, float* c1, float* c0) {
    ARG_TRY(n_total > 0 && C > 0 && scale && mean && rstd && dgamma && dbeta && c1 && c0);
    ddmp::FinalizeArgs f;
    f.kind = 2; f.C = C; f.n_total = n_total;
    f.in[0] = scale; f.in[1] = mean; f.in[2] = rstd;
    f.out[0] = dgamma; f.out[1] = dbeta; f.out[2] = c1; f.out[3] = c0;
    ddmp::finalize_pending() = f;
    return DDMP_OK;
}
extern "C" int ddmp_bn_next_cancel(void) {
    ddmp::finalize_pending() = ddmp::FinalizeArgs();
    return DDMP_OK;
}
// bit 0: BatchNorm coefficients armed (ddmp_bn_next_*), bit 1: GEMM scale slots named (ddmp_gemm_next_scales), bit 2: prepared
// weight planes announced (ddmp_gemm_next_prepared)
extern "C" int ddmp_next_pending(void) { return (ddmp::finalize_pending().kind != 0 ? 1 : 0) | ddmp::gemm_next_pending(); }
extern "C" int ddmp_next_cancel(void) {
    ddmp::finalize_pending() = ddmp::FinalizeArgs();
    ddmp::gemm_next_cancel();
    return DDMP_OK;
}

extern "C" int ddmp_f32_to_bf16(const float* in, uint16_t* out, int64_t n, ddmp_stream stream) {
    ARG_TRY(in && out && n > 0);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0, (hipStream_t)stream, in, out, n);
    LAUNCH_TRY();
    return DDMP_OK;
}
extern "C" int ddmp_bf16_to_f32(const uint16_t* in, float* out, int64_t n, ddmp_stream stream) {
    ARG_TRY(in && out && n > 0);
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0, (hipStream_t)stream, in, out, n);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_spmm(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype,
                         const float* bias, const float* ps, const float* psh, float slope, ddmp_stream st) {
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16 ? ddmp_spmm_bf16(g, (cb)X, ldx, (uint16_t*)Y, ldy, C, bias, ps, psh, slope, st)
                              : ddmp_spmm_f32(g, (cf)X, ldx, (float*)Y, ldy, C, bias, ps, psh, slope, st);
}
extern "C" size_t ddmp_spmm_bnred_ws_bytes(int64_t n_rows, int C, int dtype) {
    return dtype == DDMP_BF16 ? ddmp_spmm_bnred_bf16_workspace_bytes(n_rows, C) : ddmp_spmm_bnred_workspace_bytes(n_rows, C);
}
extern "C" int ddmp_spmm_bnred(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype,
                               const void* Yp, int64_t ldyp, const float* scale, const float* shift, const float* mean,
                               const float* rstd, float slope, double* sums2, void* ws, size_t wsb, ddmp_stream st) {
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16
               ? ddmp_spmm_bnred_bf16(g, (cb)X, ldx, (uint16_t*)Y, ldy, C, (cb)Yp, ldyp, scale, shift, mean, rstd, slope, sums2, ws, wsb, st)
               : ddmp_spmm_bnred_f32(g, (cf)X, ldx, (float*)Y, ldy, C, (cf)Yp, ldyp, scale, shift, mean, rstd, slope, sums2, ws, wsb, st);
}
extern "C" int ddmp_spmm_stats(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype,
                               const float* bias, const float* ps, const float* psh, float slope, const float* ref,
                               double* sums2, void* ws, size_t wsb, ddmp_stream st) {
    ddmp::FinalizeScope fin_scope(sums2, st, C);
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16 ? ddmp_spmm_stats_bf16(g, (cb)X, ldx, (uint16_t*)Y, ldy, C, bias, ps, psh, slope, ref, sums2, ws, wsb, st)
                              : ddmp_spmm_stats_f32(g, (cf)X, ldx, (float*)Y, ldy, C, bias, ps, psh, slope, ref, sums2, ws, wsb, st);
}
extern "C" int ddmp_spmm_bnbwd(const ddmp_graph* g, const void* dZ, int64_t lddz, const void* Yb, int64_t ldyb, void* out,
                               int64_t ld_out, int C, int dtype, const float* a, const float* b, const float* c1,
                               const float* c0, float slope, ddmp_stream st) {
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16 ? ddmp_spmm_bnbwd_bf16(g, (cb)dZ, lddz, (cb)Yb, ldyb, (uint16_t*)out, ld_out, C, a, b, c1, c0, slope, st)
                              : ddmp_spmm_bnbwd_f32(g, (cf)dZ, lddz, (cf)Yb, ldyb, (float*)out, ld_out, C, a, b, c1, c0, slope, st);
}
extern "C" size_t ddmp_gemm_rows_ws_bytes(int K, int M, int dtype) {
    return dtype == DDMP_BF16 ? ddmp_gemm_rows_bf16_workspace_bytes(K, M) : ddmp_gemm_rows_workspace_bytes(K, M);
}
extern "C" int ddmp_gemm_nt(const void* A, int64_t lda, const float* W, int64_t ldw, void* Y, int64_t ldy, int64_t n, int K,
                            int M, int dtype, const float* bias, const float* ps, const float* psh, float slope, void* ws,
                            size_t wsb, ddmp_stream st) {
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16 ? ddmp_gemm_nt_bf16((cb)A, lda, W, ldw, (uint16_t*)Y, ldy, n, K, M, bias, ps, psh, slope, ws, wsb, st)
                              : ddmp_gemm_nt_f32((cf)A, lda, W, ldw, (float*)Y, ldy, n, K, M, bias, ps, psh, slope, ws, wsb, st);
}
extern "C" int ddmp_gemm_nn(const void* A, int64_t lda, const float* W, int64_t ldw, void* Y, int64_t ldy, int64_t n, int M,
                            int K, int dtype, void* ws, size_t wsb, ddmp_stream st) {
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16 ? ddmp_gemm_nn_bf16((cb)A, lda, W, ldw, (uint16_t*)Y, ldy, n, M, K, ws, wsb, st)
                              : ddmp_gemm_nn_f32((cf)A, lda, W, ldw, (float*)Y, ldy, n, M, K, ws, wsb, st);
}
extern "C" size_t ddmp_gemm_tn_ws_bytes(int64_t n_rows, int M, int K, int dtype) {
    return dtype == DDMP_BF16 ? ddmp_gemm_tn_bf16_workspace_bytes(n_rows, M, K) : ddmp_gemm_tn_workspace_bytes(n_rows, M, K);
}
extern "C" int ddmp_gemm_tn(const void* G, int64_t ldg, const void* Z, int64_t ldz, float* dW, int64_t lddw, int64_t n, int M,
                            int K, int dtype, const float* ps, const float* psh, float slope, void* ws, size_t wsb,
                            ddmp_stream st) {
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16 ? ddmp_gemm_tn_bf16((cb)G, ldg, (cb)Z, ldz, dW, lddw, n, M, K, ps, psh, slope, ws, wsb, st)
                              : ddmp_gemm_tn_f32((cf)G, ldg, (cf)Z, ldz, dW, lddw, n, M, K, ps, psh, slope, ws, wsb, st);
}
extern "C" int ddmp_bn_stats(const void* Y, int64_t ldy, int64_t n, int C, int dtype, double* sums, void* ws, size_t wsb,
                             ddmp_stream st) {
    ddmp::FinalizeScope fin_scope(sums, st, C);
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16 ? ddmp_bn_stats_bf16((cb)Y, ldy, n, C, sums, ws, wsb, st) : ddmp_bn_stats_f32((cf)Y, ldy, n, C, sums, ws, wsb, st);
}
extern "C" int ddmp_bn_bwd_reduce(const void* dZ, int64_t lddz, const void* Y, int64_t ldy, int64_t n, int C, int dtype,
                                  const float* scale, const float* shift, const float* mean, const float* rstd, float slope,
                                  double* sums2, void* ws, size_t wsb, ddmp_stream st) {
    ddmp::FinalizeScope fin_scope(sums2, st, C);
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16 ? ddmp_bn_bwd_reduce_bf16((cb)dZ, lddz, (cb)Y, ldy, n, C, scale, shift, mean, rstd, slope, sums2, ws, wsb, st)
                              : ddmp_bn_bwd_reduce_f32((cf)dZ, lddz, (cf)Y, ldy, n, C, scale, shift, mean, rstd, slope, sums2, ws, wsb, st);
}
extern "C" int ddmp_bn_bwd_apply(const void* dZ, int64_t lddz, const void* Y, int64_t ldy, void* dY, int64_t lddy, int64_t n,
                                 int C, int dtype, const float* scale, const float* shift, const float* c1, const float* c0,
                                 float slope, double* dbias_sums, void* ws, size_t wsb, ddmp_stream st) {
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16
               ? ddmp_bn_bwd_apply_bf16((cb)dZ, lddz, (cb)Y, ldy, (uint16_t*)dY, lddy, n, C, scale, shift, c1, c0, slope, dbias_sums, ws, wsb, st)
               : ddmp_bn_bwd_apply_f32((cf)dZ, lddz, (cf)Y, ldy, (float*)dY, lddy, n, C, scale, shift, c1, c0, slope, dbias_sums, ws, wsb, st);
}
extern "C" int ddmp_head_fwd(const void* Y, int64_t ldy, int64_t n, int dtype, const float* scale, const float* shift,
                             float slope, const float* W1, const float* b1, const float* W2, const float* b2, int kind,
                             const float* x_pos, float* out, ddmp_stream st) {
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16 ? ddmp_head_fwd_bf16((cb)Y, ldy, n, scale, shift, slope, W1, b1, W2, b2, kind, x_pos, out, st)
                              : ddmp_head_fwd_f32((cf)Y, ldy, n, scale, shift, slope, W1, b1, W2, b2, kind, x_pos, out, st);
}
extern "C" int ddmp_head_bwd(const void* Y, int64_t ldy, int64_t n, int dtype, const float* scale, const float* shift,
                             float slope, const float* W1, const float* b1, const float* W2, const float* b2, int kind,
                             const float* dout, void* dZ, int64_t lddz, float* dW1, float* db1, float* dW2, float* db2,
                             void* ws, size_t wsb, ddmp_stream st) {
    ARG_TRY(dt_ok(dtype));
    return dtype == DDMP_BF16
               ? ddmp_head_bwd_bf16((cb)Y, ldy, n, scale, shift, slope, W1, b1, W2, b2, kind, dout, (uint16_t*)dZ, lddz, dW1, db1, dW2, db2, ws, wsb, st)
               : ddmp_head_bwd_f32((cf)Y, ldy, n, scale, shift, slope, W1, b1, W2, b2, kind, dout, (float*)dZ, lddz, dW1, db1, dW2, db2, ws, wsb, st);
}

// ---- row gather / scatter for halo packing and node relabelling (16-byte pieces; rows of C elements, C * elt % 16 == 0)
namespace {
__global__ __launch_bounds__(256) void rows_gather_kernel(const uint4* __restrict__ src, int64_t ld16, const int64_t* __restrict__ idx,
                                                          int64_t n, int q, uint4* __restrict__ dst, int64_t ldd16, int scatter) {
    const int64_t total = n * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / q;
        const int c = (int)(i - r * q);
        const int64_t j = idx[r];
        if (scatter) dst[j * ldd16 + c] = src[r * ld16 + c];
        else dst[r * ldd16 + c] = src[j * ld16 + c];
    }
}
}  // namespace

extern "C" int ddmp_rows_gather(const void* src, int64_t ld_src, const int64_t* idx, int64_t n, int C, int dtype, void* dst,
                                int64_t ld_dst, int scatter, ddmp_stream stream) {
    ARG_TRY(src && idx && dst && n >= 0 && C > 0 && dt_ok(dtype));
    const int es = dtype == DDMP_BF16 ? 2 : 4;
    ARG_TRY((C * es) % 16 == 0 && (ld_src * es) % 16 == 0 && (ld_dst * es) % 16 == 0 && ld_src >= C && ld_dst >= C);
    ARG_TRY(b16_aligned(src) && b16_aligned(dst));
    if (n == 0) return DDMP_OK;
    const int q = C * es / 16;
    hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n * q, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)src, ld_src * es / 16, idx, n, q, (uint4*)dst, ld_dst * es / 16, scatter);
    LAUNCH_TRY();
    return DDMP_OK;
}
