// Train-mode BatchNorm1d + LeakyReLU over [N, C] node features (util/networks.py:31-44,51-62 of the
// reference: bn_k(conv_k(.)) then LeakyReLU(0.01), batch statistics over ALL nodes, biased variance,
// eps 1e-5, running stats with momentum 0.1).  HBM-bound streaming passes:
//
//   forward   stats : one read of Y  -> per-column (sum, sumsq) in float64 (per-thread f64 accumulators,
//                     per-block partials, no atomics -> deterministic)            bytes = N*C*4
//             the normalise+activate itself is NOT a pass: consumers (next GEMM / SpMM / head) apply
//             z = LeakyReLU(scale[c]*y + shift[c]) as a load prologue; ddmp_bn_lrelu_apply_f32 exists for
//             callers that want Z materialised.
//   backward  reduce: one read of dZ and Y -> (sum g, sum g*yhat),  g = dZ * LeakyReLU'(scale*y+shift)
//             apply : dY = scale*g + c1*y + c0  (c1, c0 fold the two batch means), optional column sums
//                     of dY (= gradient of the conv bias, analytically 0 after BN).
//
// Column layout: C/4 lanes (float4) per row, 256/(C/4) rows per workgroup step, grid-stride over rows.
#include "b16_common.h"
#include "finalize.h"

#include <algorithm>

namespace {

using namespace ddmp;

constexpr int kMaxBlocks = 1024;

// element access of the two feature dtypes: VW consecutive columns per lane (16 bytes)
template <typename T> struct El;
template <> struct El<float> {
    static constexpr int VW = 4;
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void st(float* p, float (&v)[4]) { nt_store4(p, make_float4(v[0], v[1], v[2], v[3])); }
};
template <> struct El<bf16_t> {
    static constexpr int VW = 8;
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) { bf_unpack8(ld8b(p), v); }
    static __device__ __forceinline__ void st(bf16_t* p, float (&v)[8]) {       // v <- the values as stored (rounded)
        const uint4 o = bf_pack8(v);
        nt_st8b(p, o);
        bf_unpack8(o, v);
    }
};
template <int VW> __device__ __forceinline__ void ldc(const float* p, float (&v)[VW]) {
#pragma unroll
    for (int q = 0; q < VW / 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}

__host__ __device__ inline bool width_ok(int C) { return C >= 8 && C <= 1024 && (C & (C - 1)) == 0; }

int colreduce_blocks(int64_t n_rows, int C, int VW = 4) {
    // >= 32 row steps per workgroup where the input allows (every workgroup leaves a [2C] float64 record for the second
    // stage: with 4 steps each a 13k-row input wrote 8 MB of records for 27 MB of data), but never fewer workgroups than CUs
    const int rpi = 256 / (C / VW);
    const int64_t few = cdiv(n_rows, (int64_t)rpi * 4), many = cdiv(n_rows, (int64_t)rpi * 32);
    return (int)std::max<int64_t>(1, std::min<int64_t>(kMaxBlocks, std::max<int64_t>(many, std::min<int64_t>(kCu, few))));
}

// F::operator()(row, c0, s0[VW], s1[VW]) accumulates two per-column quantities for columns c0..c0+VW-1
// (VW = F::VW = 4 float32 | 8 bfloat16 columns per lane; C >= 2 VW so that a row takes at most 128 lanes)
template <class F>
__global__ __launch_bounds__(256) void colreduce_kernel(F f, int n_rows, int C, double* __restrict__ partial) {
    constexpr int VW = F::VW;
    __shared__ double sm[2 * 1024 * (VW / 4)];
    constexpr int kHalf = 1024 * (VW / 4);
    const int lpr = C / VW, rpi = 256 / lpr;
    const int tid = threadIdx.x, sl = tid % lpr, rg = tid / lpr;
    double s0[VW], s1[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) s0[k] = s1[k] = 0.0;
    for (int row = blockIdx.x * rpi + rg; row < n_rows; row += gridDim.x * rpi) f(row, sl * VW, s0, s1);
    // sm[v][rg][c]
#pragma unroll
    for (int k = 0; k < VW; ++k) {
        sm[rg * C + sl * VW + k] = s0[k];
        sm[kHalf + rg * C + sl * VW + k] = s1[k];
    }
    __syncthreads();
    for (int o = tid; o < 2 * C; o += 256) {
        const int v = o / C, c = o % C;
        double t = 0.0;
        for (int r = 0; r < rpi; ++r) t += sm[v * kHalf + r * C + c];
        partial[(int64_t)blockIdx.x * 2 * C + o] = t;
    }
}

template <typename T> struct StatsF {
    static constexpr int VW = El<T>::VW;
    const T* Y;
    int64_t ldy;
    __device__ __forceinline__ void operator()(int row, int c0, double* s0, double* s1) const {
        float v[VW];
        El<T>::ld(Y + (int64_t)row * ldy + c0, v);
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            const double a = v[k];
            s0[k] += a;
            s1[k] = fma(a, a, s1[k]);
        }
    }
};

template <typename T> struct BwdReduceF {
    static constexpr int VW = El<T>::VW;
    const T *dZ, *Y;
    const float *scale, *shift, *mean, *rstd;
    int64_t lddz, ldy;
    float slope;
    __device__ __forceinline__ void operator()(int row, int c0, double* s0, double* s1) const {
        float dz[VW], y[VW], a[VW], b[VW], mu[VW], rs[VW];
        El<T>::ld(dZ + (int64_t)row * lddz + c0, dz);
        El<T>::ld(Y + (int64_t)row * ldy + c0, y);
        ldc<VW>(scale + c0, a);
        ldc<VW>(shift + c0, b);
        ldc<VW>(mean + c0, mu);
        ldc<VW>(rstd + c0, rs);
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            const float g = dz[k] * lrelu_grad(fmaf(y[k], a[k], b[k]), slope);
            s0[k] += g;
            s1[k] = fma((double)g, (double)((y[k] - mu[k]) * rs[k]), s1[k]);
        }
    }
};

// column sums of dY produced on the fly (bias gradient); second slot unused
template <typename T> struct BwdApplyF {
    static constexpr int VW = El<T>::VW;
    const T *dZ, *Y;
    const float *scale, *shift, *c1, *c0;
    T* dY;
    int64_t lddz, ldy, lddy;
    float slope;
    __device__ __forceinline__ void operator()(int row, int cc, double* s0, double* s1) const {
        float dz[VW], y[VW], a[VW], b[VW], k1[VW], k0[VW], o[VW];
        El<T>::ld(dZ + (int64_t)row * lddz + cc, dz);
        El<T>::ld(Y + (int64_t)row * ldy + cc, y);
        ldc<VW>(scale + cc, a);
        ldc<VW>(shift + cc, b);
        ldc<VW>(c1 + cc, k1);
        ldc<VW>(c0 + cc, k0);
#pragma unroll
        for (int k = 0; k < VW; ++k)
            o[k] = fmaf(a[k], dz[k] * lrelu_grad(fmaf(y[k], a[k], b[k]), slope), fmaf(k1[k], y[k], k0[k]));
        El<T>::st(dY + (int64_t)row * lddy + cc, o);
#pragma unroll
        for (int k = 0; k < VW; ++k) s0[k] += o[k];
        (void)s1;
    }
};

struct ColsumF {
    static constexpr int VW = 4;
    const float* X;
    int64_t ldx;
    __device__ __forceinline__ void operator()(int row, int c0, double* s0, double* s1) const {
        const float4 v = *reinterpret_cast<const float4*>(X + (int64_t)row * ldx + c0);
        s0[0] += v.x; s0[1] += v.y; s0[2] += v.z; s0[3] += v.w;
        (void)s1;
    }
};

// partial[nblk][2C] -> out[2C] (double) (+ the coefficients `fin` asks for: finalize.h).  32 columns x 8 row-slices per
// workgroup, BOTH sums of a column in one thread; every slice keeps 8 independent loads per sum in flight (the serial form
// of this loop was latency-bound: 73 us per call).  Order of additions per output: slice-strided, then the 8 slices.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ partial, int nblk, int C,
                                                              double* __restrict__ out, FinalizeArgs fin) {
    __shared__ double sm[2][256];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
    const int width = 2 * C;
    double t0 = 0.0, t1 = 0.0;
    if (c < C) {
        int b = sl;
        for (; b + 56 < nblk; b += 64) {
            double v[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = partial[(int64_t)(b + 8 * u) * width + c];
                w[u] = partial[(int64_t)(b + 8 * u) * width + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                t0 += v[u];
                t1 += w[u];
            }
        }
        for (; b < nblk; b += 8) {
            t0 += partial[(int64_t)b * width + c];
            t1 += partial[(int64_t)b * width + C + c];
        }
    }
    sm[0][threadIdx.x] = t0;
    sm[1][threadIdx.x] = t1;
    __syncthreads();
    if (sl == 0 && c < C) {
        double r0 = 0.0, r1 = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            r0 += sm[0][threadIdx.x + 32 * u];
            r1 += sm[1][threadIdx.x + 32 * u];
        }
        out[c] = r0;
        out[C + c] = r1;
        finalize_column(fin, c, r0, r1);
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, FinalizeArgs fin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < fin.C) finalize_column(fin, c, sums[c], sums[fin.C + c]);
}

__global__ void f64_to_f32_kernel(const double* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

__global__ __launch_bounds__(256) void bn_lrelu_apply_kernel(const float* __restrict__ Y, int64_t ldy,
                                                             float* __restrict__ Z, int64_t ldz, int64_t n_rows,
                                                             int C, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float slope) {
    const int q = C >> 2;
    const int64_t total = n_rows * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / q;
        const int c0 = (int)(i % q) * 4;
        const float4 v = *reinterpret_cast<const float4*>(Y + row * ldy + c0);
        const float4 a = *reinterpret_cast<const float4*>(scale + c0);
        const float4 b = *reinterpret_cast<const float4*>(shift + c0);
        *reinterpret_cast<float4*>(Z + row * ldz + c0) = f4_affine_lrelu(v, a, b, slope);
    }
}

__global__ __launch_bounds__(256) void bn_lrelu_apply_b16_kernel(const bf16_t* __restrict__ Y, int64_t ldy,
                                                                 bf16_t* __restrict__ Z, int64_t ldz, int64_t n_rows, int C,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, float slope) {
    const int q = C >> 3;
    const int64_t total = n_rows * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / q;
        const int c0 = (int)(i % q) * 8;
        float v[8], a[8], b[8];
        bf_unpack8(ld8b(Y + row * ldy + c0), v);
        ld8f(scale + c0, a);
        ld8f(shift + c0, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = lrelu(fmaf(v[k], a[k], b[k]), slope);
        st8b(Z + row * ldz + c0, bf_pack8(v));
    }
}

template <class F>
int run_colreduce(const F& f, int64_t n_rows, int C, int width, double* out, void* ws, size_t ws_bytes,
                  hipStream_t st) {
    const int nblk = colreduce_blocks(n_rows, C, F::VW);
    if (!ws || ws_bytes < (size_t)nblk * 2 * C * sizeof(double)) return DDMP_EWORKSPACE;
    double* partial = (double*)ws;
    hipLaunchKernelGGL((colreduce_kernel<F>), dim3(nblk), dim3(256), 0, st, f, (int)n_rows, C, partial);
    LAUNCH_TRY();
    // partial rows are [2*C]; reduce the first `width` columns (width = C or 2C)
    // (width = 2C: the coefficients armed by ddmp_bn_next_* ride on the second stage; C: only the first half is exported)
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv(C, 32)), dim3(256), 0, st, partial, nblk, C, out,
                       width == 2 * C ? finalize_take(C) : FinalizeArgs());
    LAUNCH_TRY();
    return DDMP_OK;
}

}  // namespace

extern "C" size_t ddmp_colreduce_workspace_bytes(int64_t n_rows, int C) {
    if (n_rows <= 0 || !width_ok(C)) return 0;
    // partials + room for one [2C] double result used by calls that only export part of it
    return ((size_t)colreduce_blocks(n_rows, C) * 2 * C + 2 * (size_t)C) * sizeof(double);
}

extern "C" int ddmp_bn_stats_f32(const float* Y, int64_t ldy, int64_t n_rows, int C, double* sums,
                                 void* ws, size_t ws_bytes, ddmp_stream stream) {
    FinalizeScope fin_scope(sums, stream, C);
    ARG_TRY(Y && sums && n_rows > 0 && n_rows < INT32_MAX && width_ok(C) && ldy >= C && ldy % 4 == 0);
    StatsF<float> f{Y, ldy};
    return run_colreduce(f, n_rows, C, 2 * C, sums, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int ddmp_bn_prepare_f32(const double* sums, double n_total, int C, const float* gamma,
                                   const float* beta, float eps, float momentum, float* scale,
                                   float* shift, float* mean, float* rstd, float* running_mean,
                                   float* running_var, ddmp_stream stream) {
    ARG_TRY(sums && n_total > 0 && C > 0 && gamma && beta && scale && shift && mean && rstd);
    ARG_TRY((running_mean == nullptr) == (running_var == nullptr));
    FinalizeArgs f;
    f.kind = 1; f.C = C; f.n_total = n_total; f.eps = eps; f.momentum = momentum;
    f.in[0] = gamma; f.in[1] = beta;
    f.out[0] = scale; f.out[1] = shift; f.out[2] = mean; f.out[3] = rstd; f.out[4] = running_mean; f.out[5] = running_var;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, sums, f);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_bn_lrelu_apply_f32(const float* Y, int64_t ldy, float* Z, int64_t ldz, int64_t n_rows,
                                       int C, const float* scale, const float* shift, float slope,
                                       ddmp_stream stream) {
    ARG_TRY(Y && Z && scale && shift && n_rows > 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0 && ldz % 4 == 0);
    ARG_TRY(ldy >= C && ldz >= C);
    const int64_t total = n_rows * (C / 4);
    const int grid = (int)std::min<int64_t>(cdiv(total, 256), 256 * 8);
    hipLaunchKernelGGL(bn_lrelu_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, Y, ldy, Z, ldz,
                       n_rows, C, scale, shift, slope);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_bn_bwd_reduce_f32(const float* dZ, int64_t lddz, const float* Y, int64_t ldy,
                                      int64_t n_rows, int C, const float* scale, const float* shift,
                                      const float* mean, const float* rstd, float slope, double* sums2,
                                      void* ws, size_t ws_bytes, ddmp_stream stream) {
    FinalizeScope fin_scope(sums2, stream, C);
    ARG_TRY(dZ && Y && scale && shift && mean && rstd && sums2);
    ARG_TRY(n_rows > 0 && n_rows < INT32_MAX && width_ok(C) && ldy >= C && lddz >= C && ldy % 4 == 0 && lddz % 4 == 0);
    BwdReduceF<float> f{dZ, Y, scale, shift, mean, rstd, lddz, ldy, slope};
    return run_colreduce(f, n_rows, C, 2 * C, sums2, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int ddmp_bn_bwd_prepare_f32(const double* sums2, double n_total, int C, const float* scale,
                                       const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                       float* c1, float* c0, ddmp_stream stream) {
    ARG_TRY(sums2 && n_total > 0 && C > 0 && scale && mean && rstd && dgamma && dbeta && c1 && c0);
    FinalizeArgs f;
    f.kind = 2; f.C = C; f.n_total = n_total;
    f.in[0] = scale; f.in[1] = mean; f.in[2] = rstd;
    f.out[0] = dgamma; f.out[1] = dbeta; f.out[2] = c1; f.out[3] = c0;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, sums2, f);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_bn_bwd_apply_f32(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, float* dY,
                                     int64_t lddy, int64_t n_rows, int C, const float* scale,
                                     const float* shift, const float* c1, const float* c0, float slope,
                                     double* dbias_sums, void* ws, size_t ws_bytes, ddmp_stream stream) {
    ARG_TRY(dZ && Y && dY && scale && shift && c1 && c0);
    ARG_TRY(n_rows > 0 && n_rows < INT32_MAX && width_ok(C));
    ARG_TRY(ldy >= C && lddz >= C && lddy >= C && ldy % 4 == 0 && lddz % 4 == 0 && lddy % 4 == 0);
    BwdApplyF<float> f{dZ, Y, scale, shift, c1, c0, dY, lddz, ldy, lddy, slope};
    // result buffer [2C] sits behind the partials in the workspace; only the first C are the column sums
    const int nblk = colreduce_blocks(n_rows, C);
    const size_t need = ((size_t)nblk * 2 * C + 2 * (size_t)C) * sizeof(double);
    if (!ws || ws_bytes < need) return DDMP_EWORKSPACE;
    if (!dbias_sums) {                                           // dY only (its column sums are zero after BatchNorm)
        hipLaunchKernelGGL((colreduce_kernel<BwdApplyF<float>>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, f, (int)n_rows,
                           C, (double*)ws);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    double* tmp = (double*)ws + (size_t)nblk * 2 * C;
    int st = run_colreduce(f, n_rows, C, C, tmp, ws, (size_t)nblk * 2 * C * sizeof(double), (hipStream_t)stream);
    if (st != DDMP_OK) return st;
    HIP_TRY(hipMemcpyAsync(dbias_sums, tmp, sizeof(double) * (size_t)C, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DDMP_OK;
}

extern "C" int ddmp_colsum_f32(const float* X, int64_t ldx, int64_t n_rows, int C, double* sums /*[C]*/,
                               void* ws, size_t ws_bytes, ddmp_stream stream) {
    ARG_TRY(X && sums && n_rows > 0 && n_rows < INT32_MAX && width_ok(C) && ldx >= C && ldx % 4 == 0);
    ColsumF f{X, ldx};
    const int nblk = colreduce_blocks(n_rows, C);
    const size_t need = ((size_t)nblk * 2 * C + 2 * (size_t)C) * sizeof(double);
    if (!ws || ws_bytes < need) return DDMP_EWORKSPACE;
    double* tmp = (double*)ws + (size_t)nblk * 2 * C;
    int st = run_colreduce(f, n_rows, C, C, tmp, ws, (size_t)nblk * 2 * C * sizeof(double), (hipStream_t)stream);
    if (st != DDMP_OK) return st;
    HIP_TRY(hipMemcpyAsync(sums, tmp, sizeof(double) * (size_t)C, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DDMP_OK;
}

extern "C" int ddmp_f64_to_f32(const double* in, float* out, int64_t n, ddmp_stream stream) {
    ARG_TRY(in && out && n > 0 && n < INT32_MAX);
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, (int)n);
    LAUNCH_TRY();
    return DDMP_OK;
}


// ---------------------------------------------------------------- bfloat16 features (b16_common.h): same passes, 8 columns
// per lane, float32 arithmetic, float64 sums; C a power of two in [16, 1024]; workspace = ddmp_colreduce_workspace_bytes
extern "C" int ddmp_bn_stats_bf16(const uint16_t* Y, int64_t ldy, int64_t n_rows, int C, double* sums, void* ws,
                                  size_t ws_bytes, ddmp_stream stream) {
    FinalizeScope fin_scope(sums, stream, C);
    ARG_TRY(Y && sums && n_rows > 0 && n_rows < INT32_MAX && width_ok(C) && C >= 16 && ldy >= C && ldy % 8 == 0 && b16_aligned(Y));
    StatsF<bf16_t> f{Y, ldy};
    return run_colreduce(f, n_rows, C, 2 * C, sums, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int ddmp_bn_bwd_reduce_bf16(const uint16_t* dZ, int64_t lddz, const uint16_t* Y, int64_t ldy, int64_t n_rows,
                                       int C, const float* scale, const float* shift, const float* mean,
                                       const float* rstd, float slope, double* sums2, void* ws, size_t ws_bytes,
                                       ddmp_stream stream) {
    FinalizeScope fin_scope(sums2, stream, C);
    ARG_TRY(dZ && Y && scale && shift && mean && rstd && sums2);
    ARG_TRY(n_rows > 0 && n_rows < INT32_MAX && width_ok(C) && C >= 16 && ldy >= C && lddz >= C && ldy % 8 == 0 && lddz % 8 == 0);
    ARG_TRY(b16_aligned(dZ) && b16_aligned(Y));
    BwdReduceF<bf16_t> f{dZ, Y, scale, shift, mean, rstd, lddz, ldy, slope};
    return run_colreduce(f, n_rows, C, 2 * C, sums2, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int ddmp_bn_bwd_apply_bf16(const uint16_t* dZ, int64_t lddz, const uint16_t* Y, int64_t ldy, uint16_t* dY,
                                      int64_t lddy, int64_t n_rows, int C, const float* scale, const float* shift,
                                      const float* c1, const float* c0, float slope, double* dbias_sums, void* ws,
                                      size_t ws_bytes, ddmp_stream stream) {
    ARG_TRY(dZ && Y && dY && scale && shift && c1 && c0);
    ARG_TRY(n_rows > 0 && n_rows < INT32_MAX && width_ok(C) && C >= 16);
    ARG_TRY(ldy >= C && lddz >= C && lddy >= C && ldy % 8 == 0 && lddz % 8 == 0 && lddy % 8 == 0);
    ARG_TRY(b16_aligned(dZ) && b16_aligned(Y) && b16_aligned(dY));
    BwdApplyF<bf16_t> f{dZ, Y, scale, shift, c1, c0, dY, lddz, ldy, lddy, slope};
    const int nblk = colreduce_blocks(n_rows, C, 8);
    const size_t need = ((size_t)nblk * 2 * C + 2 * (size_t)C) * sizeof(double);
    if (!ws || ws_bytes < need) return DDMP_EWORKSPACE;
    if (!dbias_sums) {
        hipLaunchKernelGGL((colreduce_kernel<BwdApplyF<bf16_t>>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, f, (int)n_rows,
                           C, (double*)ws);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    double* tmp = (double*)ws + (size_t)nblk * 2 * C;
    int st = run_colreduce(f, n_rows, C, C, tmp, ws, (size_t)nblk * 2 * C * sizeof(double), (hipStream_t)stream);
    if (st != DDMP_OK) return st;
    HIP_TRY(hipMemcpyAsync(dbias_sums, tmp, sizeof(double) * (size_t)C, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DDMP_OK;
}

extern "C" int ddmp_bn_lrelu_apply_bf16(const uint16_t* Y, int64_t ldy, uint16_t* Z, int64_t ldz, int64_t n_rows, int C,
                                        const float* scale, const float* shift, float slope, ddmp_stream stream) {
    ARG_TRY(Y && Z && scale && shift && n_rows > 0 && C > 0 && C % 8 == 0 && ldy % 8 == 0 && ldz % 8 == 0 && ldy >= C && ldz >= C);
    ARG_TRY(b16_aligned(Y) && b16_aligned(Z));
    const int64_t total = n_rows * (C / 8);
    const int grid = (int)std::min<int64_t>(cdiv(total, 256), 256 * 8);
    hipLaunchKernelGGL(bn_lrelu_apply_b16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, Y, ldy, Z, ldz, n_rows, C,
                       scale, shift, slope);
    LAUNCH_TRY();
    return DDMP_OK;
}
