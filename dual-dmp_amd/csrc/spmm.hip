// GCN aggregation  Y[i,:] = dinv[i] * sum_{e in row i} dinv[col e] * f(X[col e,:])  (+ bias)
//
// HBM-bound gather: algorithmic bytes per call = 2*N*C*4 + 4*nnz + 4*(N+1) + 4*N.
// Layout of the work on gfx950:
//   * a workgroup (4 waves) owns a chunk of RB consecutive rows; its slice of the CSR column list and
//     the matching dinv[col] weights are staged through LDS once (coalesced), so the inner loop reads
//     neighbour ids by LDS broadcast and only feature rows come from L2/HBM;
//   * LANES = C/4 lanes (float4 each) cover one feature row, 64/LANES rows per wavefront step;
//     every gathered row segment is a full, 16-byte-aligned line fragment (coalesced 64..1024 B);
//   * up to 4 neighbour rows are in flight per lane before the first FMA (latency hiding);
//   * blockIdx -> chunk is XCD-aware: XCD x (blocks b with b % 8 == x) walks the contiguous row range
//     [x*N/8, (x+1)*N/8), so a row's ~deg re-reads by neighbouring rows hit that XCD's private L2;
//   * no atomics, fixed summation order -> deterministic.
#include "ddmp_common.h"

namespace {

using namespace ddmp;

constexpr int kRB = 64;            // rows per workgroup chunk
constexpr int kMaxE = kRB * 16;    // staged CSR entries per chunk (mesh graphs: ~7 resp. 4 per row)

template <int LANES, int CHUNKS, bool PRO>
__global__ __launch_bounds__(256) void spmm_vec_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ dinv,
    const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, int n_rows,
    const float* __restrict__ bias, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float slope, int chunks_per_xcd, int n_chunks) {
    constexpr int RPW = 64 / LANES;         // rows per wave step
    constexpr int RPB = 4 * RPW;            // rows per block step
    __shared__ int s_rowptr[kRB + 1];
    __shared__ int s_col[kMaxE];
    __shared__ float s_w[kMaxE];

    const int chunk = (blockIdx.x & (kXcd - 1)) * chunks_per_xcd + (blockIdx.x >> 3);
    if (chunk >= n_chunks) return;
    const int r0 = chunk * kRB;
    const int nr = min(kRB, n_rows - r0);
    const int tid = threadIdx.x;

    for (int i = tid; i <= nr; i += 256) s_rowptr[i] = rowptr[r0 + i];
    __syncthreads();
    const int e0 = s_rowptr[0];
    const int ne = s_rowptr[nr] - e0;
    const bool staged = ne <= kMaxE;
    if (staged) {
        for (int t = tid; t < ne; t += 256) {
            const int c = col[e0 + t];
            s_col[t] = c;
            s_w[t] = dinv[c];
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int grp = lane / LANES, sl = lane % LANES;

    float4 pa[CHUNKS], pb[CHUNKS], bs[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int off = c * LANES * 4 + sl * 4;
        if (PRO) {
            pa[c] = *reinterpret_cast<const float4*>(pscale + off);
            pb[c] = *reinterpret_cast<const float4*>(pshift + off);
        }
        bs[c] = bias ? *reinterpret_cast<const float4*>(bias + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    for (int lr = wave * RPW + grp; lr < nr; lr += RPB) {
        const int row = r0 + lr;
        const int es = s_rowptr[lr] - e0, ee = s_rowptr[lr + 1] - e0;
        float4 acc[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);

        for (int e = es; e < ee; e += 4) {
            int cj[4];
            float wj[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ek = min(e + k, ee - 1);
                if (staged) {
                    cj[k] = s_col[ek];
                    wj[k] = s_w[ek];
                } else {
                    cj[k] = col[e0 + ek];
                    wj[k] = dinv[cj[k]];
                }
                if (e + k >= ee) wj[k] = 0.f;
            }
            float4 v[4][CHUNKS];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* xr = X + (int64_t)cj[k] * ldx + sl * 4;
#pragma unroll
                for (int c = 0; c < CHUNKS; ++c)
                    v[k][c] = *reinterpret_cast<const float4*>(xr + c * LANES * 4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int c = 0; c < CHUNKS; ++c) {
                    float4 t = v[k][c];
                    if (PRO) t = f4_affine_lrelu(t, pa[c], pb[c], slope);
                    acc[c].x = fmaf(wj[k], t.x, acc[c].x);
                    acc[c].y = fmaf(wj[k], t.y, acc[c].y);
                    acc[c].z = fmaf(wj[k], t.z, acc[c].z);
                    acc[c].w = fmaf(wj[k], t.w, acc[c].w);
                }
            }
        }
        const float di = dinv[row];
        float* yr = Y + (int64_t)row * ldy + sl * 4;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            float4 o;
            o.x = fmaf(acc[c].x, di, bs[c].x);
            o.y = fmaf(acc[c].y, di, bs[c].y);
            o.z = fmaf(acc[c].z, di, bs[c].z);
            o.w = fmaf(acc[c].w, di, bs[c].w);
            *reinterpret_cast<float4*>(yr + c * LANES * 4) = o;
        }
    }
}

// any width: one thread per (row, channel)
template <bool PRO>
__global__ __launch_bounds__(256) void spmm_scalar_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ dinv,
    const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, int n_rows, int C,
    const float* __restrict__ bias, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float slope) {
    const int64_t total = (int64_t)n_rows * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int row = (int)(idx / C), c = (int)(idx % C);
        float acc = 0.f;
        for (int e = rowptr[row]; e < rowptr[row + 1]; ++e) {
            const int j = col[e];
            float t = X[(int64_t)j * ldx + c];
            if (PRO) t = lrelu(fmaf(t, pscale[c], pshift[c]), slope);
            acc = fmaf(dinv[j], t, acc);
        }
        Y[(int64_t)row * ldy + c] = fmaf(acc, dinv[row], bias ? bias[c] : 0.f);
    }
}

template <int LANES, int CHUNKS>
int launch_vec(const ddmp_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy,
               const float* bias, const float* ps, const float* psh, float slope, hipStream_t st) {
    const int n = (int)g->n_rows;
    const int n_chunks = (int)cdiv(n, kRB);
    const int cpx = (int)cdiv(n_chunks, kXcd);
    dim3 grid(cpx * kXcd), block(256);
    if (ps)
        hipLaunchKernelGGL((spmm_vec_kernel<LANES, CHUNKS, true>), grid, block, 0, st, g->rowptr, g->col,
                           g->dinv, X, ldx, Y, ldy, n, bias, ps, psh, slope, cpx, n_chunks);
    else
        hipLaunchKernelGGL((spmm_vec_kernel<LANES, CHUNKS, false>), grid, block, 0, st, g->rowptr, g->col,
                           g->dinv, X, ldx, Y, ldy, n, bias, ps, psh, slope, cpx, n_chunks);
    LAUNCH_TRY();
    return DDMP_OK;
}

}  // namespace

extern "C" int ddmp_spmm_f32(const ddmp_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy,
                             int C, const float* bias, const float* pro_scale, const float* pro_shift,
                             float slope, ddmp_stream stream) {
    ARG_TRY(g && X && Y && C > 0 && ldx >= C && ldy >= C);
    ARG_TRY((pro_scale == nullptr) == (pro_shift == nullptr));
    ARG_TRY(X != Y);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) % 16 == 0) &&
                     (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0) &&
                     (!pro_scale || (reinterpret_cast<uintptr_t>(pro_scale) | reinterpret_cast<uintptr_t>(pro_shift)) % 16 == 0);
    if (vec) {
        switch (C) {
            case 8: return launch_vec<2, 1>(g, X, ldx, Y, ldy, bias, pro_scale, pro_shift, slope, st);
            case 16: return launch_vec<4, 1>(g, X, ldx, Y, ldy, bias, pro_scale, pro_shift, slope, st);
            case 32: return launch_vec<8, 1>(g, X, ldx, Y, ldy, bias, pro_scale, pro_shift, slope, st);
            case 64: return launch_vec<16, 1>(g, X, ldx, Y, ldy, bias, pro_scale, pro_shift, slope, st);
            case 128: return launch_vec<32, 1>(g, X, ldx, Y, ldy, bias, pro_scale, pro_shift, slope, st);
            case 256: return launch_vec<64, 1>(g, X, ldx, Y, ldy, bias, pro_scale, pro_shift, slope, st);
            case 512: return launch_vec<64, 2>(g, X, ldx, Y, ldy, bias, pro_scale, pro_shift, slope, st);
            default: break;
        }
    }
    const int64_t total = g->n_rows * (int64_t)C;
    const int grid = (int)std::min<int64_t>(cdiv(total, 256), 256 * 16);
    if (pro_scale)
        hipLaunchKernelGGL((spmm_scalar_kernel<true>), dim3(grid), dim3(256), 0, st, g->rowptr, g->col, g->dinv,
                           X, ldx, Y, ldy, (int)g->n_rows, C, bias, pro_scale, pro_shift, slope);
    else
        hipLaunchKernelGGL((spmm_scalar_kernel<false>), dim3(grid), dim3(256), 0, st, g->rowptr, g->col, g->dinv,
                           X, ldx, Y, ldy, (int)g->n_rows, C, bias, pro_scale, pro_shift, slope);
    LAUNCH_TRY();
    return DDMP_OK;
}
