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
#include "finalize.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

using namespace ddmp;

constexpr int kRB = 64;            // rows per workgroup chunk
constexpr int kMaxE = kRB * 16;    // staged CSR entries per chunk (mesh graphs: ~7 resp. 4 per row)

template <int LANES, int CHUNKS, bool PRO>
__global__ __launch_bounds__(256) void spmm_vec_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ dinv, const float* __restrict__ dinv_r,
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
        const float di = dinv_r[row];
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
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ dinv, const float* __restrict__ dinv_r,
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
        Y[(int64_t)row * ldy + c] = fmaf(acc, dinv_r[row], bias ? bias[c] : 0.f);
    }
}


// ------------------------------------------------------------------------------------------------
//  Slab-looping variant (C >= 64): same chunk ownership and LDS-staged CSR slice as the row kernel, but the
//  workgroup walks its rows one SLAB of CS channels at a time (CS*4 = 128 or 256 bytes = 1-2 cache lines per
//  row) and loops over the C/CS slabs itself.  Measured on the row kernel (rocprofv3 PMC, 1M-face mesh, Morton
//  numbering): at C = 512 the L2-miss traffic is 2.8x (face graph) to 4.4x (vertex graph) the algorithmic read
//  volume with a 35 % L2 hit rate, while at C = 32 it is 1.25x: with 2 KiB rows the 256 resident workgroups of
//  an XCD push ~67 MB through its 4 MiB L2 between two uses of a row.  Walking slab by slab shortens the reuse
//  TIME: a block step touches ~60 lines (fits L1), and the staged CSR slice is reused C/CS times.
// ------------------------------------------------------------------------------------------------
// U neighbours per batch, NR rows per lane group per step: U*NR gathers in flight per lane
// (vertex graph, degree 6+1: U = 8, NR = 1; face graph, degree 3+1: U = 4, NR = 2).
// SL slabs per pass: a lane fetches SL float4 (SL * 128 bytes apart) per neighbour, so the per-neighbour work
// (index / weight read from LDS, 64-bit address arithmetic, loop control) is paid once per SL * 16 bytes.
// RED: the output is a gradient dZ that a BatchNorm+LeakyReLU backward consumes next; the epilogue also produces that
// layer's column reductions (what bn_bwd_reduce computes from dZ and the layer's pre-BatchNorm output Yp):
//     red[chunk][0][c] = sum_rows g,   red[chunk][1][c] = sum_rows g * (Yp - mean) * rstd,   g = dZ * lrelu'(a Yp + b)
// float32 partials per wave and 64-row chunk, summed in float64 by fpartials_reduce_kernel.  Costs one streaming read of Yp
// here instead of a second pass over dZ and Yp.
struct BnRed {
    const float* Yp;
    int64_t ldyp;
    const float *scale, *shift, *mean, *rstd;
    float* part;
};

// BWD: the gathered operand is the BatchNorm+LeakyReLU BACKWARD of (X = dZ, bwd.Yb = the layer's pre-BatchNorm output),
//     dY = a * dZ * lrelu'(a * Yb + b) + c1 * Yb + c0   (a, b = pscale, pshift; element for element BwdApplyF of bn.hip),
// rebuilt on the gather instead of being written by bn_bwd_apply and read back (GEMM-first layers).
struct BnBwdGather {
    const float* Yb;
    int64_t ldyb;
    const float *c1, *c0;
};

template <int LANES, int U, int NR, bool PRO, int SL, bool RED = false, bool BWD = false>
__global__ __launch_bounds__(256) void spmm_slab_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ dinv, const float* __restrict__ dinv_r,
    const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, int n_rows, int C,
    const float* __restrict__ bias, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float slope, int chunks_per_xcd, int n_chunks, BnRed red = BnRed(), BnBwdGather bwd = BnBwdGather()) {
    static_assert(!BWD || (PRO && SL == 1 && !RED), "BWD: coefficients a, b come as the prologue's, one slab per pass");
    constexpr int CS = LANES * 4;
    constexpr int RPW = 64 / LANES;
    constexpr int RPB = 4 * RPW;
    __shared__ int s_rowptr[kRB + 1];
    __shared__ int s_col[kMaxE];
    __shared__ float s_w[kMaxE];
    static_assert(!RED || SL == 1, "the fused reduction walks one slab per pass");

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
    for (int c0 = 0; c0 < C; c0 += CS * SL) {
        const int off = c0 + sl * 4;
        float4 ra, rb, rmu, rrs, q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (RED) {
            ra = *reinterpret_cast<const float4*>(red.scale + off);
            rb = *reinterpret_cast<const float4*>(red.shift + off);
            rmu = *reinterpret_cast<const float4*>(red.mean + off);
            rrs = *reinterpret_cast<const float4*>(red.rstd + off);
        }
        float4 pa[SL], pb[SL], bs[SL];
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            pa[s] = make_float4(1.f, 1.f, 1.f, 1.f);
            pb[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (PRO) {
                pa[s] = *reinterpret_cast<const float4*>(pscale + off + s * CS);
                pb[s] = *reinterpret_cast<const float4*>(pshift + off + s * CS);
            }
            bs[s] = bias ? *reinterpret_cast<const float4*>(bias + off + s * CS) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 k1 = make_float4(0.f, 0.f, 0.f, 0.f), k0 = k1;
        if (BWD) {
            k1 = *reinterpret_cast<const float4*>(bwd.c1 + off);
            k0 = *reinterpret_cast<const float4*>(bwd.c0 + off);
        }
        const float* xc = X + off;
        const float* yc = BWD ? bwd.Yb + off : nullptr;
        for (int lr0 = wave * RPW + grp; lr0 < nr; lr0 += RPB * NR) {
            int es[NR], ee[NR];
            float4 acc[NR][SL];
            int more = 0;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int lr = min(lr0 + q * RPB, nr - 1);
                es[q] = s_rowptr[lr] - e0;
                ee[q] = (lr0 + q * RPB < nr) ? s_rowptr[lr + 1] - e0 : es[q];     // rows past the chunk: empty
#pragma unroll
                for (int s = 0; s < SL; ++s) acc[q][s] = make_float4(0.f, 0.f, 0.f, 0.f);
                more |= (ee[q] > es[q]);
            }
            while (more) {
                int cj[NR][U];
                float wj[NR][U];
#pragma unroll
                for (int q = 0; q < NR; ++q)
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        const int ek = max(min(es[q] + k, ee[q] - 1), 0);
                        if (staged) {
                            cj[q][k] = s_col[ek];
                            wj[q][k] = s_w[ek];
                        } else {
                            cj[q][k] = col[e0 + ek];
                            wj[q][k] = dinv[cj[q][k]];
                        }
                        if (es[q] + k >= ee[q]) wj[q][k] = 0.f;
                    }
                float4 v[NR][U][SL], vy[BWD ? NR : 1][BWD ? U : 1];
#pragma unroll
                for (int q = 0; q < NR; ++q)
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        const float* xr = xc + (int64_t)cj[q][k] * ldx;
#pragma unroll
                        for (int s = 0; s < SL; ++s) v[q][k][s] = *reinterpret_cast<const float4*>(xr + s * CS);
                        if (BWD) vy[BWD ? q : 0][BWD ? k : 0] = *reinterpret_cast<const float4*>(yc + (int64_t)cj[q][k] * bwd.ldyb);
                    }
                more = 0;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
#pragma unroll
                    for (int k = 0; k < U; ++k)
#pragma unroll
                        for (int s = 0; s < SL; ++s) {
                            float4 t = v[q][k][s];
                            if (BWD) {
                                const float4 y = vy[BWD ? q : 0][BWD ? k : 0], a = pa[s], b = pb[s];
                                t.x = fmaf(a.x, t.x * lrelu_grad(fmaf(y.x, a.x, b.x), slope), fmaf(k1.x, y.x, k0.x));
                                t.y = fmaf(a.y, t.y * lrelu_grad(fmaf(y.y, a.y, b.y), slope), fmaf(k1.y, y.y, k0.y));
                                t.z = fmaf(a.z, t.z * lrelu_grad(fmaf(y.z, a.z, b.z), slope), fmaf(k1.z, y.z, k0.z));
                                t.w = fmaf(a.w, t.w * lrelu_grad(fmaf(y.w, a.w, b.w), slope), fmaf(k1.w, y.w, k0.w));
                            } else if (PRO) t = f4_affine_lrelu(t, pa[s], pb[s], slope);
                            acc[q][s].x = fmaf(wj[q][k], t.x, acc[q][s].x);
                            acc[q][s].y = fmaf(wj[q][k], t.y, acc[q][s].y);
                            acc[q][s].z = fmaf(wj[q][k], t.z, acc[q][s].z);
                            acc[q][s].w = fmaf(wj[q][k], t.w, acc[q][s].w);
                        }
                    es[q] = min(es[q] + U, ee[q]);
                    more |= (ee[q] > es[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int lr = lr0 + q * RPB;
                if (lr < nr) {
                    const int row = r0 + lr;
                    const float di = dinv_r[row];
#pragma unroll
                    for (int s = 0; s < SL; ++s) {
                        float4 o;
                        o.x = fmaf(acc[q][s].x, di, bs[s].x);
                        o.y = fmaf(acc[q][s].y, di, bs[s].y);
                        o.z = fmaf(acc[q][s].z, di, bs[s].z);
                        o.w = fmaf(acc[q][s].w, di, bs[s].w);
                        nt_store4(Y + (int64_t)row * ldy + off + s * CS, o);
                        if (RED) {
                            const float4 y = *reinterpret_cast<const float4*>(red.Yp + (int64_t)row * red.ldyp + off);
                            const float g0 = o.x * lrelu_grad(fmaf(y.x, ra.x, rb.x), slope);
                            const float g1 = o.y * lrelu_grad(fmaf(y.y, ra.y, rb.y), slope);
                            const float g2 = o.z * lrelu_grad(fmaf(y.z, ra.z, rb.z), slope);
                            const float g3 = o.w * lrelu_grad(fmaf(y.w, ra.w, rb.w), slope);
                            q0.x += g0; q0.y += g1; q0.z += g2; q0.w += g3;
                            q1.x = fmaf(g0, (y.x - rmu.x) * rrs.x, q1.x);
                            q1.y = fmaf(g1, (y.y - rmu.y) * rrs.y, q1.y);
                            q1.z = fmaf(g2, (y.z - rmu.z) * rrs.z, q1.z);
                            q1.w = fmaf(g3, (y.w - rmu.w) * rrs.w, q1.w);
                        }
                    }
                }
            }
        }
        if (RED) {                                               // this wave's rows -> one partial per channel and wave
            float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};      // (no workgroup barrier: the slab loop
#pragma unroll                                                                  //  of the four waves stays decoupled)
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int o = LANES; o < 64; o <<= 1) v[e] += __shfl_xor(v[e], o, 64);     // over the row groups of the wave
            if (grp == 0) {
                float* pp = red.part + ((int64_t)(chunk * 4 + wave) * 2) * C + off;
                *reinterpret_cast<float4*>(pp) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(pp + C) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    }
}

#include "fpartials.inc"
#include "spmm_lean.inc"

// chunk_list / n_list: only these chunks (the LDS-patch kernel's heavy list), one workgroup each
template <bool PRO, int RED, bool BWD>
int launch_lean(const LeanPlan& lp, const ddmp_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy, int C,
                const float* bias, const float* ps, const float* psh, float slope, hipStream_t st, BnRed red = BnRed(),
                BnBwdGather bwd = BnBwdGather(), const int* chunk_list = nullptr, int n_list = 0) {
    const int n = (int)g->n_rows;
    if (chunk_list) {
        if (n_list <= 0) return DDMP_OK;
        // a handful of chunks (the regular 1M-face mesh in RCB order leaves ~100 of 15,625 face chunks to this launch): one
        // workgroup per chunk would walk its C / 32 slabs in sequence on ~100 CUs while the rest of the chip idles -- the slabs of a
        // chunk are split over workgroups (the kernel's slab groups), ~2048 workgroups in all
        const int n_sl = std::max(C / 32, 1);
        const int want = std::min(n_sl, std::max(1, 2048 / n_list));
        const int per = (n_sl + want - 1) / want;
        const int groups_l = (n_sl + per - 1) / per;
        hipLaunchKernelGGL((spmm_lean_kernel<PRO, RED, BWD>), dim3(n_list, groups_l), dim3(256), 0, st, g->rowptr, g->col, g->ew, g->dinv, g->dinv_r, X,
                           ldx, Y, ldy, n, C, bias, ps, psh, slope, 0, n_list, chunk_list, red, bwd);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    const int cpx = (int)cdiv(lp.n_chunks, kXcd);
    // small graphs: split the slabs of a chunk over several workgroups (the staged CSR slice is cheap to repeat) until ~2048
    // workgroups exist -- a workgroup per 64 rows alone leaves a 13k-row graph with 205 long-running workgroups on 256 CUs
    int groups = 1;
    const int n_slabs = C / 32;
    if (lp.n_chunks < 1024 && n_slabs > 1) {
        const int want = std::min(n_slabs, std::max(1, 2048 / std::max(lp.n_chunks, 1)));
        const int per = (n_slabs + want - 1) / want;
        groups = (n_slabs + per - 1) / per;
    }
    hipLaunchKernelGGL((spmm_lean_kernel<PRO, RED, BWD>), dim3(cpx * kXcd, groups), dim3(256), 0, st, g->rowptr, g->col, g->ew, g->dinv, g->dinv_r, X, ldx,
                       Y, ldy, n, C, bias, ps, psh, slope, cpx, lp.n_chunks, (const int*)nullptr, red, bwd);
    LAUNCH_TRY();
    return DDMP_OK;
}

template <int LANES, int U, int NR, int SL = 1>
int launch_slab(const ddmp_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy, int C, const float* bias,
                const float* ps, const float* psh, float slope, hipStream_t st) {
    const int n = (int)g->n_rows;
    const int n_chunks = (int)cdiv(n, kRB);
    const int cpx = (int)cdiv(n_chunks, kXcd);
    dim3 grid(cpx * kXcd), block(256);
    if (LANES == 8 && U == 4 && NR == 1 && SL == 1) {
        const LeanPlan lp = lean_plan(g, ldx, ldy, C);
        if (lp.kind && ps) return launch_lean<true, 0, false>(lp, g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st);
        if (lp.kind) return launch_lean<false, 0, false>(lp, g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st);
    }
    if (ps)
        hipLaunchKernelGGL((spmm_slab_kernel<LANES, U, NR, true, SL>), grid, block, 0, st, g->rowptr, g->col, g->dinv, g->dinv_r, X,
                           ldx, Y, ldy, n, C, bias, ps, psh, slope, cpx, n_chunks);
    else
        hipLaunchKernelGGL((spmm_slab_kernel<LANES, U, NR, false, SL>), grid, block, 0, st, g->rowptr, g->col, g->dinv, g->dinv_r, X,
                           ldx, Y, ldy, n, C, bias, ps, psh, slope, cpx, n_chunks);
    LAUNCH_TRY();
    return DDMP_OK;
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
                           g->dinv, g->dinv_r, X, ldx, Y, ldy, n, bias, ps, psh, slope, cpx, n_chunks);
    else
        hipLaunchKernelGGL((spmm_vec_kernel<LANES, CHUNKS, false>), grid, block, 0, st, g->rowptr, g->col,
                           g->dinv, g->dinv_r, X, ldx, Y, ldy, n, bias, ps, psh, slope, cpx, n_chunks);
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
    if (vec && C % 32 == 0) {                  // LDS-patch kernel (spmm_patch.hip) where it applies
        const LeanPlan lp = lean_plan(g, ldx, ldy, C);           // (its heavy chunks are the lean gather's)
        if (lp.kind || g->n_heavy == 0) {
            const int rc = ddmp::spmm_patch(g, X, ldx, Y, ldy, C, DDMP_F32, bias, pro_scale, pro_shift, slope, nullptr, 0, nullptr,
                                            nullptr, nullptr, nullptr, nullptr, st);
            if (rc == DDMP_OK && g->n_heavy > 0)
                return pro_scale ? launch_lean<true, 0, false>(lp, g, X, ldx, Y, ldy, C, bias, pro_scale, pro_shift, slope, st, BnRed(),
                                                               BnBwdGather(), g->heavy, g->n_heavy)
                                 : launch_lean<false, 0, false>(lp, g, X, ldx, Y, ldy, C, bias, nullptr, nullptr, slope, st, BnRed(),
                                                                BnBwdGather(), g->heavy, g->n_heavy);
            if (rc != ddmp::kPatchNotApplicable) return rc;
        }
    }
    if (vec && C >= 32 && C % 32 == 0) {
        // 4 gathers in flight per lane, one 128-byte slab per pass (8 in flight -- one batch of 8, or two rows x 4 -- and two / four
        // slabs per pass measured 3-5 % slower in rounds 2-3: bound by on-chip issue / L1 cost per gathered row, not by latency)
        return launch_slab<8, 4, 1>(g, X, ldx, Y, ldy, C, bias, pro_scale, pro_shift, slope, st);
    }
    if (vec) {                                                   // (widths below one 128-byte slab)
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
        hipLaunchKernelGGL((spmm_scalar_kernel<true>), dim3(grid), dim3(256), 0, st, g->rowptr, g->col, g->dinv, g->dinv_r,
                           X, ldx, Y, ldy, (int)g->n_rows, C, bias, pro_scale, pro_shift, slope);
    else
        hipLaunchKernelGGL((spmm_scalar_kernel<false>), dim3(grid), dim3(256), 0, st, g->rowptr, g->col, g->dinv, g->dinv_r,
                           X, ldx, Y, ldy, (int)g->n_rows, C, bias, pro_scale, pro_shift, slope);
    LAUNCH_TRY();
    return DDMP_OK;
}


// SpMM whose output dZ feeds a BatchNorm+LeakyReLU backward: also returns that layer's column reductions
// (= ddmp_bn_bwd_reduce_f32(Y, Yp, ...)).  Fused into the slab kernel's epilogue where that kernel runs.
extern "C" size_t ddmp_spmm_bnred_workspace_bytes(int64_t n_rows, int C) {
    if (n_rows <= 0 || C <= 0) return 0;
    const size_t fused = (size_t)ddmp::cdiv(n_rows, kRB) * 4 * 2 * (size_t)C * sizeof(float) + 256 + fpartials_mid_bytes(C);
    return std::max(fused, ddmp_colreduce_workspace_bytes(n_rows, C));
}

extern "C" int ddmp_spmm_bnred_f32(const ddmp_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy, int C,
                                   const float* Yp, int64_t ldyp, const float* scale, const float* shift,
                                   const float* mean, const float* rstd, float slope, double* sums2, void* ws,
                                   size_t ws_bytes, ddmp_stream stream) {
    ddmp::FinalizeScope fin_scope(sums2, stream, C);
    ARG_TRY(g && X && Y && Yp && scale && shift && mean && rstd && sums2 && ws && C > 0 && ldx >= C && ldy >= C && ldyp >= C);
    ARG_TRY(X != Y);
    if (ws_bytes < ddmp_spmm_bnred_workspace_bytes(g->n_rows, C)) return DDMP_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool fused = C % 32 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldyp % 4 == 0 && al16(X) && al16(Y) && al16(Yp) &&
                       al16(scale) && al16(shift) && al16(mean) && al16(rstd) && al16(ws);
    if (!fused) {
        int rc = ddmp_spmm_f32(g, X, ldx, Y, ldy, C, nullptr, nullptr, nullptr, slope, stream);
        if (rc != DDMP_OK) return rc;
        return ddmp_bn_bwd_reduce_f32(Y, ldy, Yp, ldyp, g->n_rows, C, scale, shift, mean, rstd, slope, sums2, ws, ws_bytes,
                                      stream);
    }
    const int n = (int)g->n_rows;
    const int n_chunks = (int)cdiv(n, kRB);
    const int cpx = (int)cdiv(n_chunks, kXcd);
    BnRed red{Yp, ldyp, scale, shift, mean, rstd, (float*)ws};
    const LeanPlan lp = lean_plan(g, ldx, ldy, C, 2);
    if (lp.kind || g->n_heavy == 0) {                            // LDS-patch kernel: one record per chunk, like the lean gather
        int rc = ddmp::spmm_patch(g, X, ldx, Y, ldy, C, DDMP_F32, nullptr, nullptr, nullptr, slope, Yp, ldyp, scale, shift,
                                  mean, rstd, (float*)ws, st);
        if (rc == DDMP_OK && g->n_heavy > 0)
            rc = launch_lean<false, 1, false>(lp, g, X, ldx, Y, ldy, C, nullptr, nullptr, nullptr, slope, st, red, BnBwdGather(),
                                              g->heavy, g->n_heavy);
        if (rc == DDMP_OK) {
            const size_t pb2 = ((size_t)n_chunks * 4 * 2 * (size_t)C * sizeof(float) + 255) / 256 * 256;
            fpartials_reduce((const float*)ws, n_chunks + g->n_split, C, C, (double*)((char*)ws + pb2), sums2, st);   // (+ the split chunks' second records)
            LAUNCH_TRY();
            return DDMP_OK;
        }
        if (rc != ddmp::kPatchNotApplicable) return rc;
    }
    int red_groups = n_chunks * 4;                               // slab kernel: one record per wave and chunk
    if (lp.kind) {
        const int rc = launch_lean<false, 1, false>(lp, g, X, ldx, Y, ldy, C, nullptr, nullptr, nullptr, slope, st, red);
        if (rc != DDMP_OK) return rc;
        red_groups = lp.n_chunks;                                // lean kernel: one record per chunk
    } else {
        hipLaunchKernelGGL((spmm_slab_kernel<8, 4, 1, false, 1, true>), dim3(cpx * kXcd), dim3(256), 0, st, g->rowptr, g->col,
                           g->dinv, g->dinv_r, X, ldx, Y, ldy, n, C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           slope, cpx, n_chunks, red);
        LAUNCH_TRY();
    }
    const size_t pbytes = ((size_t)n_chunks * 4 * 2 * (size_t)C * sizeof(float) + 255) / 256 * 256;
    fpartials_reduce((const float*)ws, red_groups, C, C, (double*)((char*)ws + pbytes), sums2, st);
    LAUNCH_TRY();
    return DDMP_OK;
}

// SpMM whose output is a conv output Y (transform-first layers, forward): also returns its BatchNorm statistics
// (= ddmp_bn_stats_f32(Y)), from the lean kernel's epilogue around the per-column reference `ref` (spmm_lean.inc, RED = 2).
extern "C" int ddmp_spmm_stats_supported(int C) { return (C % 32 == 0 && C >= 32 && C <= 1024) ? 1 : 0; }

extern "C" int ddmp_spmm_stats_f32(const ddmp_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy, int C,
                                   const float* bias, const float* pro_scale, const float* pro_shift, float slope,
                                   const float* ref, double* sums2, void* ws, size_t ws_bytes, ddmp_stream stream) {
    ddmp::FinalizeScope fin_scope(sums2, stream, C);
    ARG_TRY(g && X && Y && sums2 && ws && C > 0 && ldx >= C && ldy >= C && X != Y);
    ARG_TRY((pro_scale == nullptr) == (pro_shift == nullptr));
    if (ws_bytes < ddmp_spmm_bnred_workspace_bytes(g->n_rows, C)) return DDMP_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const LeanPlan lp = lean_plan(g, ldx, ldy, C, 2);
    const bool fused = ref && ddmp_spmm_stats_supported(C) && lp.kind && ldx % 4 == 0 && ldy % 4 == 0 && al16(X) && al16(Y) &&
                       al16(ref) && al16(ws) && al16(bias) && al16(pro_scale) && al16(pro_shift);
    if (!fused) {
        int rc = ddmp_spmm_f32(g, X, ldx, Y, ldy, C, bias, pro_scale, pro_shift, slope, stream);
        if (rc != DDMP_OK) return rc;
        return ddmp_bn_stats_f32(Y, ldy, g->n_rows, C, sums2, ws, ws_bytes, stream);
    }
    BnRed red{nullptr, 0, nullptr, nullptr, ref, nullptr, (float*)ws};
    // LDS-patch kernel where selected (one record per chunk, like the lean gather; its heavy chunks: the lean gather's list)
    int rc = ddmp::spmm_patch(g, X, ldx, Y, ldy, C, DDMP_F32, bias, pro_scale, pro_shift, slope, nullptr, 0, nullptr, nullptr, ref,
                              nullptr, (float*)ws, st);
    const bool patched = rc == DDMP_OK;
    if (rc != DDMP_OK && rc != ddmp::kPatchNotApplicable) return rc;
    const int* list = patched ? g->heavy : nullptr;
    if (!patched || g->n_heavy > 0)
        rc = pro_scale ? launch_lean<true, 2, false>(lp, g, X, ldx, Y, ldy, C, bias, pro_scale, pro_shift, slope, st, red, BnBwdGather(),
                                                     list, g->n_heavy)
                       : launch_lean<false, 2, false>(lp, g, X, ldx, Y, ldy, C, bias, nullptr, nullptr, slope, st, red, BnBwdGather(),
                                                      list, g->n_heavy);
    if (rc != DDMP_OK) return rc;
    const size_t pbytes = ((size_t)lp.n_chunks * 4 * 2 * (size_t)C * sizeof(float) + 255) / 256 * 256;
    // (a split chunk of the LDS-patch kernel has a second record behind the chunks': the records' area is sized for four per chunk)
    fpartials_reduce((const float*)ws, lp.n_chunks + (patched ? g->n_split : 0), C, C, (double*)((char*)ws + pbytes), sums2, st, ref,
                     (double)g->n_rows);
    LAUNCH_TRY();
    return DDMP_OK;
}

// out = A_hat . dY with dY = BatchNorm+LeakyReLU backward of (dZ, Yb) rebuilt on the gather (see BnBwdGather)
extern "C" int ddmp_spmm_bnbwd_supported(int C) { return (C % 32 == 0 && C >= 32) ? 1 : 0; }

extern "C" int ddmp_spmm_bnbwd_f32(const ddmp_graph* g, const float* dZ, int64_t lddz, const float* Yb, int64_t ldyb,
                                   float* out, int64_t ld_out, int C, const float* a, const float* b, const float* c1,
                                   const float* c0, float slope, ddmp_stream stream) {
    ARG_TRY(g && dZ && Yb && out && a && b && c1 && c0 && C > 0 && lddz >= C && ldyb >= C && ld_out >= C);
    ARG_TRY(dZ != out && Yb != out);
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!ddmp_spmm_bnbwd_supported(C) || lddz % 4 || ldyb % 4 || ld_out % 4 || !al16(dZ) || !al16(Yb) || !al16(out) ||
        !al16(a) || !al16(b) || !al16(c1) || !al16(c0))
        return DDMP_EINVAL;
    const int n = (int)g->n_rows;
    const int n_chunks = (int)cdiv(n, kRB);
    const int cpx = (int)cdiv(n_chunks, kXcd);
    BnBwdGather bwd{Yb, ldyb, c1, c0};
    const LeanPlan lp = lddz == ldyb ? lean_plan(g, lddz, ld_out, C, 4) : LeanPlan{0, 0};   // one staged offset serves both matrices
    if (lp.kind || g->n_heavy == 0) {                            // LDS-patch form where it applies; its heavy chunks: the lean form
        const int rc = ddmp::spmm_patch_bwd(g, dZ, lddz, Yb, ldyb, out, ld_out, C, DDMP_F32, a, b, c1, c0, slope, (hipStream_t)stream);
        if (rc == DDMP_OK && g->n_heavy > 0)
            return launch_lean<true, 0, true>(lp, g, dZ, lddz, out, ld_out, C, nullptr, a, b, slope, (hipStream_t)stream, BnRed(), bwd,
                                              g->heavy, g->n_heavy);
        if (rc != ddmp::kPatchNotApplicable) return rc;
    }
    if (lp.kind) return launch_lean<true, 0, true>(lp, g, dZ, lddz, out, ld_out, C, nullptr, a, b, slope, (hipStream_t)stream, BnRed(), bwd);
    hipLaunchKernelGGL((spmm_slab_kernel<8, 4, 1, true, 1, false, true>), dim3(cpx * kXcd), dim3(256), 0,
                       (hipStream_t)stream, g->rowptr, g->col, g->dinv, g->dinv_r, dZ, lddz, out, ld_out, n, C, (const float*)nullptr,
                       a, b, slope, cpx, n_chunks, BnRed(), bwd);
    LAUNCH_TRY();
    return DDMP_OK;
}
