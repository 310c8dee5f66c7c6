// Geometric losses of the training step (util/loss.py of the reference) and their analytic gradients.
//
//   L1 pos_rec    (:16-35)   sqrt( sum_v |real_v - pos_v|^2 / V + 1e-6 )         float64 (real is f64)
//   L2 laplacian  (:37-53)   sqrt( sum_v |pos_v - mean_{j in N(v)} pos_j|^2 / V + 1e-12 )
//   L3 norm_rec   (:55-84)   sum_f |n_f - real_f|_1 / F                          float64
//   L4 fn_bnf     (:86-138)  `loop` bilateral-filter passes over the <=3 edge neighbours, L1 distance
//                            to the input normals / F; pos is a constant; -1 slots of f2f gather the
//                            LAST face and count in sigma_c but carry zero weight
//   L5 pos_norm   (:140-160) sum_f sum_k |(p_fk - c_f) . n_f| / V
//
// All five are a few passes over [V,3] / [F,3] arrays (<2 % of the step).  Every reduction is
// per-block partial sums in float64 + one finalize kernel (no atomics, deterministic); every gradient
// is written in gather form (vertex -> incident faces / 1-ring, face -> 3 neighbours) instead of the
// reference's scatter-add.  f2f must be symmetric on its valid entries (it is for edge adjacency).
#include "ddmp_common.h"

#include <algorithm>

namespace {

using namespace ddmp;

constexpr int kNB = 256;                 // blocks of every reducing loss kernel
constexpr float kSigmaS2 = 0.3f * 0.3f;  // util/loss.py:110

enum { P_S1 = 0, P_S2, P_S3, P_S4, P_S5, P_SIG, P_COUNT };

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 ld3(const float* p, int64_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void st3(float* p, int64_t i, F3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }
__device__ __forceinline__ double sgn(double x) { return (x > 0.0) - (x < 0.0); }

__device__ __forceinline__ void put_partial(double v, double* partials, int which, double* sm) {
    const double t = block_sum(v, sm);
    if (threadIdx.x == 0) partials[which * kNB + blockIdx.x] = t;
}
__device__ __forceinline__ double sum_partials(const double* partials, int which) {
    // every thread walks the same kNB doubles (L2-resident, 2 KB)
    double t = 0.0;
    for (int i = 0; i < kNB; ++i) t += partials[which * kNB + i];
    return t;
}

__global__ __launch_bounds__(256) void vertex_fwd_kernel(int V, const float* __restrict__ pos,
                                                         const double* __restrict__ real,
                                                         const int* __restrict__ vv_ptr,
                                                         const int* __restrict__ vv_idx, float* __restrict__ resid,
                                                         double* __restrict__ partials,
                                                         const unsigned char* __restrict__ own /* nullable: sums over own[v] != 0 */) {
    __shared__ double sm[4];
    double s1 = 0.0, s2 = 0.0;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += kNB * 256) {
        const double m = (own && !own[v]) ? 0.0 : 1.0;           // ghost rows of a shard are computed, not counted
        const F3 p = ld3(pos, v);
        const double dx = real[3 * (int64_t)v] - (double)p.x, dy = real[3 * (int64_t)v + 1] - (double)p.y,
                     dz = real[3 * (int64_t)v + 2] - (double)p.z;
        s1 += m * (dx * dx + dy * dy + dz * dz);
        F3 acc = {0.f, 0.f, 0.f};
        const int b = vv_ptr[v], e = vv_ptr[v + 1];
        for (int k = b; k < e; ++k) acc = acc + ld3(pos, vv_idx[k]);
        const float deg = (float)(e - b);
        const F3 r = {p.x - acc.x / deg, p.y - acc.y / deg, p.z - acc.z / deg};
        st3(resid, v, r);
        s2 += m * (double)(r.x * r.x + r.y * r.y + r.z * r.z);
    }
    put_partial(s1, partials, P_S1, sm);
    put_partial(s2, partials, P_S2, sm);
}

__global__ __launch_bounds__(256) void face_fwd_kernel(int F, const float* __restrict__ pos,
                                                       const float* __restrict__ nrm,
                                                       const double* __restrict__ real_n,
                                                       const int* __restrict__ faces, float* __restrict__ fc,
                                                       float* __restrict__ fa, float* __restrict__ pn_coef,
                                                       float* __restrict__ pn_dn, double* __restrict__ partials,
                                                       const unsigned char* __restrict__ own) {
    __shared__ double sm[4];
    double s3 = 0.0, s5 = 0.0;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += kNB * 256) {
        const double m = (own && !own[f]) ? 0.0 : 1.0;
        const int i0 = faces[3 * (int64_t)f], i1 = faces[3 * (int64_t)f + 1], i2 = faces[3 * (int64_t)f + 2];
        const F3 p0 = ld3(pos, i0), p1 = ld3(pos, i1), p2 = ld3(pos, i2);
        const F3 c = {(p0.x + p1.x + p2.x) / 3.0f, (p0.y + p1.y + p2.y) / 3.0f, (p0.z + p1.z + p2.z) / 3.0f};
        const F3 a = p1 - p0, b = p2 - p0;
        const F3 cr = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
        st3(fc, f, c);
        fa[f] = 0.5f * sqrtf(dot(cr, cr) + 1.0e-12f);
        const F3 n = ld3(nrm, f);
        s3 += m * (fabs((double)n.x - real_n[3 * (int64_t)f]) + fabs((double)n.y - real_n[3 * (int64_t)f + 1]) +
                   fabs((double)n.z - real_n[3 * (int64_t)f + 2]));
        const F3 q0 = p0 - c, q1 = p1 - c, q2 = p2 - c;
        const float d0 = dot(q0, n), d1 = dot(q1, n), d2 = dot(q2, n);
        s5 += m * (double)(fabsf(d0) + fabsf(d1) + fabsf(d2));
        const float g0 = sgn(d0), g1 = sgn(d1), g2 = sgn(d2);
        const float gs = (g0 + g1 + g2) / 3.0f;
        st3(pn_coef, f, F3{g0 - gs, g1 - gs, g2 - gs});
        st3(pn_dn, f, g0 * q0 + g1 * q1 + g2 * q2);
    }
    put_partial(s3, partials, P_S3, sm);
    put_partial(s5, partials, P_S5, sm);
}

__global__ __launch_bounds__(256) void bnf_sigma_kernel(int F, const float* __restrict__ fc,
                                                        const int* __restrict__ f2f, float* __restrict__ fcd,
                                                        double* __restrict__ partials,
                                                        const unsigned char* __restrict__ own) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += kNB * 256) {
        const double m = (own && !own[f]) ? 0.0 : 1.0;
        const F3 c = ld3(fc, f);
        float d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int j = f2f[3 * (int64_t)f + k];
            if (j < 0) j = F - 1;                       // python negative index: the last face
            const F3 t = ld3(fc, j) - c;
            d[k] = dot(t, t);
            s += m * (double)sqrtf(d[k] + 1.0e-12f);
        }
        st3(fcd, f, F3{d[0], d[1], d[2]});
    }
    put_partial(s, partials, P_SIG, sm);
}

__device__ __forceinline__ float sigma_c_inv2(const double* partials, int F) {
    const float sc = (float)(sum_partials(partials, P_SIG) / (3.0 * (double)F));
    return 1.0f / (2.0f * sc * sc);
}

__global__ __launch_bounds__(256) void bnf_iter_kernel(int F, const float* __restrict__ cur,
                                                       const int* __restrict__ f2f,
                                                       const float* __restrict__ fcd, const float* __restrict__ fa,
                                                       const double* __restrict__ partials,
                                                       float* __restrict__ Aout, float* __restrict__ next, int F_glob) {
    const float i2sc = sigma_c_inv2(partials, F_glob);           // sigma_c is a mean over ALL faces of the mesh
    const float i2ss = 1.0f / (2.0f * kSigmaS2);
    for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += gridDim.x * 256) {
        const F3 nf = ld3(cur, f);
        const F3 dd = ld3(fcd, f);
        const float dk[3] = {dd.x, dd.y, dd.z};
        F3 A = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = f2f[3 * (int64_t)f + k];
            const int jj = j < 0 ? F - 1 : j;
            const F3 nj = ld3(cur, jj);
            const F3 t = nj - nf;
            const float w = expf(-dk[k] * i2sc) * expf(-dot(t, t) * i2ss) * (j < 0 ? 0.f : fa[jj]);
            A = A + w * nj;
        }
        st3(Aout, f, A);
        const float q = sqrtf(dot(A, A) + 1.0e-12f);
        const float inv = 1.0f / (q + 1.0e-12f);
        st3(next, f, inv * A);
    }
}

__global__ __launch_bounds__(256) void bnf_diff_kernel(int F, const float* __restrict__ n_last,
                                                       const float* __restrict__ n0,
                                                       double* __restrict__ partials,
                                                       const unsigned char* __restrict__ own) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 3 * (int64_t)F; i += kNB * 256)
        if (!own || own[i / 3]) s += (double)fabsf(n_last[i] - n0[i]);
    put_partial(s, partials, P_S4, sm);
}

// ---- finalize: partials -> loss values, weighted total and gradient coefficients
// lossbuf (double): [0..4] L1..L5 (L4 ungated)  [5] total  [6..10] c1..c5  [11] sigma_c
__global__ void loss_finalize_kernel(const double* __restrict__ partials, int V, int F, double k1, double k2,
                                     double k3, double k4, double k5, double gate4,
                                     double* __restrict__ lossbuf) {
    // one lane per quantity, each walking its partials in the order sum_partials defines (the six walks one after the other
    // in one thread took 26 us of dependent L2 loads on the join point of every iteration)
    __shared__ double Ss[P_COUNT];
    if (blockIdx.x != 0) return;
    if (threadIdx.x < P_COUNT) Ss[threadIdx.x] = sum_partials(partials, threadIdx.x);
    __syncthreads();
    if (threadIdx.x != 0) return;
    double S[P_COUNT];
    for (int w = 0; w < P_COUNT; ++w) S[w] = Ss[w];
    const double L1 = sqrt(S[P_S1] / V + 1.0e-6);
    const float L2 = sqrtf((float)(S[P_S2] / V) + 1.0e-12f);
    const double L3 = S[P_S3] / F;
    const float L4 = (float)(S[P_S4] / F);
    const float L5 = (float)(S[P_S5] / V);
    lossbuf[0] = L1; lossbuf[1] = L2; lossbuf[2] = L3; lossbuf[3] = L4; lossbuf[4] = L5;
    lossbuf[5] = k1 * L1 + k2 * (double)L2 + k3 * L3 + k4 * ((double)L4 * gate4) + k5 * (double)L5;
    lossbuf[6] = k1 / ((double)V * L1);
    lossbuf[7] = k2 / ((double)V * (double)L2);
    lossbuf[8] = k3 / (double)F;
    lossbuf[9] = k4 * gate4 / (double)F;
    lossbuf[10] = k5 / (double)V;
    lossbuf[11] = S[P_SIG] / (3.0 * (double)F);
}

// ---- BNF backward
__global__ __launch_bounds__(256) void bnf_bwd_init_kernel(int F, const float* __restrict__ n_last,
                                                           const float* __restrict__ n0,
                                                           const double* __restrict__ coef4, float* __restrict__ G) {
    const float c4 = (float)coef4[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 3 * (int64_t)F; i += (int64_t)gridDim.x * 256)
        G[i] = c4 * sgn(n_last[i] - n0[i]);
}

__global__ __launch_bounds__(256) void bnf_bwd_dA_kernel(int F, const float* __restrict__ G,
                                                         const float* __restrict__ A, float* __restrict__ dA) {
    for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += gridDim.x * 256) {
        const F3 g = ld3(G, f), a = ld3(A, f);
        const float q = sqrtf(dot(a, a) + 1.0e-12f);
        const float den = q + 1.0e-12f;
        const float k = dot(g, a) / (den * den * q);
        st3(dA, f, (1.0f / den) * g - k * a);
    }
}

__global__ __launch_bounds__(256) void bnf_bwd_gather_kernel(int F, const float* __restrict__ cur,
                                                             const float* __restrict__ dA,
                                                             const int* __restrict__ f2f,
                                                             const float* __restrict__ fcd,
                                                             const float* __restrict__ fa,
                                                             const double* __restrict__ partials,
                                                             float* __restrict__ Gout, int F_glob) {
    const float i2sc = sigma_c_inv2(partials, F_glob);
    const float i2ss = 1.0f / (2.0f * kSigmaS2);
    const float iss = 1.0f / kSigmaS2;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < F; j += gridDim.x * 256) {
        const F3 nj = ld3(cur, j), dAj = ld3(dA, j);
        const F3 dd = ld3(fcd, j);
        const float dk[3] = {dd.x, dd.y, dd.z};
        const float faj = fa[j];
        F3 g = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int f = f2f[3 * (int64_t)j + k];
            if (f < 0) continue;
            const F3 nf = ld3(cur, f), dAf = ld3(dA, f);
            const F3 t = nj - nf;
            const float ww = expf(-dk[k] * i2sc) * expf(-dot(t, t) * i2ss);
            // j as a neighbour of f: weight carries j's own area
            const float Wfj = ww * faj;
            const float sfj = dot(dAf, nj) * Wfj * iss;
            g = g + Wfj * dAf - sfj * t;
            // j as the centre, f as its neighbour: weight carries f's area
            const float Wjf = ww * fa[f];
            const float sjf = dot(dAj, nf) * Wjf * iss;
            g = g - sjf * t;        // s_jf * (n_f - n_j)
        }
        st3(Gout, j, g);
    }
}

__global__ __launch_bounds__(256) void face_bwd_kernel(int F, const float* __restrict__ nrm,
                                                       const double* __restrict__ real_n,
                                                       const float* __restrict__ pn_dn,
                                                       const float* __restrict__ G0, const float* __restrict__ n_last,
                                                       const double* __restrict__ coef /*c1..c5*/,
                                                       float* __restrict__ dnorm) {
    const double c3 = coef[2];
    const float c4 = (float)coef[3], c5 = (float)coef[4];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 3 * (int64_t)F; i += (int64_t)gridDim.x * 256) {
        const float n = nrm[i];
        float g = (float)(c3 * sgn((double)n - real_n[i])) + c5 * pn_dn[i];
        if (G0) g += G0[i] - c4 * sgn(n_last[i] - n);
        dnorm[i] = g;
    }
}

__global__ __launch_bounds__(256) void vertex_bwd_kernel(int V, const float* __restrict__ pos,
                                                         const double* __restrict__ real,
                                                         const float* __restrict__ resid,
                                                         const int* __restrict__ vv_ptr,
                                                         const int* __restrict__ vv_idx,
                                                         const int* __restrict__ vf_ptr,
                                                         const int* __restrict__ vf_corner,
                                                         const float* __restrict__ pn_coef,
                                                         const float* __restrict__ nrm,
                                                         const double* __restrict__ coef,
                                                         float* __restrict__ dpos) {
    const double c1 = coef[0];
    const float c2 = (float)coef[1], c5 = (float)coef[4];
    for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
        const F3 p = ld3(pos, v);
        F3 g = {(float)(c1 * ((double)p.x - real[3 * (int64_t)v])), (float)(c1 * ((double)p.y - real[3 * (int64_t)v + 1])),
                (float)(c1 * ((double)p.z - real[3 * (int64_t)v + 2]))};
        if (c2 != 0.f) {
            F3 acc = ld3(resid, v);
            for (int k = vv_ptr[v]; k < vv_ptr[v + 1]; ++k) {
                const int j = vv_idx[k];
                const float dj = (float)(vv_ptr[j + 1] - vv_ptr[j]);
                const F3 rj = ld3(resid, j);
                acc = {acc.x - rj.x / dj, acc.y - rj.y / dj, acc.z - rj.z / dj};
            }
            g = g + c2 * acc;
        }
        if (c5 != 0.f) {
            F3 acc = {0.f, 0.f, 0.f};
            for (int k = vf_ptr[v]; k < vf_ptr[v + 1]; ++k) {
                const int fk = vf_corner[k];
                acc = acc + pn_coef[fk] * ld3(nrm, fk / 3);
            }
            g = g + c5 * acc;
        }
        st3(dpos, v, g);
    }
}

int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256), 256 * 8)); }

}  // namespace

extern "C" size_t ddmp_loss_partials_bytes(void) { return sizeof(double) * P_COUNT * kNB; }

extern "C" int ddmp_loss_vertex_fwd_part(int64_t V, const float* pos, const double* real_pos, const int32_t* vv_ptr,
                                         const int32_t* vv_idx, float* resid, double* partials,
                                         const uint8_t* own, ddmp_stream stream) {
    ARG_TRY(V > 0 && V < INT32_MAX && pos && real_pos && vv_ptr && vv_idx && resid && partials);
    hipLaunchKernelGGL(vertex_fwd_kernel, dim3(kNB), dim3(256), 0, (hipStream_t)stream, (int)V, pos, real_pos,
                       vv_ptr, vv_idx, resid, partials, own);
    LAUNCH_TRY();
    return DDMP_OK;
}
extern "C" int ddmp_loss_vertex_fwd(int64_t V, const float* pos, const double* real_pos, const int32_t* vv_ptr,
                                    const int32_t* vv_idx, float* resid, double* partials,
                                    ddmp_stream stream) {
    return ddmp_loss_vertex_fwd_part(V, pos, real_pos, vv_ptr, vv_idx, resid, partials, nullptr, stream);
}

extern "C" int ddmp_loss_face_fwd_part(int64_t F, const float* pos, const float* norm, const double* real_norm,
                                       const int32_t* faces, float* fc, float* fa, float* pn_coef, float* pn_dn,
                                       double* partials, const uint8_t* own, ddmp_stream stream) {
    ARG_TRY(F > 0 && F < INT32_MAX / 3 && pos && norm && real_norm && faces && fc && fa && pn_coef && pn_dn && partials);
    hipLaunchKernelGGL(face_fwd_kernel, dim3(kNB), dim3(256), 0, (hipStream_t)stream, (int)F, pos, norm, real_norm,
                       faces, fc, fa, pn_coef, pn_dn, partials, own);
    LAUNCH_TRY();
    return DDMP_OK;
}
extern "C" int ddmp_loss_face_fwd(int64_t F, const float* pos, const float* norm, const double* real_norm,
                                  const int32_t* faces, float* fc, float* fa, float* pn_coef, float* pn_dn,
                                  double* partials, ddmp_stream stream) {
    return ddmp_loss_face_fwd_part(F, pos, norm, real_norm, faces, fc, fa, pn_coef, pn_dn, partials, nullptr, stream);
}

// bnf_n: [(loop+1), F, 3] (slot 0 receives a copy of `norm`), bnf_A: [loop, F, 3], fcd: [F,3].
// Two halves so that a sharded caller can all-reduce the sigma_c partial sums in between:
//   ddmp_loss_bnf_sigma   copy of norm into slot 0, centroid distances fcd, partial sums of sigma_c (over own faces)
//   ddmp_loss_bnf_filter  `loop` filter passes with sigma_c = sum(partials) / (3 F_glob), then the L1 partial sums
extern "C" int ddmp_loss_bnf_sigma(int64_t F, const float* norm, const int32_t* f2f, const float* fc, float* fcd,
                                   float* bnf_n, double* partials, const uint8_t* own, ddmp_stream stream) {
    ARG_TRY(F > 0 && F < INT32_MAX / 3 && norm && f2f && fc && fcd && bnf_n && partials);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(bnf_n, norm, sizeof(float) * 3 * (size_t)F, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(bnf_sigma_kernel, dim3(kNB), dim3(256), 0, st, (int)F, fc, f2f, fcd, partials, own);
    LAUNCH_TRY();
    return DDMP_OK;
}
extern "C" int ddmp_loss_bnf_filter(int64_t F, int64_t F_glob, const int32_t* f2f, const float* fcd, const float* fa,
                                    int loop, float* bnf_n, float* bnf_A, double* partials, const uint8_t* own,
                                    ddmp_stream stream) {
    ARG_TRY(F > 0 && F < INT32_MAX / 3 && F_glob > 0 && F_glob < INT32_MAX / 3 && f2f && fcd && fa && loop >= 0 && bnf_n && partials);
    ARG_TRY(loop == 0 || bnf_A);
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(F);
    for (int t = 0; t < loop; ++t) {
        hipLaunchKernelGGL(bnf_iter_kernel, dim3(grid), dim3(256), 0, st, (int)F, bnf_n + (size_t)t * 3 * F, f2f, fcd,
                           fa, partials, bnf_A + (size_t)t * 3 * F, bnf_n + (size_t)(t + 1) * 3 * F, (int)F_glob);
        LAUNCH_TRY();
    }
    hipLaunchKernelGGL(bnf_diff_kernel, dim3(kNB), dim3(256), 0, st, (int)F, bnf_n + (size_t)loop * 3 * F, bnf_n,
                       partials, own);
    LAUNCH_TRY();
    return DDMP_OK;
}
extern "C" int ddmp_loss_bnf_fwd(int64_t F, const float* norm, const int32_t* f2f, const float* fc,
                                 const float* fa, int loop, float* fcd, float* bnf_n, float* bnf_A,
                                 double* partials, ddmp_stream stream) {
    ARG_TRY(F > 0 && F < INT32_MAX / 3 && norm && f2f && fc && fa && loop >= 0 && fcd && bnf_n && partials);
    int rc = ddmp_loss_bnf_sigma(F, norm, f2f, fc, fcd, bnf_n, partials, nullptr, stream);
    if (rc != DDMP_OK) return rc;
    return ddmp_loss_bnf_filter(F, F, f2f, fcd, fa, loop, bnf_n, bnf_A, partials, nullptr, stream);
}

extern "C" int ddmp_loss_finalize(const double* partials, int64_t V, int64_t F, const double* k5 /*host [5]*/,
                                  double gate4, double* lossbuf /*[12]*/, ddmp_stream stream) {
    ARG_TRY(partials && V > 0 && F > 0 && k5 && lossbuf);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, (int)V, (int)F,
                       k5[0], k5[1], k5[2], k5[3], k5[4], gate4, lossbuf);
    LAUNCH_TRY();
    return DDMP_OK;
}

// gradient of L4 w.r.t. the input normals, scaled by coef[3]; writes G0 [F,3] (the part that flows through
// the filter; the direct -sign term is added by ddmp_loss_face_bwd).  scratch: 2 x [F,3] floats.
extern "C" int ddmp_loss_bnf_bwd(int64_t F, const int32_t* f2f, const float* fa, const float* fcd, int loop,
                                 const float* bnf_n, const float* bnf_A, const double* partials,
                                 const double* coef, float* G0, float* scratch, ddmp_stream stream) {
    return ddmp_loss_bnf_bwd_part(F, F, f2f, fa, fcd, loop, bnf_n, bnf_A, partials, coef, G0, scratch, stream);
}
extern "C" int ddmp_loss_bnf_bwd_part(int64_t F, int64_t F_glob, const int32_t* f2f, const float* fa, const float* fcd,
                                      int loop, const float* bnf_n, const float* bnf_A, const double* partials,
                                      const double* coef, float* G0, float* scratch, ddmp_stream stream) {
    ARG_TRY(F > 0 && F < INT32_MAX / 3 && F_glob > 0 && f2f && fa && fcd && loop >= 0 && bnf_n && partials && coef && G0);
    ARG_TRY(loop == 0 || (bnf_A && scratch));
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(3 * F), gridf = grid_for(F);
    float* Ga = loop % 2 == 0 ? G0 : scratch;           // ping-pong so the last gather lands in G0
    float* Gb = loop % 2 == 0 ? scratch : G0;
    float* dA = scratch + 3 * (size_t)F;
    hipLaunchKernelGGL(bnf_bwd_init_kernel, dim3(grid), dim3(256), 0, st, (int)F, bnf_n + (size_t)loop * 3 * F, bnf_n,
                       coef + 3, Ga);
    LAUNCH_TRY();
    for (int t = loop - 1; t >= 0; --t) {
        hipLaunchKernelGGL(bnf_bwd_dA_kernel, dim3(gridf), dim3(256), 0, st, (int)F, Ga, bnf_A + (size_t)t * 3 * F, dA);
        LAUNCH_TRY();
        hipLaunchKernelGGL(bnf_bwd_gather_kernel, dim3(gridf), dim3(256), 0, st, (int)F, bnf_n + (size_t)t * 3 * F, dA,
                           f2f, fcd, fa, partials, Gb, (int)F_glob);
        LAUNCH_TRY();
        std::swap(Ga, Gb);
    }
    return DDMP_OK;
}

extern "C" int ddmp_loss_face_bwd(int64_t F, const float* norm, const double* real_norm, const float* pn_dn,
                                  const float* G0 /*nullable*/, const float* n_last /*with G0*/,
                                  const double* coef, float* dnorm, ddmp_stream stream) {
    ARG_TRY(F > 0 && F < INT32_MAX / 3 && norm && real_norm && pn_dn && coef && dnorm);
    ARG_TRY(!G0 || n_last);
    hipLaunchKernelGGL(face_bwd_kernel, dim3(grid_for(3 * F)), dim3(256), 0, (hipStream_t)stream, (int)F, norm,
                       real_norm, pn_dn, G0, n_last, coef, dnorm);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_loss_vertex_bwd(int64_t V, const float* pos, const double* real_pos, const float* resid,
                                    const int32_t* vv_ptr, const int32_t* vv_idx, const int32_t* vf_ptr,
                                    const int32_t* vf_corner, const float* pn_coef, const float* norm,
                                    const double* coef, float* dpos, ddmp_stream stream) {
    ARG_TRY(V > 0 && V < INT32_MAX && pos && real_pos && resid && vv_ptr && vv_idx && vf_ptr && vf_corner);
    ARG_TRY(pn_coef && norm && coef && dpos);
    hipLaunchKernelGGL(vertex_bwd_kernel, dim3(grid_for(V)), dim3(256), 0, (hipStream_t)stream, (int)V, pos, real_pos,
                       resid, vv_ptr, vv_idx, vf_ptr, vf_corner, pn_coef, norm, coef, dpos);
    LAUNCH_TRY();
    return DDMP_OK;
}
