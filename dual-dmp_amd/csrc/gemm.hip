// Dense per-node feature x weight contractions on the matrix cores, exact f32
// (v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate, 157.3 TF peak on MI355X; gfx950 has no TF32).
//
//   nt / nn :  Y[n, M] = f(A[n, K]) . op(B)  (+ bias)      tall-skinny: n = 0.5-8 M rows, K, M <= 512
//   tn      :  dW[M, K] = G[n, M]^T . f(Z[n, K])            reduction over the n rows, split over blocks
//
// Tiling for 64-wide wavefronts: a 256-thread workgroup (4 waves as 2x2) owns a 128 x (32*2*TN) output
// tile; each wave owns (2 x TN) MFMA tiles of 32x32 (16 accumulator VGPRs each).  Operands go
// global -> registers (fused BatchNorm+LeakyReLU prologue on A / Z) -> LDS (double buffered, one barrier
// per K tile) -> ds_read_b128 fragments.  The K order inside an 8-wide group is permuted so that one
// b128 read feeds four consecutive MFMAs (lane half h supplies k = 8*kk + 4*h + s for step s): the sum
// over k is the same set of products.  Row stride 36 floats keeps the b128 fragment reads
// bank-conflict free (36*r mod 64 = 4*(9r mod 16), distinct for the 16 rows of a lane group).
//
// blockIdx is XCD-aware: the column tiles of one row panel are adjacent on ONE XCD, so the A panel is
// fetched from HBM once and re-read from that XCD's L2; W (<= 1 MiB) lives in every L2.
#include "ddmp_common.h"
#include "finalize.h"
#include "gemm_tn_rm.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

using namespace ddmp;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128;      // rows per block tile
constexpr int kBK = 32;       // K per LDS tile
constexpr int kLd = kBK + 4;  // LDS row stride (floats)

// ------------------------------------------------------------------------------------------------
//  Y = f(A) . W^T (B_KMAJOR = false, W is [M,K])   or   Y = A . W (B_KMAJOR = true, W is [K_red, M_out])
//  generic naming inside: A[n, KD] (reduction dim KD), B gives Bs[m][k], Y[n, MD]
// ------------------------------------------------------------------------------------------------
template <int TN, bool B_KMAJOR, bool PRO>
__global__ __launch_bounds__(256) void gemm_rows_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
    float* __restrict__ Y, int64_t ldy, int n_rows, int KD, int MD,
    const float* __restrict__ bias, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float slope, int n_row_tiles, int n_col_tiles) {
    constexpr int BN = 64 * TN;
    __shared__ __attribute__((aligned(16))) float As[2][kBM * kLd];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * kLd];

    // XCD-aware tile id: consecutive ids on one XCD share the row panel
    const int per_xcd = (int)cdiv((int64_t)n_row_tiles, kXcd) * n_col_tiles;
    const int lin = (blockIdx.x & (kXcd - 1)) * per_xcd + (blockIdx.x >> 3);
    const int row_tile = lin / n_col_tiles, col_tile = lin % n_col_tiles;
    if (row_tile >= n_row_tiles) return;
    const int row0 = row_tile * kBM, col0 = col_tile * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // staging assignment: thread -> (k quad, row) ; 8 quads per row, 32 rows per pass
    const int kq = tid & 7, rr = tid >> 3;
    float4 ra[4], rb[(B_KMAJOR ? 4 : BN / 32)];
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);

    // unconditional clamped loads (all in flight together); prologue + zero-masking at LDS-store time
    auto load_tiles = [&](int k0) {
        const int kcol = min(k0 + kq * 4, KD - 4);
        if (PRO) {
            psc = *reinterpret_cast<const float4*>(pscale + kcol);
            psh = *reinterpret_cast<const float4*>(pshift + kcol);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = min(row0 + p * 32 + rr, n_rows - 1);
            ra[p] = *reinterpret_cast<const float4*>(A + (int64_t)r * lda + kcol);
        }
        if (!B_KMAJOR) {
#pragma unroll
            for (int p = 0; p < BN / 32; ++p) {
                const int m = min(col0 + p * 32 + rr, MD - 1);
                rb[p] = *reinterpret_cast<const float4*>(B + (int64_t)m * ldb + kcol);
            }
        } else {
            constexpr int QPR = BN / 4, KPP = 256 / QPR, NP = kBK / KPP;
            static_assert(NP <= 4, "rb too small");
            const int mq = tid % QPR, kr = tid / QPR;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int k = min(k0 + p * KPP + kr, KD - 1), m = min(col0 + mq * 4, ((MD + 3) / 4) * 4 - 4);
                rb[p] = *reinterpret_cast<const float4*>(B + (int64_t)k * ldb + m);
            }
        }
    };
    auto store_tiles = [&](int buf, int k0) {
        const bool kok = k0 + kq * 4 < KD;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float4 v = ra[p];
            if (PRO) v = f4_affine_lrelu(v, psc, psh, slope);
            if (!(kok && row0 + p * 32 + rr < n_rows)) v = zero4;
            *reinterpret_cast<float4*>(&As[buf][(p * 32 + rr) * kLd + kq * 4]) = v;
        }
        if (!B_KMAJOR) {
#pragma unroll
            for (int p = 0; p < BN / 32; ++p) {
                float4 v = rb[p];
                if (!(kok && col0 + p * 32 + rr < MD)) v = zero4;
                *reinterpret_cast<float4*>(&Bs[buf][(p * 32 + rr) * kLd + kq * 4]) = v;
            }
        } else {
            constexpr int QPR = BN / 4, KPP = 256 / QPR, NP = kBK / KPP;
            const int mq = tid % QPR, kr = tid / QPR;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int k = p * KPP + kr;
                const bool ok = (k0 + k < KD) && (col0 + mq * 4 < MD);
                Bs[buf][(mq * 4 + 0) * kLd + k] = ok ? rb[p].x : 0.f;
                Bs[buf][(mq * 4 + 1) * kLd + k] = ok ? rb[p].y : 0.f;
                Bs[buf][(mq * 4 + 2) * kLd + k] = ok ? rb[p].z : 0.f;
                Bs[buf][(mq * 4 + 3) * kLd + k] = ok ? rb[p].w : 0.f;
            }
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (KD + kBK - 1) / kBK;
    load_tiles(0);
    store_tiles(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * kBK);
        const float* as = &As[buf][(wm * 64 + l31) * kLd + lh * 4];
        const float* bs = &Bs[buf][(wn * 32 * TN + l31) * kLd + lh * 4];
#pragma unroll
        for (int kk = 0; kk < kBK / 8; ++kk) {
            float4 af[2], bf[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(as + i * 32 * kLd + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(bs + j * 32 * kLd + kk * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1, (kt + 1) * kBK);
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: reg r -> row (r&3) + 8*(r>>2) + 4*lh, col l31
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = col0 + wn * 32 * TN + j * 32 + l31;
        if (c >= MD) continue;
        const float bv = bias ? bias[c] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < n_rows) Y[(int64_t)row * ldy + c] = acc[i][j][r] + bv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
//  wgrad: partial[s][m][k] = sum_{rows of split s} G[row][m] * f(Z[row][k])
//  block tile (64*T) x (64*T) of dW; rows consumed 32 at a time; both operands are row(k)-major so
//  fragments are conflict-free ds_read_b32 (consecutive lanes -> consecutive floats).
// ------------------------------------------------------------------------------------------------
template <int T, bool PRO>
__global__ __launch_bounds__(256) void gemm_tn_kernel(
    const float* __restrict__ G, int64_t ldg, const float* __restrict__ Z, int64_t ldz,
    float* __restrict__ out, int64_t ld_out, int64_t split_stride, int n_rows, int M, int K,
    int rows_per_split, int n_tiles_m, int n_tiles_k, int n_splits,
    const float* __restrict__ pscale, const float* __restrict__ pshift, float slope) {
    constexpr int BT = 64 * T;                 // tile edge
    constexpr int QPR = BT / 4;                // float4 per tile row
    constexpr int RPP = 256 / QPR;             // rows per pass
    constexpr int NP = 32 / RPP;               // passes per 32-row tile
    __shared__ __attribute__((aligned(16))) float Gs[2][32 * BT];
    __shared__ __attribute__((aligned(16))) float Zs[2][32 * BT];

    // same-split tiles adjacent on one XCD (they share the G / Z row panels through L2)
    const int n_tiles = n_tiles_m * n_tiles_k;
    const int xcd = blockIdx.x & (kXcd - 1), local = blockIdx.x >> 3;
    const int tile = local % n_tiles;
    const int split = (local / n_tiles) * kXcd + xcd;
    if (split >= n_splits) return;
    const int tm0 = (tile / n_tiles_k) * BT, tk0 = (tile % n_tiles_k) * BT;
    const int r_begin = split * rows_per_split;
    const int r_end = min(n_rows, r_begin + rows_per_split);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int q = tid % QPR, pr = tid / QPR;

    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    const int zc = tk0 + q * 4, gc = tm0 + q * 4;
    if (PRO && zc < K) {
        sc = *reinterpret_cast<const float4*>(pscale + zc);
        sh = *reinterpret_cast<const float4*>(pshift + zc);
    }
    float4 rg[NP], rz[NP];
    const int gcl = min(gc, ((M + 3) / 4) * 4 - 4), zcl = min(zc, ((K + 3) / 4) * 4 - 4);
    auto load_tiles = [&](int r0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = min(r0 + p * RPP + pr, r_end - 1);
            rg[p] = *reinterpret_cast<const float4*>(G + (int64_t)r * ldg + gcl);
            rz[p] = *reinterpret_cast<const float4*>(Z + (int64_t)r * ldz + zcl);
        }
    };
    auto store_tiles = [&](int buf, int r0) {
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bool rok = r0 + p * RPP + pr < r_end;
            float4 z = rz[p];
            if (PRO) z = f4_affine_lrelu(z, sc, sh, slope);
            *reinterpret_cast<float4*>(&Gs[buf][(p * RPP + pr) * BT + q * 4]) = (rok && gc < M) ? rg[p] : zero4;
            *reinterpret_cast<float4*>(&Zs[buf][(p * RPP + pr) * BT + q * 4]) = (rok && zc < K) ? z : zero4;
        }
    };

    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (r_end - r_begin + 31) / 32;
    if (nt > 0) {
        load_tiles(r_begin);
        store_tiles(0, r_begin);
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tiles(r_begin + (t + 1) * 32);
        const float* gs = &Gs[buf][lh * BT + wm * 32 * T + l31];
        const float* zs = &Zs[buf][lh * BT + wn * 32 * T + l31];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            float a[T], b[T];
#pragma unroll
            for (int i = 0; i < T; ++i) a[i] = gs[ks * 2 * BT + i * 32];
#pragma unroll
            for (int j = 0; j < T; ++j) b[j] = zs[ks * 2 * BT + j * 32];
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) store_tiles(buf ^ 1, r_begin + (t + 1) * 32);
        __syncthreads();
    }

    float* o = out + (int64_t)split * split_stride;
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int k = tk0 + wn * 32 * T + j * 32 + l31;
        if (k >= K) continue;
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = tm0 + wm * 32 * T + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) o[(int64_t)m * ld_out + k] = acc[i][j][r];
            }
    }
}

// dW = sum over splits of the partial tiles.  64 float4 columns per workgroup, the splits dealt round-robin to its four waves
// (4 loads in flight each), float64 partial sums combined through LDS in wave order: deterministic.  (One thread per float4
// walking all n_splits partials -- 64 dependent-latency steps for a 512 x 512 panel on 4 waves per CU -- took 37-72 us.)
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ part, int64_t split_stride,
                                                            int n_splits, float* __restrict__ dW, int64_t lddw,
                                                            int M, int K) {
    __shared__ double sm[3][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;              // float4 index (M*K is a multiple of 4)
    const bool on = q * 4 < M * K;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (on) {
        int i = wave;
        for (; i + 12 < n_splits; i += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(part + (int64_t)(i + 4 * u) * split_stride + 4 * q);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s0 += v[u].x; s1 += v[u].y; s2 += v[u].z; s3 += v[u].w; }
        }
        for (; i < n_splits; i += 4) {
            const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)i * split_stride + 4 * q);
            s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        }
    }
    if (wave > 0) {
        sm[wave - 1][lane][0] = s0; sm[wave - 1][lane][1] = s1; sm[wave - 1][lane][2] = s2; sm[wave - 1][lane][3] = s3;
    }
    __syncthreads();
    if (wave == 0 && on) {
#pragma unroll
        for (int w = 0; w < 3; ++w) { s0 += sm[w][lane][0]; s1 += sm[w][lane][1]; s2 += sm[w][lane][2]; s3 += sm[w][lane][3]; }
        const int idx = 4 * q, m = idx / K, k = idx % K;    // K % 4 == 0: the four elements share a row
        float* o = dW + (int64_t)m * lddw + k;
        o[0] = (float)s0; o[1] = (float)s1; o[2] = (float)s2; o[3] = (float)s3;
    }
}

struct TnPlan {
    int T, n_tiles_m, n_tiles_k, n_splits, rows_per_split;      // T = 4: 512-thread panels of tm x tk (256 x 256; f16x3 route
    int tm, tk;                                                  // of round 4 also 256 x 128 | 128 x 256: gemm_tn_rm.hip)
};

bool tn_panel_enabled();
int64_t tn_panel_min_rows();

// narrow_panels: the caller takes the f16x3 route of gemm_tn_rm.hip, which also has 256 x 128 and 128 x 256 panels (the
// 128 <-> 256 layers)
TnPlan tn_plan(int64_t n_rows, int M, int K, bool narrow_panels = false) {
    TnPlan p;
    p.T = (M >= 128 && K >= 128) ? 2 : 1;
    p.tm = p.tk = 0;
    const bool wide = M >= 256 && K >= 256;
    const bool narrow = narrow_panels && ((M >= 256 && K == 128) || (M == 128 && K >= 256));
    if ((wide || narrow) && n_rows >= tn_panel_min_rows() && tn_panel_enabled()) {
        p.T = 4;
        p.tm = M >= 256 ? 256 : 128;
        p.tk = K >= 256 ? 256 : 128;
        p.n_tiles_m = (int)cdiv(M, p.tm);
        p.n_tiles_k = (int)cdiv(K, p.tk);
        const int tiles = p.n_tiles_m * p.n_tiles_k;
        // one 512-thread workgroup per CU; at least 256 rows (16 stages) per split; multiple of 8 splits (XCD mapping).
        // Round 6: every split writes a tm x tk float32 partial panel that reduce_splits_kernel reads back -- a fixed cost per launch
        // (64 splits x 4 panels x 256 KB = 64 MB at 512 x 512) that does not shrink with the mesh.  Below 400k rows THREE QUARTERS
        // of the splits (48 instead of 64): the step, interleaved A/B, 500k faces 22.75 against 22.94 ms, 250k 12.34 / 12.58, 125k
        // 7.14 / 7.37 (half: 22.86 / 12.23 / 7.21; 3/8 and twice: worse everywhere) -- the idle quarter of the CUs is not idle, the
        // other net's kernels run there from the second stream.  At 1M faces the same rule returned 0.16 ms of the step (43.43 /
        // 43.59) but cost the wgrad family 0.9 ms when it runs ALONE (serialised profile: 10.99 against 10.1 ms): not taken there.
        const int64_t cus = n_rows >= 400000 ? kCu : (3 * kCu) / 4;
        int64_t s = std::min<int64_t>(std::max<int64_t>(1, cus / tiles), std::max<int64_t>(1, n_rows / 256));
        s = std::max<int64_t>(kXcd, (s / kXcd) * kXcd);
        int64_t rps = cdiv(cdiv(n_rows, s), 32) * 32;
        p.rows_per_split = (int)rps;
        p.n_splits = (int)cdiv(n_rows, rps);
        return p;
    }
    const int bt = 64 * p.T;
    p.n_tiles_m = (int)cdiv(M, bt);
    p.n_tiles_k = (int)cdiv(K, bt);
    const int tiles = p.n_tiles_m * p.n_tiles_k;
    // aim at ~3 workgroups per CU (2 below 30k rows, where EVERY wgrad runs here and the partial panels are the larger part of
    // the work: 13k faces 2.62 against 2.66 ms per iteration on one stream in three interleaved rounds, 26k faces 3.56 / 3.62;
    // profiles/r06_wgrad_splits_ab.txt); at least 128 rows per split (small meshes: a split of 512 rows is 32 dependent stages,
    // 40-50 us per wgrad at 13k rows); multiple of 8 splits (XCD mapping)
    int64_t want = std::max<int64_t>(1, ((n_rows < 30000 ? 2 : 3) * kCu) / tiles);
    int64_t max_by_rows = std::max<int64_t>(1, n_rows / 128);
    int64_t s = std::min(want, max_by_rows);
    s = std::max<int64_t>(kXcd, (s / kXcd) * kXcd);
    int64_t rps = cdiv(n_rows, s);
    rps = cdiv(rps, 32) * 32;
    p.rows_per_split = (int)rps;
    p.n_splits = (int)cdiv(n_rows, rps);
    return p;
}

#include "gemm_bf16x.inc"
#include "gemm_ws.inc"
#include "gemm_panel.inc"
#include "gemm_rr.inc"

}  // namespace

namespace {
int64_t tn_panel_min_rows() {
    static const int64_t v = [] {
        const char* e = getenv("DDMP_TN_PANEL_MIN_ROWS");
        return (e && atoll(e) > 0) ? (int64_t)atoll(e) : (int64_t)30000;
    }();
    return v;
}
bool tn_panel_enabled() {                                       // DDMP_GEMM_PANEL=0 / DDMP_GEMM_MODE=0: tiled kernels
    const char* v = getenv("DDMP_GEMM_PANEL");
    return !(v && atoi(v) == 0) && ddmp_get_gemm_mode() != 0;
}
}  // namespace

static int device_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = ddmp::kCu;
    }
    return n;
}

static bool ws_enabled() { return true; }

// ---------------------------------------------------------------- weights prepared once per iteration (round 3)
// Every GEMM call used to split its weight matrix into 16-bit planes itself: absolute maximum (memset + kernel) + split
// kernel, ~90 launches of 4-10 us per training iteration.  ddmp_gemm_prepare_weights does all matrices of a net in TWO
// launches into caller-owned plane buffers; a GEMM call whose workspace IS such a buffer and that was announced with
// ddmp_gemm_next_prepared() skips its own split -- if and only if the recorded layout is exactly the one its route wants
// (same matrix, shape, orientation, plane format); otherwise it splits as before.  Same device code, same values.
enum { kWPanelF16 = 1, kWPanelB16 = 2, kWTiled = 3, kWPlain = 4 };
struct WPrepDesc {
    const float* W;
    void* planes;
    int64_t ldw;
    int kind, nterm, MD, KD, transpose, P;                       // P: MP (panel kinds) | BN (tiled) | 0
};
constexpr int kWPrepMax = 24;
struct WPrepBatch {
    int n;
    WPrepDesc d[kWPrepMax];
};
static inline bool wprep_same(const WPrepDesc& a, const WPrepDesc& b) {
    return a.W == b.W && a.planes == b.planes && a.ldw == b.ldw && a.kind == b.kind && a.nterm == b.nterm && a.MD == b.MD &&
           a.KD == b.KD && a.transpose == b.transpose && a.P == b.P;
}
static thread_local std::vector<WPrepDesc> g_w_registry;         // what the last ddmp_gemm_prepare_weights calls wrote
static thread_local bool g_w_next_prepared = false, g_w_call_prepared = false;
// at a split site: does `planes` already hold exactly this layout (and did the caller say so)?
static bool w_prepared(const float* W, int64_t ldw, void* planes, int kind, int nterm, int MD, int KD, int transpose, int P) {
    if (!g_w_call_prepared) return false;
    const WPrepDesc want{W, planes, ldw, kind, nterm, MD, KD, transpose, P};
    for (const WPrepDesc& e : g_w_registry)
        if (wprep_same(e, want)) return true;
    return false;
}

// per f16 matrix: 8 partial absolute maxima (no atomics, no zeroing)
__global__ __launch_bounds__(256) void wprep_max_kernel(WPrepBatch b, float* __restrict__ scratch) {
    const WPrepDesc d = b.d[blockIdx.y];
    if (d.kind != kWPanelF16) return;
    const int rows = d.transpose ? d.KD : d.MD, cols = d.transpose ? d.MD : d.KD;     // W as stored: [rows][cols]
    float m = 0.f;
    for (int r = blockIdx.x; r < rows; r += gridDim.x)
        for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, fabsf(d.W[(int64_t)r * d.ldw + c]));
    m = f16s_wave_max(m);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        if (!(m <= 3.0e38f)) m = 3.0e38f;
        scratch[blockIdx.y * 8 + blockIdx.x] = m;
    }
}
__global__ __launch_bounds__(256) void wprep_split_kernel(WPrepBatch b, const float* __restrict__ scratch) {
    const WPrepDesc d = b.d[blockIdx.y];
    const int bid = blockIdx.x, nblk = gridDim.x;
    if (d.kind == kWPanelF16) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, scratch[blockIdx.y * 8 + j]);
        float* wscale = (float*)((char*)d.planes + (size_t)2 * d.P * d.KD * 2);      // behind the planes (launch_panel)
        if (bid == 0 && threadIdx.x == 0) *wscale = m;
        split_w_panel_body<2, _Float16>(d.W, d.ldw, d.MD, d.KD, d.transpose, d.P, (_Float16*)d.planes,
                                        f16s_scale(m, kF16TargetExact), bid, nblk);
    } else if (d.kind == kWPanelB16) {
        if (d.nterm == 3) split_w_panel_body<3, __bf16>(d.W, d.ldw, d.MD, d.KD, d.transpose, d.P, (__bf16*)d.planes, 1.f, bid, nblk);
        else split_w_panel_body<2, __bf16>(d.W, d.ldw, d.MD, d.KD, d.transpose, d.P, (__bf16*)d.planes, 1.f, bid, nblk);
    } else if (d.kind == kWTiled) {
        if (d.nterm == 3) split_w_tiled_body<3>(d.W, d.ldw, d.MD, d.KD, d.transpose, d.P, (__bf16*)d.planes, bid, nblk);
        else split_w_tiled_body<2>(d.W, d.ldw, d.MD, d.KD, d.transpose, d.P, (__bf16*)d.planes, bid, nblk);
    } else if (d.kind == kWPlain) {
        if (d.nterm == 3) split_w_body<3>(d.W, d.ldw, d.MD, d.KD, d.transpose, (__bf16*)d.planes, bid, nblk);
        else split_w_body<2>(d.W, d.ldw, d.MD, d.KD, d.transpose, (__bf16*)d.planes, bid, nblk);
    }
}

// persistent wave-specialised kernel: pre-split W planes [MD][KD], reduction length KD % 32 == 0, KD >= 96
template <bool PRO>
static void launch_ws(int mode, const float* A, int64_t lda, const float* W, int64_t ldw, int transpose, void* planes,
                      float* Y, int64_t ldy, int n_rows, int KD, int MD, const float* bias, const float* ps,
                      const float* psh, float slope, hipStream_t st) {
    const int n_row_tiles = (int)ddmp::cdiv(n_rows, kBM);
    const int TN = MD > 64 ? 2 : 1;
    {
        const int BN = 64 * TN;
        const int64_t total = ddmp::cdiv(MD, BN) * BN * (int64_t)KD;
        const int sgrid = (int)std::min<int64_t>(ddmp::cdiv(total, 256), 1024);
        if (w_prepared(W, ldw, planes, kWTiled, mode == 6 ? 3 : 2, MD, KD, transpose, BN)) {
        } else if (mode == 6)
            hipLaunchKernelGGL((split_w_tiled_kernel<3>), dim3(sgrid), dim3(256), 0, st, W, ldw, MD, KD, transpose, BN, (__bf16*)planes);
        else
            hipLaunchKernelGGL((split_w_tiled_kernel<2>), dim3(sgrid), dim3(256), 0, st, W, ldw, MD, KD, transpose, BN, (__bf16*)planes);
    }
    const int n_col_tiles = (int)ddmp::cdiv(MD, 64 * TN);
    const int total = (int)ddmp::cdiv(n_row_tiles, ddmp::kXcd) * ddmp::kXcd * n_col_tiles;
    const int cus = device_cus() / ddmp::kXcd * ddmp::kXcd;
    dim3 grid((unsigned)std::min(total, cus)), block(512);
    const __bf16* Bp = (const __bf16*)planes;
#define DDMP_WS(TN_, NT_)                                                                                      \
    hipLaunchKernelGGL((gemm_rows_ws_kernel<TN_, NT_, PRO>), grid, block, 0, st, A, lda, Bp, Y, ldy, n_rows, KD, \
                       MD, bias, ps, psh, slope, n_row_tiles, n_col_tiles)
    if (TN == 2) {
        if (mode == 6) DDMP_WS(2, 3); else DDMP_WS(2, 2);
    } else {
        if (mode == 6) DDMP_WS(1, 3); else DDMP_WS(1, 2);
    }
#undef DDMP_WS
}
// DDMP_GEMM_PANEL=0 disables the row-panel kernel (A/B comparisons)
static bool panel_enabled() {
    static int e = -1;
    if (e < 0) {
        const char* v = getenv("DDMP_GEMM_PANEL");
        e = (v && atoi(v) == 0) ? 0 : 1;
    }
    return e == 1;
}
// Row panels leave CUs idle on small inputs (one 512-thread workgroup per 128/256 rows): measured better from ~25k
// rows (50k-face mesh 4.74 vs 5.02 ms/iteration, 13k-face mesh 4.08 vs 3.03); the wgrad panels from ~30k rows
// (62k rows: 512x512 243 vs 291 us, 256x256 90 vs 87 us; row splits as short as 256 rows keep every CU busy).
static int64_t env_rows(const char* name, int64_t dflt) {
    const char* v = getenv(name);
    return (v && atoll(v) > 0) ? atoll(v) : dflt;
}
// (DDMP_PANEL_MIN_ROWS / DDMP_TN_PANEL_MIN_ROWS override the thresholds for A/B runs)
static const int64_t kPanelMinRows = env_rows("DDMP_PANEL_MIN_ROWS", 20000);
static const int64_t kTnPanelMinRows = env_rows("DDMP_TN_PANEL_MIN_ROWS", 30000);
static int gemm_f16();
static bool rr_enabled();
// The wide f16x3 outputs go to the row-register kernel (gemm_rr.inc: 128-row x 256-column work items, two workgroups per CU,
// half the MFMA work of bf16x6): worth it from far fewer rows than the 512-thread row panels (DDMP_RR_MIN_ROWS)
// (13,068-face mesh, replayed graph + two streams: 2.29 ms per iteration with 20000, 2.13 with 10000 -- NormalNet's 13k
// rows on the kernel --, 2.21 with 3000: PosNet's 6.5k rows = 52 work items are too few for it; 49k faces: 4.04 / 4.00 / 3.92)
static const int64_t kRRMinRows = env_rows("DDMP_RR_MIN_ROWS", 10000);
static inline int64_t panel_min_rows(int KD, int MD) {
    return (MD > 128 && KD >= 64 && KD <= 512 && gemm_f16() && rr_enabled()) ? std::min(kRRMinRows, kPanelMinRows) : kPanelMinRows;
}
static inline bool panel_ok(int KD, int MD, const float* Y, int64_t ldy, const void* ws, size_t ws_bytes,
                            int64_t n_rows) {
    return panel_enabled() && n_rows >= panel_min_rows(KD, MD) && KD % 32 == 0 && KD >= 32 && (KD >= 64 || MD <= 128) &&
           MD % 4 == 0 && MD >= 16 && MD <= 512 && ws &&
           (reinterpret_cast<uintptr_t>(ws) & 15) == 0 && ws_bytes >= ddmp_gemm_rows_workspace_bytes(KD, MD) &&
           ldy >= MD && Y;
}
// DDMP_GEMM_RR=0 keeps the row-panel kernel for the wide f16x3 GEMMs (A/B comparisons); DDMP_RR_MIN_ROWS: threshold
static bool rr_enabled() {
    static int e = -1;
    if (e < 0) {
        const char* v = getenv("DDMP_GEMM_RR");
        e = (v && atoi(v) == 0) ? 0 : 1;
    }
    return e == 1;
}
// (given the f16x3 row-panel route: wide output, f16 mode) does the call take the row-register kernel?
static inline bool rr_route_ok(int64_t n_rows, int KD, int64_t lda, int64_t lda2) {
    return rr_enabled() && KD >= 64 && KD <= 512 && n_rows >= kRRMinRows &&
           n_rows * lda * 4 < ((int64_t)1 << 32) && n_rows * lda2 * 4 < ((int64_t)1 << 32);      // (32-bit lane offsets)
}
// f16 split mode (gemm_f16s.inc): 0 | 13; operand scale slots of the NEXT ddmp_gemm_* call on this host thread
struct ScaleCtx {
    float* a = nullptr;
    float* b = nullptr;
    int prime = 0;
};
static thread_local ScaleCtx g_scale_ctx;
namespace ddmp {
int gemm_next_pending() { return ((g_scale_ctx.a || g_scale_ctx.b || g_scale_ctx.prime) ? 2 : 0) | (g_w_next_prepared ? 4 : 0); }
void gemm_next_cancel() {
    g_scale_ctx = ScaleCtx();
    g_w_next_prepared = false;
}
}  // namespace ddmp
static ScaleCtx take_scale_ctx() {                               // (top of every GEMM entry point)
    ScaleCtx c = g_scale_ctx;
    g_scale_ctx = ScaleCtx();
    g_w_call_prepared = g_w_next_prepared;
    g_w_next_prepared = false;
    return c;
}
// slot[0] = max |f(A)|, exactly (pre-pass)
template <int PM>
static void f16s_measure(const float* A, int64_t lda, const float* A2, int64_t lda2, int64_t n_rows, int C,
                         const float* pa, const float* pb, const float* pk1, const float* pk0, float slope, float* slot,
                         hipStream_t st) {
    (void)hipMemsetAsync(slot, 0, 16, st);
    const int64_t total = n_rows * (C / 4);
    const int grid = (int)std::min<int64_t>(cdiv(total, 256 * 8), 8 * device_cus());
    hipLaunchKernelGGL((f16s_absmax_kernel<PM>), dim3((unsigned)std::max(grid, 1)), dim3(256), 0, st, A, lda, A2, lda2,
                       n_rows, C, pa, pb, pk1, pk0, slope, slot);
}

template <int PM>
static void launch_panel(int mode, const float* A, int64_t lda, const float* A2, int64_t lda2, const float* W,
                         int64_t ldw, int transpose, void* planes, float* Y, int64_t ldy, int n_rows, int KD, int MD,
                         const float* bias, const float* ps, const float* psh, const float* pc1, const float* pc0,
                         float slope, hipStream_t st, double* stats = nullptr, double* sums = nullptr,
                         ScaleCtx ctx = ScaleCtx(), const float* red_yp = nullptr, int64_t red_ldyp = 0,
                         const float* const* red_bn4 = nullptr /* scale, shift, mean, rstd: stats = the backward reductions */) {
    // wide outputs: 128 x 512 | 256 x 256 blocks (64 x 128 per wave); narrow outputs (MD <= 128): 512-row blocks, 64 x 32 NJ
    const int WC = MD > 256 ? 4 : MD > 128 ? 2 : 1, WR = 8 / WC;
    const int NJ = MD > 128 ? 4 : MD > 64 ? 4 : MD > 32 ? 2 : 1;
    const int MP = 32 * NJ * WC, BMR = 64 * WR;
    const int n_row_tiles = (int)ddmp::cdiv(n_rows, BMR);
    const int64_t total = (int64_t)MP * KD;
    const int sgrid = (int)std::min<int64_t>(ddmp::cdiv(total, 256), 1024);
    dim3 grid((unsigned)std::min(n_row_tiles, device_cus())), block(512);
    const int f16 = (mode == 6 && WC > 1 && lda % 4 == 0 && (PM != 2 || lda2 % 4 == 0)) ? gemm_f16() : 0;
    if (f16) {
        // two f16 planes of W * wscale; the scale and (without caller slots) the A operand's slot sit behind them
        char* tail = (char*)planes + (size_t)2 * MP * KD * 2;
        float* wscale = (float*)tail;
        float* slot = ctx.a ? ctx.a : (float*)(tail + 16);
        const bool prime = !ctx.a || ctx.prime;
        const bool w_ready = w_prepared(W, ldw, planes, kWPanelF16, 2, MD, KD, transpose, MP);
        if (!w_ready) {
            (void)hipMemsetAsync(wscale, 0, 4, st);               // = max |W| (the kernels derive the power of two)
            hipLaunchKernelGGL(f16s_wmax_kernel, dim3((unsigned)std::min(transpose ? KD : MD, 128)), dim3(256), 0, st, W, ldw,
                               transpose ? KD : MD, transpose ? MD : KD, wscale);
        }
        // PM = 2 with two column halves (512 <- 512 dgrad): the halves are separate, freely drifting workgroups and BOTH stream
        // (dZ, Y) -- PMC: the family read 27.3 GB per step for 18.4 GB algorithmic, the second half's rows mostly missing the
        // XCD's L2 -- while the row-panel kernel reads them once at the same speed (1499 vs 1490-1511 us): that shape keeps it.
        const bool rr = rr_route_ok(n_rows, KD, lda, PM == 2 ? lda2 : 0) && !(PM == 2 && MD > kRRCols);
        if (!w_ready)
            hipLaunchKernelGGL((split_w_panel_kernel<2, _Float16>), dim3(sgrid), dim3(256), 0, st, W, ldw, MD, KD, transpose, MP,
                               (_Float16*)planes, (const float*)wscale);
        if (prime) f16s_measure<PM>(A, lda, A2, lda2, n_rows, KD, ps, psh, pc1, pc0, slope, slot, st);
        const int target = prime ? kF16TargetExact : kF16TargetStale;
        const void* Bh = planes;
        if (rr) {
            // row-register kernel (gemm_rr.inc): 128-row tiles x 256-column halves, two workgroups per CU
            const int n_halves = MP / kRRCols;
            const int tiles = (int)ddmp::cdiv(n_rows, kRRRows);
            // (a multiple of 8 that covers all tiles in ONE round when they fit: surplus blocks return at once)
            // (round 6: 3/4 or 1/2 of these slots measured +1.1 / +1.9 ms per step at 1M faces, nothing at 125k)
            const int slots = std::max(8, std::min((tiles + 7) / 8 * 8, 2 * device_cus() / n_halves / 8 * 8));
            dim3 rgrid((unsigned)(slots * n_halves)), rblock(256);
            for (int heal = 0; heal <= (prime ? 0 : 1); ++heal) {
                if (PM == 0 && stats && red_yp)
                    hipLaunchKernelGGL((gemm_rr_kernel<0, 2, 3>), rgrid, rblock, 0, st, A, lda, A2, lda2, (const _Float16*)Bh, MP,
                                       Y, ldy, n_rows, KD, MD, bias, ps, psh, pc1, pc0, slope, tiles, stats, slot,
                                       (const float*)wscale, target, heal, red_yp, red_ldyp, red_bn4[0], red_bn4[1], red_bn4[2],
                                       red_bn4[3]);
                else if (PM != 2 && stats)
                    hipLaunchKernelGGL((gemm_rr_kernel<PM == 2 ? 0 : PM, 1, 3>), rgrid, rblock, 0, st, A, lda, A2, lda2,
                                       (const _Float16*)Bh, MP, Y, ldy, n_rows, KD, MD, bias, ps, psh, pc1, pc0, slope, tiles,
                                       stats, slot, (const float*)wscale, target, heal);
                else
                    hipLaunchKernelGGL((gemm_rr_kernel<PM, 0, PM == 2 ? 2 : 3>), rgrid, rblock, 0, st, A, lda, A2, lda2, (const _Float16*)Bh, MP,
                                       Y, ldy, n_rows, KD, MD, bias, ps, psh, pc1, pc0, slope, tiles, stats, slot,
                                       (const float*)wscale, target, heal);
            }
            if (stats && sums) {
                const size_t pbytes = ((size_t)tiles * 2 * 2 * MP * sizeof(double) + 255) / 256 * 256;
                fpartials_reduce(stats, tiles * 2, MP, MD, (double*)((char*)stats + pbytes), sums, st);
            }
            return;
        }
#define DDMP_PANEL_H(WR_, WC_, AR_, HEAL_)                                                                        \
    hipLaunchKernelGGL((gemm_panel_kernel<WR_, WC_, AR_, PM, 4>), grid, block, 0, st, A, lda, A2, lda2, Bh, Y, ldy,   \
                       n_rows, KD, MD, bias, ps, psh, pc1, pc0, slope, n_row_tiles, stats, slot, (const float*)wscale, \
                       target, HEAL_)
        // stale scale in use: a second launch redoes the product if (and only if) an operand outgrew it (gemm_f16s.inc)
        for (int heal = 0; heal <= (prime ? 0 : 1); ++heal) {
            if (WC == 4) {
                DDMP_PANEL_H(2, 4, 13, heal);
            } else {
                DDMP_PANEL_H(4, 2, 13, heal);
            }
        }
#undef DDMP_PANEL_H
        if (stats && sums) {
            const size_t pbytes = ((size_t)n_row_tiles * WR * 2 * MP * sizeof(double) + 255) / 256 * 256;
            fpartials_reduce(stats, n_row_tiles * WR, MP, MD, (double*)((char*)stats + pbytes), sums, st);
        }
        return;
    }
    if (w_prepared(W, ldw, planes, kWPanelB16, mode == 6 ? 3 : 2, MD, KD, transpose, MP)) {
    } else if (mode == 6)
        hipLaunchKernelGGL((split_w_panel_kernel<3, __bf16>), dim3(sgrid), dim3(256), 0, st, W, ldw, MD, KD, transpose, MP, (__bf16*)planes, (const float*)nullptr);
    else
        hipLaunchKernelGGL((split_w_panel_kernel<2, __bf16>), dim3(sgrid), dim3(256), 0, st, W, ldw, MD, KD, transpose, MP, (__bf16*)planes, (const float*)nullptr);
    const void* Bp = planes;
#define DDMP_PANEL(WR_, WC_, NT_, NJ_)                                                                            \
    hipLaunchKernelGGL((gemm_panel_kernel<WR_, WC_, NT_, PM, NJ_>), grid, block, 0, st, A, lda, A2, lda2, Bp, Y, ldy, \
                       n_rows, KD, MD, bias, ps, psh, pc1, pc0, slope, n_row_tiles, stats, (float*)nullptr,            \
                       (const float*)nullptr, 0, 0)
    if (WC == 1 && PM == 0 && stats && red_yp && (NJ == 4 || NJ == 2)) {
        // narrow transform-first dgrads (outputs of 128 | 64 columns) with the next BatchNorm-backward reductions
#define DDMP_PANEL_RS(NT_, NJ_)                                                                                     \
    hipLaunchKernelGGL((gemm_panel_kernel<8, 1, NT_, 0, NJ_, true>), grid, block, 0, st, A, lda, A2, lda2, Bp, Y, ldy, \
                       n_rows, KD, MD, bias, ps, psh, pc1, pc0, slope, n_row_tiles, stats, (float*)nullptr,            \
                       (const float*)nullptr, 0, 0, red_yp, red_ldyp, red_bn4[0], red_bn4[1], red_bn4[2], red_bn4[3])
        if constexpr (PM == 0) {
            if (NJ == 4) {
                if (mode == 6) DDMP_PANEL_RS(3, 4); else DDMP_PANEL_RS(2, 4);
            } else {
                if (mode == 6) DDMP_PANEL_RS(3, 2); else DDMP_PANEL_RS(2, 2);
            }
        }
#undef DDMP_PANEL_RS
    } else if (WC == 4) {
        if (mode == 6) DDMP_PANEL(2, 4, 3, 4); else DDMP_PANEL(2, 4, 2, 4);
    } else if (WC == 2) {
        if (mode == 6) DDMP_PANEL(4, 2, 3, 4); else DDMP_PANEL(4, 2, 2, 4);
    } else {
        if (NJ == 4) {
            if (mode == 6) DDMP_PANEL(8, 1, 3, 4); else DDMP_PANEL(8, 1, 2, 4);
        } else if (NJ == 2) {
            if (mode == 6) DDMP_PANEL(8, 1, 3, 2); else DDMP_PANEL(8, 1, 2, 2);
        } else {
            if (mode == 6) DDMP_PANEL(8, 1, 3, 1); else DDMP_PANEL(8, 1, 2, 1);
        }
    }
#undef DDMP_PANEL
    if (stats && sums) {
        const size_t pbytes = ((size_t)n_row_tiles * WR * 2 * MP * sizeof(double) + 255) / 256 * 256;
        fpartials_reduce(stats, n_row_tiles * WR, MP, MD, (double*)((char*)stats + pbytes), sums, st);
    }
}

static inline bool ws_ok(int KD, int MD, const float* Y, int64_t ldy, const void* ws, size_t ws_bytes) {
    return ws_enabled() && KD % 32 == 0 && KD >= 96 && MD % 4 == 0 && MD <= 1024 && ldy % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && ws && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 &&
           ws_bytes >= ddmp_gemm_rows_workspace_bytes(KD, MD);
}

// GEMM arithmetic: 13 (default) = f16x3 split MFMA in the row-panel kernels (gemm_f16s.inc), bf16x6 everywhere else;
// 6 = bf16x6 split MFMA everywhere; both f32-class accuracy.  3 = bf16x3 (~2^-16), 0 = f32-input MFMA
static int g_gemm_mode = -1, g_gemm_f16 = 0;
static int gemm_mode() {
    if (g_gemm_mode < 0) {
        const char* e = getenv("DDMP_GEMM_MODE");
        int m = e ? atoi(e) : 13;
        g_gemm_f16 = m == 13 ? m : 0;
        g_gemm_mode = (m == 0 || m == 3 || m == 6) ? m : 6;
    }
    return g_gemm_mode;
}
static int gemm_f16() {
    (void)gemm_mode();
    return g_gemm_f16;
}
extern "C" int ddmp_set_gemm_mode(int mode) {
    if (mode != 0 && mode != 3 && mode != 6 && mode != 13) return DDMP_EINVAL;
    g_gemm_f16 = mode > 10 ? mode : 0;
    g_gemm_mode = mode > 10 ? 6 : mode;
    return DDMP_OK;
}
extern "C" int ddmp_get_gemm_mode(void) { return gemm_f16() ? gemm_f16() : gemm_mode(); }

extern "C" int ddmp_gemm_next_scales(float* slot_a, float* slot_b, int prime) {
    g_scale_ctx.a = slot_a;
    g_scale_ctx.b = slot_b;
    g_scale_ctx.prime = prime;
    return DDMP_OK;
}
extern "C" int ddmp_gemm_scales_roll(float* slots, int n_slots, ddmp_stream stream) {
    ARG_TRY(slots && n_slots > 0);
    hipLaunchKernelGGL(f16s_roll_kernel, dim3((unsigned)cdiv(n_slots, 64)), dim3(64), 0, (hipStream_t)stream, slots, n_slots);
    LAUNCH_TRY();
    return DDMP_OK;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" size_t ddmp_gemm_rows_workspace_bytes(int K, int M) {
    if (K <= 0 || M <= 0) return 0;
    const size_t kp = ((size_t)K + 127) / 128 * 128, mp = ((size_t)M + 127) / 128 * 128;
    return 3 * kp * mp * 2;                                     // three bf16 planes of W, tile-padded
}

// bf16 modes with a workspace: W is split ONCE here instead of by every row-tile workgroup
static bool presplit_w(const float* W, int64_t ldw, int M, int K, int transpose, void* ws, size_t ws_bytes,
                       hipStream_t st) {
    const int mode = gemm_mode();
    const int RK = transpose ? M : K;                           // reduction length of the consumer
    if (mode == 0 || !ws || ws_bytes < ddmp_gemm_rows_workspace_bytes(K, M) || RK % 8 != 0 ||
        (reinterpret_cast<uintptr_t>(ws) & 15))
        return false;
    const int grid = (int)std::min<int64_t>(cdiv((int64_t)M * K, 256), 1024);
    if (w_prepared(W, ldw, ws, kWPlain, mode == 6 ? 3 : 2, M, K, transpose, 0)) return true;
    if (mode == 6)
        hipLaunchKernelGGL((split_w_kernel<3>), dim3(grid), dim3(256), 0, st, W, ldw, M, K, transpose, (__bf16*)ws);
    else
        hipLaunchKernelGGL((split_w_kernel<2>), dim3(grid), dim3(256), 0, st, W, ldw, M, K, transpose, (__bf16*)ws);
    return hipGetLastError() == hipSuccess;
}

// The plane layout the entry points below will want for W [M,K] in the forward (form 0: Y[n,M] = f(A[n,K]) . W^T) or dgrad
// (form 1: Y[n,K] = A[n,M] . W) product over n_rows rows -- the same tests, in the same order, as those entry points
// (which stay the ground truth: a layout they do not want is simply ignored and re-made).
static WPrepDesc plan_w(int form, int64_t n_rows, const float* W, int64_t ldw, int M, int K, int has_pro, void* planes,
                        size_t planes_bytes) {
    WPrepDesc d{W, planes, ldw, 0, 0, 0, 0, 0, 0};
    const int mode = gemm_mode();
    if (mode == 0 || !planes || (reinterpret_cast<uintptr_t>(planes) & 15) || planes_bytes < ddmp_gemm_rows_workspace_bytes(K, M))
        return d;
    const int KD = form ? M : K, MD = form ? K : M;              // reduction length / output width
    d.transpose = form ? 1 : 0;
    d.nterm = mode == 6 ? 3 : 2;
    float dummy;
    if (!(has_pro && KD > 512) && panel_ok(KD, MD, &dummy, MD, planes, planes_bytes, n_rows)) {
        const int WC = MD > 256 ? 4 : MD > 128 ? 2 : 1;
        const int NJ = MD > 128 ? 4 : MD > 64 ? 4 : MD > 32 ? 2 : 1;
        d.P = 32 * NJ * WC;
        d.MD = MD;
        d.KD = KD;
        if (mode == 6 && WC > 1 && gemm_f16()) {
            d.kind = kWPanelF16;
            d.nterm = 2;
        } else {
            d.kind = kWPanelB16;
        }
        return d;
    }
    if (ws_ok(KD, MD, (const float*)nullptr, 4, planes, planes_bytes)) {
        d.kind = kWTiled;
        d.MD = MD;
        d.KD = KD;
        d.P = 64 * (MD > 64 ? 2 : 1);
        return d;
    }
    if (KD % 8 == 0) {                                           // presplit_w(W, ldw, M, K, transpose, ...)
        d.kind = kWPlain;
        d.MD = M;
        d.KD = K;
    }
    return d;
}

extern "C" int ddmp_gemm_next_prepared(void) {
    g_w_next_prepared = true;
    return DDMP_OK;
}
// the owner of a plane buffer is about to free it (or to stop maintaining it): drop what this thread recorded for it, so that
// a later allocation at the same address is never taken for prepared planes.  planes == NULL: everything.
extern "C" int ddmp_gemm_forget_planes(const void* planes) {
    for (size_t e = 0; e < g_w_registry.size();)
        if (planes == nullptr || g_w_registry[e].planes == planes) g_w_registry.erase(g_w_registry.begin() + e); else ++e;
    return DDMP_OK;
}

extern "C" int ddmp_gemm_prepare_weights(int n, const float* const* W, const int64_t* ldw, const int* M, const int* K,
                                         const int* form, const int* has_pro, void* const* planes, const size_t* planes_bytes,
                                         int64_t n_rows, float* scratch, ddmp_stream stream) {
    ARG_TRY(n > 0 && n <= kWPrepMax && W && ldw && M && K && form && planes && planes_bytes && scratch && n_rows > 0);
    WPrepBatch b;
    b.n = 0;
    bool any_f16 = false;
    for (int i = 0; i < n; ++i) {
        ARG_TRY(W[i] && M[i] > 0 && K[i] > 0 && ldw[i] >= K[i] && aligned16(W[i]) && ldw[i] % 4 == 0);
        const WPrepDesc d = plan_w(form[i], n_rows, W[i], ldw[i], M[i], K[i], has_pro ? has_pro[i] : 0, planes[i], planes_bytes[i]);
        // whatever was recorded for this buffer is overwritten (or no longer maintained) from here on
        for (size_t e = 0; e < g_w_registry.size();)
            if (g_w_registry[e].planes == planes[i]) g_w_registry.erase(g_w_registry.begin() + e); else ++e;
        if (d.kind == 0) continue;
        any_f16 |= d.kind == kWPanelF16;
        b.d[b.n++] = d;
    }
    if (b.n == 0) return DDMP_OK;
    hipStream_t st = (hipStream_t)stream;
    if (any_f16) hipLaunchKernelGGL(wprep_max_kernel, dim3(8, (unsigned)b.n), dim3(256), 0, st, b, scratch);
    hipLaunchKernelGGL(wprep_split_kernel, dim3(128, (unsigned)b.n), dim3(256), 0, st, b, (const float*)scratch);
    LAUNCH_TRY();
    for (int i = 0; i < b.n; ++i) g_w_registry.push_back(b.d[i]);
    return DDMP_OK;
}

extern "C" int ddmp_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y,
                                int64_t ldy, int64_t n_rows, int K, int M, const float* bias,
                                const float* pro_scale, const float* pro_shift, float slope,
                                void* workspace, size_t workspace_bytes, ddmp_stream stream) {
    const ScaleCtx ctx = take_scale_ctx();
    ARG_TRY(A && W && Y && n_rows > 0 && K > 0 && M > 0 && n_rows < INT32_MAX);
    ARG_TRY(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= K && ldy >= M);
    ARG_TRY(aligned16(A) && aligned16(W));
    ARG_TRY((pro_scale == nullptr) == (pro_shift == nullptr));
    ARG_TRY(!pro_scale || (aligned16(pro_scale) && aligned16(pro_shift)));
    hipStream_t st = (hipStream_t)stream;
    const int n_row_tiles = (int)cdiv(n_rows, kBM);
    const int TN = M > 64 ? 2 : 1;
    const int n_col_tiles = (int)cdiv(M, 64 * TN);
    dim3 grid((unsigned)(cdiv(n_row_tiles, kXcd) * kXcd * n_col_tiles)), block(256);
    if (gemm_mode() != 0 && !(pro_scale && K > 512) && panel_ok(K, M, Y, ldy, workspace, workspace_bytes, n_rows)) {
        if (pro_scale) launch_panel<1>(gemm_mode(), A, lda, nullptr, 0, W, ldw, 0, workspace, Y, ldy, (int)n_rows, K, M, bias, pro_scale, pro_shift, nullptr, nullptr, slope, st, nullptr, nullptr, ctx);
        else launch_panel<0>(gemm_mode(), A, lda, nullptr, 0, W, ldw, 0, workspace, Y, ldy, (int)n_rows, K, M, bias, nullptr, nullptr, nullptr, nullptr, slope, st, nullptr, nullptr, ctx);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    if (gemm_mode() != 0 && !(pro_scale && K > 512) && ws_ok(K, M, Y, ldy, workspace, workspace_bytes)) {
        if (pro_scale) launch_ws<true>(gemm_mode(), A, lda, W, ldw, 0, workspace, Y, ldy, (int)n_rows, K, M, bias, pro_scale, pro_shift, slope, st);
        else launch_ws<false>(gemm_mode(), A, lda, W, ldw, 0, workspace, Y, ldy, (int)n_rows, K, M, bias, nullptr, nullptr, slope, st);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    const bool pre = !(pro_scale && K > 512) && presplit_w(W, ldw, M, K, 0, workspace, workspace_bytes, st);
    const float* Bop = pre ? (const float*)workspace : W;
#define DDMP_LAUNCH_NT(KERNEL_, PRO_)                                                                    \
    hipLaunchKernelGGL((KERNEL_), grid, block, 0, st, A, lda, Bop, ldw, Y, ldy, (int)n_rows, K, M, bias, \
                       pro_scale, pro_shift, slope, n_row_tiles, n_col_tiles)
#define DDMP_NT_PICK(TN_, NTERM_, PRO_)                                                                   \
    do {                                                                                                  \
        if (pre && kt_) DDMP_LAUNCH_NT((gemm_rows_bf16_kernel<TN_, NTERM_, false, PRO_, true, true>), PRO_);   \
        else if (pre) DDMP_LAUNCH_NT((gemm_rows_bf16_kernel<TN_, NTERM_, false, PRO_, false, true>), PRO_);    \
        else if (kt_) DDMP_LAUNCH_NT((gemm_rows_bf16_kernel<TN_, NTERM_, false, PRO_, true, false>), PRO_);    \
        else DDMP_LAUNCH_NT((gemm_rows_bf16_kernel<TN_, NTERM_, false, PRO_, false, false>), PRO_);            \
    } while (0)
#define DDMP_NT_BY_MODE(TN_, PRO_)                                                          \
    do {                                                                                    \
        const int mode_ = (PRO_ && K > 512) ? 0 : gemm_mode();                              \
        const bool kt_ = (K % 16) != 0;                                                     \
        if (mode_ == 6) DDMP_NT_PICK(TN_, 3, PRO_);                                         \
        else if (mode_ == 3) DDMP_NT_PICK(TN_, 2, PRO_);                                    \
        else hipLaunchKernelGGL((gemm_rows_kernel<TN_, false, PRO_>), grid, block, 0, st, A, lda, W, ldw, Y, ldy,  \
                                (int)n_rows, K, M, bias, pro_scale, pro_shift, slope, n_row_tiles, n_col_tiles); \
    } while (0)
    if (TN == 2) {
        if (pro_scale) DDMP_NT_BY_MODE(2, true); else DDMP_NT_BY_MODE(2, false);
    } else {
        if (pro_scale) DDMP_NT_BY_MODE(1, true); else DDMP_NT_BY_MODE(1, false);
    }
#undef DDMP_NT_PICK
#undef DDMP_NT_BY_MODE
#undef DDMP_LAUNCH_NT
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_gemm_nn_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y,
                                int64_t ldy, int64_t n_rows, int M, int K, void* workspace,
                                size_t workspace_bytes, ddmp_stream stream) {
    // Y[n,K] = A[n,M] . W[M,K] : reduction over M, output width K
    const ScaleCtx ctx = take_scale_ctx();
    ARG_TRY(A && W && Y && n_rows > 0 && K > 0 && M > 0 && n_rows < INT32_MAX);
    ARG_TRY(M % 4 == 0 && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= M && ldw >= K && ldy >= K);
    ARG_TRY(aligned16(A) && aligned16(W));
    hipStream_t st = (hipStream_t)stream;
    const int n_row_tiles = (int)cdiv(n_rows, kBM);
    const int TN = K > 64 ? 2 : 1;
    const int n_col_tiles = (int)cdiv(K, 64 * TN);
    dim3 grid((unsigned)(cdiv(n_row_tiles, kXcd) * kXcd * n_col_tiles)), block(256);
    // pre-split W^T: planes [K_out][M] so that the reduction index M is contiguous; the rows kernel then runs
    // in its row-major (NT) form on the planes
    if (gemm_mode() != 0 && panel_ok(M, K, Y, ldy, workspace, workspace_bytes, n_rows)) {
        launch_panel<0>(gemm_mode(), A, lda, nullptr, 0, W, ldw, 1, workspace, Y, ldy, (int)n_rows, M, K, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, st, nullptr, nullptr, ctx);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    if (gemm_mode() != 0 && ws_ok(M, K, Y, ldy, workspace, workspace_bytes)) {
        launch_ws<false>(gemm_mode(), A, lda, W, ldw, 1, workspace, Y, ldy, (int)n_rows, M, K, nullptr, nullptr, nullptr, 0.f, st);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    const bool pre = presplit_w(W, ldw, M, K, 1, workspace, workspace_bytes, st);
#define DDMP_LAUNCH_NN(KERNEL_)                                                                        \
    hipLaunchKernelGGL((KERNEL_), grid, block, 0, st, A, lda, W, ldw, Y, ldy, (int)n_rows, M, K, nullptr, \
                       nullptr, nullptr, 0.f, n_row_tiles, n_col_tiles)
#define DDMP_LAUNCH_NNP(KERNEL_)                                                                       \
    hipLaunchKernelGGL((KERNEL_), grid, block, 0, st, A, lda, (const float*)workspace, (int64_t)M, Y, ldy, \
                       (int)n_rows, M, K, nullptr, nullptr, nullptr, 0.f, n_row_tiles, n_col_tiles)
    const int mode = gemm_mode();
    const bool kt = (M % 16) != 0;                 // reduction dimension of the NN form is M
    if (pre) {
        if (TN == 2) {
            if (mode == 6 && kt) DDMP_LAUNCH_NNP((gemm_rows_bf16_kernel<2, 3, false, false, true, true>));
            else if (mode == 6) DDMP_LAUNCH_NNP((gemm_rows_bf16_kernel<2, 3, false, false, false, true>));
            else if (kt) DDMP_LAUNCH_NNP((gemm_rows_bf16_kernel<2, 2, false, false, true, true>));
            else DDMP_LAUNCH_NNP((gemm_rows_bf16_kernel<2, 2, false, false, false, true>));
        } else {
            if (mode == 6 && kt) DDMP_LAUNCH_NNP((gemm_rows_bf16_kernel<1, 3, false, false, true, true>));
            else if (mode == 6) DDMP_LAUNCH_NNP((gemm_rows_bf16_kernel<1, 3, false, false, false, true>));
            else if (kt) DDMP_LAUNCH_NNP((gemm_rows_bf16_kernel<1, 2, false, false, true, true>));
            else DDMP_LAUNCH_NNP((gemm_rows_bf16_kernel<1, 2, false, false, false, true>));
        }
    } else if (TN == 2) {
        if (mode == 6 && kt) DDMP_LAUNCH_NN((gemm_rows_bf16_kernel<2, 3, true, false, true, false>));
        else if (mode == 6) DDMP_LAUNCH_NN((gemm_rows_bf16_kernel<2, 3, true, false, false, false>));
        else if (mode == 3 && kt) DDMP_LAUNCH_NN((gemm_rows_bf16_kernel<2, 2, true, false, true, false>));
        else if (mode == 3) DDMP_LAUNCH_NN((gemm_rows_bf16_kernel<2, 2, true, false, false, false>));
        else DDMP_LAUNCH_NN((gemm_rows_kernel<2, true, false>));
    } else {
        if (mode == 6 && kt) DDMP_LAUNCH_NN((gemm_rows_bf16_kernel<1, 3, true, false, true, false>));
        else if (mode == 6) DDMP_LAUNCH_NN((gemm_rows_bf16_kernel<1, 3, true, false, false, false>));
        else if (mode == 3 && kt) DDMP_LAUNCH_NN((gemm_rows_bf16_kernel<1, 2, true, false, true, false>));
        else if (mode == 3) DDMP_LAUNCH_NN((gemm_rows_bf16_kernel<1, 2, true, false, false, false>));
        else DDMP_LAUNCH_NN((gemm_rows_kernel<1, true, false>));
    }
#undef DDMP_LAUNCH_NNP
#undef DDMP_LAUNCH_NN
    LAUNCH_TRY();
    return DDMP_OK;
}

// The wide f16x3 wgrad: row-major staging kernel of round 4 (gemm_tn_rm.hip).  One split of each operand is one buffer
// descriptor: its bytes (and the offsets of the rows the kernel requests beyond it: up to 3 stages of 16) fit 31 bits.
static inline bool tn_wide_ok(int rows_per_split, int64_t ld0, int64_t ld1, int64_t ld2) {
    const int64_t r = (int64_t)rows_per_split + 64, lim = (int64_t)1 << 31;
    return r * ld0 * 4 < lim && r * ld1 * 4 < lim && r * ld2 * 4 < lim;
}
static void launch_tn_wide(const float* G, int64_t ldg, const float* G2, int64_t ldg2, const float* Z, int64_t ldz, float* part,
                           int64_t sstride, int64_t n_rows, int M, int K, const TnPlan& p, const float* ps, const float* psh,
                           const float* ga, const float* gb, const float* gk1, const float* gk0, float slope, float* gslot,
                           float* zslot, int target, int heal, hipStream_t st) {
    ddmp::TnRmArgs a;
    a.G = G; a.ldg = ldg; a.G2 = G2; a.ldg2 = ldg2; a.Z = Z; a.ldz = ldz; a.part = part; a.ld_out = K; a.split_stride = sstride;
    a.n_rows = (int)n_rows; a.M = M; a.K = K; a.rows_per_split = p.rows_per_split; a.n_tiles_m = p.n_tiles_m;
    a.n_tiles_k = p.n_tiles_k; a.n_splits = p.n_splits; a.pscale = ps; a.pshift = psh; a.ga = ga; a.gb = gb; a.gk1 = gk1;
    a.gk0 = gk0; a.slope = slope; a.gslot = gslot; a.zslot = zslot; a.target = target; a.heal = heal;
    a.tm = p.tm; a.tk = p.tk;
    ddmp::launch_tn_rm(a, st);
}

extern "C" size_t ddmp_gemm_tn_workspace_bytes(int64_t n_rows, int M, int K) {
    if (n_rows <= 0 || M <= 0 || K <= 0) return 0;
    const TnPlan p = tn_plan(n_rows, M, K), q = tn_plan(n_rows, M, K, true);    // (whichever route the call takes)
    return (size_t)std::max(p.n_splits, q.n_splits) * (size_t)M * (size_t)K * sizeof(float) + 64;    // + two scale slots (f16 modes)
}

extern "C" int ddmp_gemm_tn_f32(const float* G, int64_t ldg, const float* Z, int64_t ldz, float* dW,
                                int64_t lddw, int64_t n_rows, int M, int K, const float* pro_scale,
                                const float* pro_shift, float slope, void* workspace,
                                size_t workspace_bytes, ddmp_stream stream) {
    const ScaleCtx ctx = take_scale_ctx();
    ARG_TRY(G && Z && dW && n_rows > 0 && M > 0 && K > 0 && n_rows < INT32_MAX);
    ARG_TRY(M % 4 == 0 && K % 4 == 0 && ldg % 4 == 0 && ldz % 4 == 0 && ldg >= M && ldz >= K && lddw >= K);
    ARG_TRY(aligned16(G) && aligned16(Z));
    ARG_TRY((pro_scale == nullptr) == (pro_shift == nullptr));
    ARG_TRY(!pro_scale || (aligned16(pro_scale) && aligned16(pro_shift)));
    hipStream_t st = (hipStream_t)stream;
    const bool f16_route = gemm_mode() == 6 && gemm_f16();
    TnPlan p = tn_plan(n_rows, M, K, f16_route);
    if (p.T == 4 && p.tm + p.tk < 512 && !tn_wide_ok(p.rows_per_split, ldg, 0, ldz)) p = tn_plan(n_rows, M, K);
    const size_t need = (size_t)p.n_splits * (size_t)M * (size_t)K * sizeof(float) + 64;
    if (!workspace || workspace_bytes < need) return DDMP_EWORKSPACE;
    float* part = (float*)workspace;
    const int64_t sstride = (int64_t)M * K;
    const int n_tiles = p.n_tiles_m * p.n_tiles_k;
    if (p.T == 4) {
        dim3 pgrid((unsigned)(cdiv(p.n_splits, kXcd) * kXcd * n_tiles)), pblock(512);
        float* gslot = nullptr;
        float* zslot = nullptr;
        int target = 0, heal_ = 0;
#define DDMP_LAUNCH_TNP(KERNEL_)                                                                              \
    hipLaunchKernelGGL((KERNEL_), pgrid, pblock, 0, st, G, ldg, (const float*)nullptr, (int64_t)0, Z, ldz, part, \
                       (int64_t)K, sstride, (int)n_rows, M, K, p.rows_per_split, p.n_tiles_m, p.n_tiles_k,       \
                       p.n_splits, pro_scale, pro_shift, (const float*)nullptr, (const float*)nullptr,           \
                       (const float*)nullptr, (const float*)nullptr, slope, gslot, zslot, target, heal_)
        const int mode_ = gemm_mode();
        if (mode_ == 6 && gemm_f16() && tn_wide_ok(p.rows_per_split, ldg, 0, ldz)) {
            float* tail = (float*)((char*)workspace + need - 64);
            const bool own = !(ctx.a && ctx.b);
            gslot = own ? tail : ctx.a;
            zslot = own ? tail + 4 : ctx.b;
            const bool prime = own || ctx.prime;
            target = prime ? kF16TargetExact : kF16TargetStale;
            if (prime) {
                f16s_measure<0>(G, ldg, nullptr, 0, n_rows, M, nullptr, nullptr, nullptr, nullptr, slope, gslot, st);
                if (pro_scale) f16s_measure<1>(Z, ldz, nullptr, 0, n_rows, K, pro_scale, pro_shift, nullptr, nullptr, slope, zslot, st);
                else f16s_measure<0>(Z, ldz, nullptr, 0, n_rows, K, nullptr, nullptr, nullptr, nullptr, slope, zslot, st);
            }
            for (heal_ = 0; heal_ <= (prime ? 0 : 1); ++heal_)        // second launch: redo on overflow (gemm_f16s.inc)
                launch_tn_wide(G, ldg, nullptr, 0, Z, ldz, part, sstride, n_rows, M, K, p, pro_scale, pro_shift, nullptr, nullptr,
                               nullptr, nullptr, slope, gslot, zslot, target, heal_, st);
            heal_ = 0;
        } else if (pro_scale) {
            if (mode_ == 6) DDMP_LAUNCH_TNP((gemm_tn_panel_kernel<3, true, false>)); else DDMP_LAUNCH_TNP((gemm_tn_panel_kernel<2, true, false>));
        } else {
            if (mode_ == 6) DDMP_LAUNCH_TNP((gemm_tn_panel_kernel<3, false, false>)); else DDMP_LAUNCH_TNP((gemm_tn_panel_kernel<2, false, false>));
        }
#undef DDMP_LAUNCH_TNP
        LAUNCH_TRY();
        hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)cdiv((int64_t)M * K, 256)), dim3(256), 0, st, part,
                           sstride, p.n_splits, dW, lddw, M, K);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    dim3 grid((unsigned)(cdiv(p.n_splits, kXcd) * kXcd * n_tiles)), block(256);
#define DDMP_LAUNCH_TN(KERNEL_)                                                                           \
    hipLaunchKernelGGL((KERNEL_), grid, block, 0, st, G, ldg, Z, ldz, part, (int64_t)K, sstride, (int)n_rows, \
                       M, K, p.rows_per_split, p.n_tiles_m, p.n_tiles_k, p.n_splits, pro_scale, pro_shift, slope)
#define DDMP_TN_BY_MODE(T_, PRO_)                                                 \
    do {                                                                          \
        const int mode_ = gemm_mode();                                            \
        if (mode_ == 6) DDMP_LAUNCH_TN((gemm_tn_bf16_kernel<T_, 3, PRO_, (T_ == 1 ? 32 : 16)>));       \
        else if (mode_ == 3) DDMP_LAUNCH_TN((gemm_tn_bf16_kernel<T_, 2, PRO_, (T_ == 1 ? 32 : 16)>));  \
        else DDMP_LAUNCH_TN((gemm_tn_kernel<T_, PRO_>));                          \
    } while (0)
    if (p.T == 2) {
        if (pro_scale) DDMP_TN_BY_MODE(2, true); else DDMP_TN_BY_MODE(2, false);
    } else {
        if (pro_scale) DDMP_TN_BY_MODE(1, true); else DDMP_TN_BY_MODE(1, false);
    }
#undef DDMP_TN_BY_MODE
#undef DDMP_LAUNCH_TN
    LAUNCH_TRY();
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)cdiv((int64_t)M * K, 256)), dim3(256), 0, st, part,
                       sstride, p.n_splits, dW, lddw, M, K);
    LAUNCH_TRY();
    return DDMP_OK;
}


// ---- BatchNorm+LeakyReLU backward fused into the operand load of the two GEMMs that consume dY (agg-first layers)
// dgrad of a transform-first layer with the NEXT BatchNorm-backward reductions from its epilogue (row-register kernel):
//   out[n, K] = A[n, M] . W[M, K]   and   sums2[2K] = (sum_rows g, sum_rows g yhat) of out as the gradient behind the previous
//   layer's BatchNorm+LeakyReLU (Yp = that layer's conv output [n, K]; scale, shift, mean, rstd = its bn4 rows) --
//   what ddmp_bn_bwd_reduce_f32(out, Yp, ...) returns, without reading out again
extern "C" int ddmp_gemm_nn_bnred_supported(int M, int K, int64_t n_rows) {
    if (gemm_mode() == 6 && gemm_f16() && rr_enabled() && panel_enabled() && n_rows >= kRRMinRows && M % 32 == 0 && M >= 64 &&
        M <= kMaxProK && K > 128 && K <= 512 && K % 4 == 0)
        return 1;                                                // wide outputs: row-register kernel, STATS = 2
    static const bool narrow = !ddmp::unfused("dgrad_red_narrow");   // (A/B)
    // narrow outputs (128 | 64 columns): the 512-row panel kernel's RS epilogue (bf16 split terms)
    return (narrow && (gemm_mode() == 6 || gemm_mode() == 3) && panel_enabled() && n_rows >= kPanelMinRows && M % 32 == 0 &&
            M >= 32 && M <= kMaxProK && (K == 128 || K == 64)) ? 2 : 0;
}
extern "C" int ddmp_gemm_nn_bnred_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* out, int64_t ld_out,
                                      int64_t n_rows, int M, int K, const float* Yp, int64_t ldyp, const float* scale,
                                      const float* shift, const float* mean, const float* rstd, float slope, double* sums2,
                                      void* workspace, size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes,
                                      ddmp_stream stream) {
    ddmp::FinalizeScope fin_scope(sums2, stream, K);
    const ScaleCtx ctx = take_scale_ctx();
    ARG_TRY(A && W && out && Yp && scale && shift && mean && rstd && sums2 && n_rows > 0 && n_rows < INT32_MAX);
    ARG_TRY(M % 4 == 0 && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= M && ldw >= K && ld_out >= K && ldyp >= K);
    ARG_TRY(aligned16(A) && aligned16(W));
    const int form = ddmp_gemm_nn_bnred_supported(M, K, n_rows);
    if (!form || (form == 1 && !rr_route_ok(n_rows, M, lda, 0)) || !panel_ok(M, K, out, ld_out, workspace, workspace_bytes, n_rows))
        return DDMP_EINVAL;
    if (stats_ws_bytes < ddmp_gemm_nt_stats_workspace_bytes(n_rows, K)) return DDMP_EWORKSPACE;
    const float* bn4[4] = {scale, shift, mean, rstd};
    launch_panel<0>(gemm_mode(), A, lda, nullptr, 0, W, ldw, 1, workspace, out, ld_out, (int)n_rows, M, K, nullptr, nullptr, nullptr,
                    nullptr, nullptr, slope, (hipStream_t)stream, (double*)stats_ws, sums2, ctx, Yp, ldyp, bn4);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_gemm_bnbwd_supported(int cout, int cin, int64_t n_rows) {
    if (gemm_mode() == 0 || !panel_enabled() || !tn_panel_enabled() || n_rows < kTnPanelMinRows) return 0;
    // wide layers (cin > 128): f16x3 row-register / row-panel dgrad + 256 x 256 wgrad panels; narrow aggregate-first layers
    // (round 3: 32 -> 64 ... 128 -> 256): the 512-row panel dgrad and the tiled wgrad with the same operand prologue
    const bool nn = cout % 32 == 0 && cout >= 64 && cout <= kMaxProK && cin % 4 == 0 && cin >= 16 && cin <= 512;
    const bool tn = cout % 4 == 0 && ((cout >= 256 && cin >= 256) || cin <= 128);
    static const bool narrow = !ddmp::unfused("bnbwd_narrow");   // (A/B)
    return (nn && tn && (cin > 128 || narrow)) ? 1 : 0;
}

// (the wgrad alone: layer 0 has no dgrad)
extern "C" int ddmp_gemm_tn_bnbwd_supported(int cout, int cin, int64_t n_rows) {
    if (gemm_mode() == 0 || !tn_panel_enabled() || n_rows < kTnPanelMinRows) return 0;
    const int mode_ = gemm_mode();
    const bool panel = cout >= 256 && cin >= 256;
    return (cout % 4 == 0 && cin % 4 == 0 && (panel || mode_ == 6 || mode_ == 3)) ? 1 : 0;
}

extern "C" int ddmp_gemm_nn_bnbwd_f32(const float* dZ, int64_t lddz, const float* Yb, int64_t ldyb, const float* W,
                                      int64_t ldw, float* out, int64_t ld_out, int64_t n_rows, int M, int K,
                                      const float* a, const float* b, const float* c1, const float* c0, float slope,
                                      void* workspace, size_t workspace_bytes, ddmp_stream stream) {
    // out[n,K] = dY[n,M] . W[M,K],  dY = a * dZ * lrelu'(a * Yb + b) + c1 * Yb + c0  (per column of M)
    const ScaleCtx ctx = take_scale_ctx();
    ARG_TRY(dZ && Yb && W && out && a && b && c1 && c0 && n_rows > 0 && n_rows < INT32_MAX && M > 0 && K > 0);
    ARG_TRY(lddz % 4 == 0 && ldyb % 4 == 0 && ldw % 4 == 0 && lddz >= M && ldyb >= M && ldw >= K && ld_out >= K);
    ARG_TRY(aligned16(dZ) && aligned16(Yb) && aligned16(W));
    if (!ddmp_gemm_bnbwd_supported(M, K, n_rows) || !panel_ok(M, K, out, ld_out, workspace, workspace_bytes, n_rows)) return DDMP_EINVAL;
    launch_panel<2>(gemm_mode(), dZ, lddz, Yb, ldyb, W, ldw, 1, workspace, out, ld_out, (int)n_rows, M, K, nullptr, a, b,
                    c1, c0, slope, (hipStream_t)stream, nullptr, nullptr, ctx);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_gemm_tn_bnbwd_f32(const float* dZ, int64_t lddz, const float* Yb, int64_t ldyb, const float* Z,
                                      int64_t ldz, float* dW, int64_t lddw, int64_t n_rows, int M, int K,
                                      const float* a, const float* b, const float* c1, const float* c0,
                                      const float* pro_scale, const float* pro_shift, float slope, void* workspace,
                                      size_t workspace_bytes, ddmp_stream stream) {
    // dW[M,K] = dY^T . f(Z),  dY as above (columns of M), f = optional BatchNorm+LeakyReLU prologue on Z (columns of K)
    const ScaleCtx ctx = take_scale_ctx();
    ARG_TRY(dZ && Yb && Z && dW && a && b && c1 && c0 && n_rows > 0 && n_rows < INT32_MAX && M > 0 && K > 0);
    ARG_TRY(M % 4 == 0 && K % 4 == 0 && lddz >= M && ldyb >= M && ldz >= K && lddw >= K);
    ARG_TRY((pro_scale == nullptr) == (pro_shift == nullptr));
    if (!ddmp_gemm_tn_bnbwd_supported(M, K, n_rows)) return DDMP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool f16_route = gemm_mode() == 6 && gemm_f16() && lddz % 4 == 0 && ldyb % 4 == 0 && ldz % 4 == 0 && aligned16(dZ) &&
                           aligned16(Yb) && aligned16(Z);
    TnPlan p = tn_plan(n_rows, M, K, f16_route);
    if (p.T == 4 && p.tm + p.tk < 512 && !tn_wide_ok(p.rows_per_split, lddz, ldyb, ldz)) p = tn_plan(n_rows, M, K);
    const size_t need = (size_t)p.n_splits * (size_t)M * (size_t)K * sizeof(float) + 64;
    if (!workspace || workspace_bytes < need) return DDMP_EWORKSPACE;
    float* part = (float*)workspace;
    const int64_t sstride = (int64_t)M * K;
    const int n_tiles = p.n_tiles_m * p.n_tiles_k;
    if (p.T != 4) {                                              // narrow layers: 64 x 64 | 128 x 128 tiles, bf16 split terms
        const int mode_ = gemm_mode();
        if (mode_ != 6 && mode_ != 3) return DDMP_EINVAL;
        dim3 grid((unsigned)(cdiv(p.n_splits, kXcd) * kXcd * n_tiles)), block(256);
#define DDMP_LAUNCH_TNG(T_, NT_, PRO_)                                                                                 \
    hipLaunchKernelGGL((gemm_tn_bf16_kernel<T_, NT_, PRO_, (T_ == 1 ? 32 : 16), true>), grid, block, 0, st, dZ, lddz, Z, ldz,  \
                       part, (int64_t)K, sstride, (int)n_rows, M, K, p.rows_per_split, p.n_tiles_m, p.n_tiles_k,       \
                       p.n_splits, pro_scale, pro_shift, slope, Yb, ldyb, a, b, c1, c0)
        if (p.T == 2) {
            if (mode_ == 6) { if (pro_scale) DDMP_LAUNCH_TNG(2, 3, true); else DDMP_LAUNCH_TNG(2, 3, false); }
            else { if (pro_scale) DDMP_LAUNCH_TNG(2, 2, true); else DDMP_LAUNCH_TNG(2, 2, false); }
        } else {
            if (mode_ == 6) { if (pro_scale) DDMP_LAUNCH_TNG(1, 3, true); else DDMP_LAUNCH_TNG(1, 3, false); }
            else { if (pro_scale) DDMP_LAUNCH_TNG(1, 2, true); else DDMP_LAUNCH_TNG(1, 2, false); }
        }
#undef DDMP_LAUNCH_TNG
        LAUNCH_TRY();
        hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)cdiv((int64_t)M * K, 256)), dim3(256), 0, st, part, sstride,
                           p.n_splits, dW, lddw, M, K);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    dim3 pgrid((unsigned)(cdiv(p.n_splits, kXcd) * kXcd * n_tiles)), pblock(512);
    float* gslot = nullptr;
    float* zslot = nullptr;
    int target = 0, heal_ = 0;
#define DDMP_LAUNCH_TNP(KERNEL_)                                                                                 \
    hipLaunchKernelGGL((KERNEL_), pgrid, pblock, 0, st, dZ, lddz, Yb, ldyb, Z, ldz, part, (int64_t)K, sstride,     \
                       (int)n_rows, M, K, p.rows_per_split, p.n_tiles_m, p.n_tiles_k, p.n_splits, pro_scale,       \
                       pro_shift, a, b, c1, c0, slope, gslot, zslot, target, heal_)
    const int mode_ = gemm_mode();
    if (mode_ == 6 && gemm_f16() && lddz % 4 == 0 && ldyb % 4 == 0 && ldz % 4 == 0 && tn_wide_ok(p.rows_per_split, lddz, ldyb, ldz) &&
        aligned16(dZ) && aligned16(Yb) && aligned16(Z)) {
        float* tail = (float*)((char*)workspace + need - 64);
        const bool own = !(ctx.a && ctx.b);
        gslot = own ? tail : ctx.a;
        zslot = own ? tail + 4 : ctx.b;
        const bool prime = own || ctx.prime;
        target = prime ? kF16TargetExact : kF16TargetStale;
        if (prime) {
            f16s_measure<2>(dZ, lddz, Yb, ldyb, n_rows, M, a, b, c1, c0, slope, gslot, st);
            if (pro_scale) f16s_measure<1>(Z, ldz, nullptr, 0, n_rows, K, pro_scale, pro_shift, nullptr, nullptr, slope, zslot, st);
            else f16s_measure<0>(Z, ldz, nullptr, 0, n_rows, K, nullptr, nullptr, nullptr, nullptr, slope, zslot, st);
        }
        for (heal_ = 0; heal_ <= (prime ? 0 : 1); ++heal_)            // second launch: redo on overflow (gemm_f16s.inc)
            launch_tn_wide(dZ, lddz, Yb, ldyb, Z, ldz, part, sstride, n_rows, M, K, p, pro_scale, pro_shift, a, b, c1, c0, slope,
                           gslot, zslot, target, heal_, st);
        heal_ = 0;
    } else if (pro_scale) {
        if (mode_ == 6) DDMP_LAUNCH_TNP((gemm_tn_panel_kernel<3, true, true>)); else DDMP_LAUNCH_TNP((gemm_tn_panel_kernel<2, true, true>));
    } else {
        if (mode_ == 6) DDMP_LAUNCH_TNP((gemm_tn_panel_kernel<3, false, true>)); else DDMP_LAUNCH_TNP((gemm_tn_panel_kernel<2, false, true>));
    }
#undef DDMP_LAUNCH_TNP
    LAUNCH_TRY();
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)cdiv((int64_t)M * K, 256)), dim3(256), 0, st, part, sstride,
                       p.n_splits, dW, lddw, M, K);
    LAUNCH_TRY();
    return DDMP_OK;
}


// ---- forward GEMM that also returns the BatchNorm statistics of its output (column sums of Y and Y^2, float64 [2M]):
//      fused into the row-panel kernel's epilogue where that kernel runs, otherwise GEMM + ddmp_bn_stats_f32
extern "C" size_t ddmp_gemm_nt_stats_workspace_bytes(int64_t n_rows, int M) {
    if (n_rows <= 0 || M <= 0) return 0;
    const size_t fused = (size_t)(cdiv(n_rows, 64) + 8) * 2 * 512 * sizeof(double) + 256 + fpartials_mid_bytes(512);
    return std::max(fused, ddmp_colreduce_workspace_bytes(n_rows, M));
}

extern "C" int ddmp_gemm_nt_stats_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* Y, int64_t ldy,
                                      int64_t n_rows, int K, int M, const float* bias, const float* pro_scale,
                                      const float* pro_shift, float slope, double* sums2, void* workspace,
                                      size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes,
                                      ddmp_stream stream) {
    ddmp::FinalizeScope fin_scope(sums2, stream, M);
    const ScaleCtx ctx = take_scale_ctx();
    ARG_TRY(sums2 && stats_ws);
    if (stats_ws_bytes < ddmp_gemm_nt_stats_workspace_bytes(n_rows, M)) return DDMP_EWORKSPACE;
    const bool fused = A && W && Y && n_rows > 0 && n_rows < INT32_MAX && K > 0 && M > 0 && K % 4 == 0 && lda % 4 == 0 &&
                       ldw % 4 == 0 && lda >= K && ldw >= K && ldy >= M && aligned16(A) && aligned16(W) &&
                       (pro_scale == nullptr) == (pro_shift == nullptr) && gemm_mode() != 0 && !(pro_scale && K > 512) &&
                       panel_ok(K, M, Y, ldy, workspace, workspace_bytes, n_rows);
    if (fused) {
        hipStream_t st = (hipStream_t)stream;
        if (pro_scale) launch_panel<1>(gemm_mode(), A, lda, nullptr, 0, W, ldw, 0, workspace, Y, ldy, (int)n_rows, K, M, bias, pro_scale, pro_shift, nullptr, nullptr, slope, st, (double*)stats_ws, sums2, ctx);
        else launch_panel<0>(gemm_mode(), A, lda, nullptr, 0, W, ldw, 0, workspace, Y, ldy, (int)n_rows, K, M, bias, nullptr, nullptr, nullptr, nullptr, slope, st, (double*)stats_ws, sums2, ctx);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    g_scale_ctx = ctx;
    g_w_next_prepared = g_w_call_prepared;
    int rc = ddmp_gemm_nt_f32(A, lda, W, ldw, Y, ldy, n_rows, K, M, bias, pro_scale, pro_shift, slope, workspace,
                              workspace_bytes, stream);
    if (rc != DDMP_OK) return rc;
    return ddmp_bn_stats_f32(Y, ldy, n_rows, M, sums2, stats_ws, stats_ws_bytes, stream);
}
