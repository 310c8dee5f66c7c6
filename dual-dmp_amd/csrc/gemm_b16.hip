// Dense steps of GCNConv on bfloat16 features (bf16-feature mode, b16_common.h): GCNConv.lin and its autograd
// (util/networks.py:51-62 of the reference through PyG 2.2.0: X.W^T, dH.W, dH^T.X) with bf16 activations / activation
// gradients in HBM, float32 master weights, ONE v_mfma_f32_32x32x16_bf16 product per contraction step (no operand
// splitting, no scale slots), float32 accumulation, float32 weight gradients.
//
//   nt : Y[n,M]  = bf16( f(A[n,K]) . W[M,K]^T + bias )        forward (f = BatchNorm+LeakyReLU prologue, optional)
//   nn : Y[n,K]  = bf16( A[n,M] . W[M,K] )                    dgrad
//   tn : dW[M,K] = G[n,M]^T . f(Z[n,K])                       wgrad (float32 out, rows split over workgroups)
//
// At bf16 MFMA rates every one of these shapes is HBM-bound (1M x 512 x 512: 0.22 ms of matrix pipe at peak against
// 2 GB = 0.33 ms of HBM traffic at the 6.3 TB/s a copy reaches), so the kernels are built around the byte stream:
//   * row-panel form for nt / nn: one 512-thread workgroup owns ALL output columns of its rows (block tile 128 x 512,
//     256 x 256, 256 x 128 ...), persistent over row tiles -- A is read from HBM exactly once, Y written once;
//   * the weights are converted once per call into bf16 "planes" whose global layout IS the LDS image of a stage
//     (64-byte rows of 32 k, 16-byte slots swizzled by (row >> 2) & 3: conflict-free ds_read_b128 fragments) and go
//     global -> LDS with global_load_lds_dwordx4 (no registers, no ds_write);
//   * three stage buffers: W copies run two stages ahead, the A rows three (registers), one barrier per stage with a
//     counted vmcnt that leaves exactly the newest requests in flight;
//   * the MFMA is issued with its operands swapped (D = W_tile . A_tile^T), so a lane ends up holding 4 consecutive output
//     COLUMNS of one output row: two v_permlane32_swap make 16-byte pieces and the bf16 tile is stored without a trip
//     through LDS (16 dwordx4 stores per lane and tile instead of 128 dword stores in the float32 kernels);
//   * tn: both operands are contracted over rows, i.e. needed "8 consecutive rows of one column" -- they are staged
//     row-major (coalesced 16-byte loads, 64-byte chunks swizzled by row & 3) and read with ds_read_b64_tr_b16.
#include "b16_common.h"
#include "finalize.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

using namespace ddmp;

constexpr int kBK = 32;                       // contraction elements per stage (two MFMA k-steps)
constexpr int kMaxK = 512;                    // widest contraction with a prologue (coefficient tables in LDS)

__device__ __forceinline__ f32x16 mfma_b16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// element offset of (row, 16-byte slot) in a stage image of 32-element (64-byte) rows
__device__ __host__ __forceinline__ int img_off(int row, int slot) { return row * kBK + ((slot ^ ((row >> 2) & 3)) << 3); }

// W [M,K] (or its transpose) -> bf16 planes [stage][MP rows = output columns][32 contraction elements], zero padded
__global__ __launch_bounds__(256) void w_planes_b16_kernel(const float* __restrict__ W, int64_t ldw, int MD, int KD,
                                                           int transpose, int MP, __bf16* __restrict__ planes) {
    const int ns = (KD + kBK - 1) / kBK;
    const int total = ns * MP * kBK;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int kk = idx % kBK;
        const int r = (idx / kBK) % MP;
        const int st = idx / (kBK * MP);
        const int c = st * kBK + kk;
        float x = 0.f;
        if (r < MD && c < KD) x = transpose ? W[(int64_t)c * ldw + r] : W[(int64_t)r * ldw + c];
        planes[(int64_t)st * MP * kBK + img_off(r, kk >> 3) + (kk & 7)] = (__bf16)x;
    }
}

template <int N> __device__ __forceinline__ void wait_barrier() {
    // this wave's LDS writes (lgkmcnt) and all but its N youngest VMEM operations (register loads, global->LDS copies,
    // stores: one in-order counter on gfx9) have completed before it arrives
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void lds_barrier() {                  // LDS traffic only; register loads stay in flight
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ------------------------------------------------------------------------------------------------
//  Row-panel kernel.  8 waves as WR x WC, wave tile 64 rows x (32 NJ) columns, block tile (64 WR) x (32 NJ WC).
//  PM: 0 = A as is; 1 = lrelu(A * scale[k] + shift[k]) (BatchNorm+LeakyReLU of the previous layer on the operand load).
// ------------------------------------------------------------------------------------------------
template <int WR, int WC, int NJ, int PM>
__global__ __launch_bounds__(512) void gemm_rows_b16_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const __bf16* __restrict__ Bp, bf16_t* __restrict__ Y, int64_t ldy,
    int n_rows, int KD, int MD, const float* __restrict__ bias, const float* __restrict__ pscale,
    const float* __restrict__ pshift, float slope, int n_row_tiles) {
    static_assert(WR * WC == 8, "eight waves");
    constexpr int BMR = 64 * WR, WN = 32 * NJ, MP = WN * WC;
    constexpr int NA = BMR / 128;                                // 16-byte pieces of A per thread and stage
    constexpr int kAStage = BMR * kBK, kBStage = MP * kBK;       // bf16 elements
    constexpr int NCH = kBStage * 2 / 1024;                      // 1 KB chunks of a W stage
    constexpr int NCW = NCH >= 8 ? NCH / 8 : 1;                  // copies per wave and stage (narrow outputs: duplicates)
    constexpr int NV = NCW + NA;                                 // VMEM operations per wave and iteration
    constexpr int NST = 4 * NJ;                                  // epilogue stores per lane
    constexpr int NB = 3;
    static_assert(NCH >= 8 ? NCH % 8 == 0 : 8 % NCH == 0, "every wave issues the same number of W copies");
    static_assert(NV + NST <= 63, "vmcnt range");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NB * (kAStage + kBStage) * 2 + MP * 4 + (PM ? 2 * kMaxK * 4 : 0)];
    __bf16* As = reinterpret_cast<__bf16*>(smem);
    __bf16* Bs = As + NB * kAStage;
    float* s_bias = reinterpret_cast<float*>(Bs + NB * kBStage);
    float* s_pro = s_bias + MP;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ns = KD / kBK;
    const int my_tiles = (n_row_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = my_tiles * ns;
    const int last_t = (int)blockIdx.x + (my_tiles - 1) * (int)gridDim.x;

    if (PM) {
        for (int i = tid; i < KD; i += 512) {
            s_pro[i] = pscale[i];
            s_pro[kMaxK + i] = pshift[i];
        }
    }
    for (int i = tid; i < MP; i += 512) s_bias[i] = (bias && i < MD) ? bias[i] : 0.f;
    __syncthreads();

    // ---- A stream.  PM == 0: global -> LDS copies like W (16 rows x 64 bytes per wave instruction; the lane's SOURCE
    // slot is pre-swizzled so that the lane-linear LDS image is the swizzled one), two stages ahead, no registers.
    // PM == 1: global (bf16) -> registers (three stages ahead) -> prologue -> ds_write into the image.
    const int aslot = PM ? (tid & 3) : ((lane & 3) ^ ((lane >> 4) & 3));      // 16-byte slot of the 64-byte stage row
    const int ar = PM ? (tid >> 2) : (wave * 16 + (lane >> 2));               // row within a 128-row pass
    uint4 R[NB][NA];
    const bf16_t* aptr[NA];
    int lt = blockIdx.x, lk = 0;
    auto a_tile = [&](int t) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const int64_t row = min(t * BMR + p * 128 + ar, n_rows - 1);
            aptr[p] = A + row * lda + aslot * 8;
        }
    };
    auto a_copy = [&](int buf) {                                 // PM == 0: the next stage -> buffer buf
        __bf16* dst = As + buf * kAStage + wave * 16 * kBK;
#pragma unroll
        for (int p = 0; p < NA; ++p) dma16(aptr[p] + lk, dst + p * 128 * kBK);
    };
    a_tile(lt);
    auto a_load = [&](uint4 (&r)[NA]) {
#pragma unroll
        for (int p = 0; p < NA; ++p) r[p] = *reinterpret_cast<const uint4*>(aptr[p] + lk);
    };
    auto a_advance = [&]() {
        lk += kBK;
        if (lk == KD) {
            lk = 0;
            lt = min(lt + (int)gridDim.x, last_t);               // past the end: harmless re-loads of the last tile
            a_tile(lt);
        }
    };
    int sk = 0;
    auto a_store = [&](int buf, const uint4 (&r)[NA]) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            uint4 v = r[p];
            if (PM == 1) {
                float f[8], sc[8], sh[8];
                bf_unpack8(v, f);
                ld8f(&s_pro[sk + aslot * 8], sc);
                ld8f(&s_pro[kMaxK + sk + aslot * 8], sh);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = lrelu(fmaf(f[e], sc[e], sh[e]), slope);
                v = bf_pack8(f);
            }
            *reinterpret_cast<uint4*>(As + buf * kAStage + img_off(p * 128 + ar, aslot)) = v;
        }
        sk += kBK;
        sk = sk == KD ? 0 : sk;
    };
    // ---- W stream: planes -> LDS, 1 KB per wave instruction
    const __bf16* bsrc = Bp + lane * 8;
    int bk = 0;
    auto b_copy = [&](int buf) {
        __bf16* dst = Bs + buf * kBStage;
#pragma unroll
        for (int c0 = 0; c0 < NCW; ++c0) {
            const int c = NCH >= 8 ? c0 * 8 + wave : wave % NCH;
            dma16(bsrc + c * 512, dst + c * 512);
        }
        ++bk;
        const bool wrap = bk == ns;
        bk = wrap ? 0 : bk;
        bsrc = wrap ? Bp + lane * 8 : bsrc + kBStage;
    };

    // ---- MFMA side (operands swapped: the accumulator tile is D[m = output column][n = row])
    f32x16 acc[2][NJ];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    const int swz = (l31 >> 2) & 3;                              // rows of a fragment: base (multiple of 32) + l31
    int a_off[2], b_off[2];                                      // per k-step: slot 2 ks + lh
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_off[ks] = (wr * 64 + l31) * kBK + (((2 * ks + lh) ^ swz) << 3);
        b_off[ks] = (wc * WN + l31) * kBK + (((2 * ks + lh) ^ swz) << 3);
    }
    bf16x8 af[2][2], bf[2][NJ];
    auto rd = [&](int buf, int ks) {
        const __bf16* as = As + buf * kAStage + a_off[ks];
        const __bf16* bs = Bs + buf * kBStage + b_off[ks];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(as + i * 32 * kBK);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[ks][j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * kBK);
    };
    auto mm = [&](int ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma_b16(bf[ks][j], af[ks][i], acc[i][j]);
    };
    auto epilogue = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = t * BMR + wr * 64 + i * 32 + l31;
            bf16_t* yrow = Y + (int64_t)row * ldy;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int cb = wc * WN + j * 32;                 // this lane: columns cb + 8 g + 4 lh + (0..3), g = 0..3
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    unsigned w[2][2];                            // [group of the pair][dword]
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int g = 2 * gp + h;
                        const float4 bv = *reinterpret_cast<const float4*>(&s_bias[cb + 8 * g + 4 * lh]);
                        w[h][0] = bf_pack(acc[i][j][4 * g + 0] + bv.x, acc[i][j][4 * g + 1] + bv.y);
                        w[h][1] = bf_pack(acc[i][j][4 * g + 2] + bv.z, acc[i][j][4 * g + 3] + bv.w);
                    }
                    // lanes >= 32 of group 2gp <-> lanes < 32 of group 2gp+1: afterwards lh = 0 holds columns
                    // cb + 16 gp + (0..7), lh = 1 holds cb + 16 gp + (8..15), 16 contiguous bytes each
                    const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                    const int col = cb + 16 * gp + 8 * lh;
                    if (row < n_rows && col < MD)
                        *reinterpret_cast<uint4*>(yrow + col) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
        }
    };

    zero_acc();
    int cur_t = blockIdx.x, cnt = 0;
    bool stored = false;
    // iteration g: stage g from buffer g % 3; A image of stage g+1 <- ring slot (g+1) % 3; W copy of stage g+2;
    // ring slot g % 3 <- A loads of stage g+3
    auto iteration = [&](int buf, int bufa, int bufw, uint4 (&rl)[NA], const uint4 (&rs)[NA]) {
        rd(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(0);                                                   // the second k-step's fragments arrive behind these MFMAs
        rd(buf, 1);
        if (PM) a_store(bufa, rs);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        b_copy(bufw);
        asm volatile("" ::: "memory");                           // the A requests stay BEHIND the W copies in the VMEM stream
        if (PM) a_load(rl); else a_copy(bufw);
        __builtin_amdgcn_sched_barrier(0);
        a_advance();
        stored = false;
        if (++cnt == ns) {
            cnt = 0;
            epilogue(cur_t);
            zero_acc();
            // the counted wait below assumes EVERY lane issued all NST stores: only true for a full tile of a full-width
            // panel (the stores are predicated per lane; a wave whose lanes all fail issues none).  Otherwise the
            // smaller count is used, which only over-waits.
            stored = MD == MP && (cur_t + 1) * BMR <= n_rows;
            cur_t += gridDim.x;
        }
    };
    b_copy(0);
    b_copy(1);
    if (PM) {
        a_load(R[0]);
        a_advance();
        a_load(R[1]);
        a_advance();
        a_load(R[2]);
        a_advance();
        a_store(0, R[0]);
    } else {
        a_copy(0);
        a_advance();
        a_copy(1);
        a_advance();
    }
    wait_barrier<0>();
    for (int g = 0; g < G; g += 3) {
        iteration(0, 1, 2, R[0], R[1]);
        if (stored) wait_barrier<NV + NST>(); else wait_barrier<NV>();
        if (g + 1 < G) iteration(1, 2, 0, R[1], R[2]);
        if (stored) wait_barrier<NV + NST>(); else wait_barrier<NV>();
        if (g + 2 < G) iteration(2, 0, 1, R[2], R[0]);
        if (stored) wait_barrier<NV + NST>(); else wait_barrier<NV>();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no LDS copy outlives the workgroup
}

// contraction over 8 or 16 elements (the first layer: K = 16 | 8 input channels): plain VALU, 4 lanes per row,
// 8 output columns each -- 0.5 GFMA at 1M rows, nothing for the matrix pipe to win
template <int KD>
__global__ __launch_bounds__(256) void gemm_rows_smallk_b16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                   const float* __restrict__ W, int64_t ldw,
                                                                   bf16_t* __restrict__ Y, int64_t ldy, int n_rows, int MD,
                                                                   const float* __restrict__ bias) {
    extern __shared__ float s_w[];                               // [MD][KD] + bias[MD]
    for (int i = threadIdx.x; i < MD * KD; i += 256) s_w[i] = W[(int64_t)(i / KD) * ldw + (i % KD)];
    for (int i = threadIdx.x; i < MD; i += 256) s_w[MD * KD + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int groups = MD / 8;
    const int64_t total = (int64_t)n_rows * groups;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / groups;
        const int c0 = (int)(idx % groups) * 8;
        float a[KD];
#pragma unroll
        for (int q = 0; q < KD / 8; ++q) {
            float t[8];
            bf_unpack8(ld8b(A + row * lda + q * 8), t);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[q * 8 + e] = t[e];
        }
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float s = s_w[MD * KD + c0 + c];
#pragma unroll
            for (int k = 0; k < KD; ++k) s = fmaf(a[k], s_w[(c0 + c) * KD + k], s);
            o[c] = s;
        }
        st8b(Y + row * ldy + c0, bf_pack8(o));
    }
}

// ------------------------------------------------------------------------------------------------
//  wgrad:  partial[split][m][k] = sum_{rows of the split} G[row][m] * f(Z[row][k])        (float32)
//  One workgroup = one 256 x 256 output panel x one row range; 8 waves as 4 (m) x 2 (k), each 64 x 128.
//  Both operands are staged row-major [32 rows][256 columns] (16-byte coalesced loads -> ds_write_b128; the 64-byte
//  chunk c of row r sits at chunk c ^ (r & 3)) and the MFMA fragments -- 8 consecutive rows of one column -- come out
//  of ds_read_b64_tr_b16 (two per fragment).  Two LDS buffers, two register slots: the loads of stage s+3 are issued in
//  iteration s and consumed in iteration s+2.
// ------------------------------------------------------------------------------------------------
constexpr int kTT = 256;                                          // panel edge
__device__ __forceinline__ int tn_off(int row, int col) {         // element offset in a [32][256] stage image
    return row * kTT + ((((col >> 5) ^ (row & 3)) << 5) | (col & 31));
}

// GDUAL: the G operand is the BatchNorm+LeakyReLU backward of (G = dZ, G2 = Yb) rebuilt on the fly and rounded to bf16 --
// what bn_bwd_apply would have written (coefficients per column of G: ga, gb, gk1, gk0)
template <bool ZPRO, bool GDUAL = false>
__global__ __launch_bounds__(512) void gemm_tn_b16_kernel(
    const bf16_t* __restrict__ G, int64_t ldg, const bf16_t* __restrict__ Z, int64_t ldz, float* __restrict__ out,
    int64_t ld_out, int64_t split_stride, int n_rows, int M, int K, int rows_per_split, int n_tiles_m, int n_tiles_k,
    int n_splits, const float* __restrict__ pscale, const float* __restrict__ pshift, float slope,
    const bf16_t* __restrict__ G2 = nullptr, int64_t ldg2 = 0, const float* __restrict__ ga = nullptr,
    const float* __restrict__ gb = nullptr, const float* __restrict__ gk1 = nullptr, const float* __restrict__ gk0 = nullptr) {
    constexpr int kStage = kBK * kTT;                            // elements of one operand stage
    __shared__ __attribute__((aligned(16))) __bf16 Gs[2][kStage];
    __shared__ __attribute__((aligned(16))) __bf16 Zs[2][kStage];

    const int n_tiles = n_tiles_m * n_tiles_k;
    const int xcd = blockIdx.x & (kXcd - 1), local = blockIdx.x >> 3;
    const int tile = local % n_tiles;
    const int split = (local / n_tiles) * kXcd + xcd;
    if (split >= n_splits) return;
    const int tm0 = (tile / n_tiles_k) * kTT, tk0 = (tile % n_tiles_k) * kTT;
    const int r_begin = split * rows_per_split;
    const int r_end = min(n_rows, r_begin + rows_per_split);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;                     // 4 x 2 waves: 64 (m) x 128 (k) each
    const int l31 = lane & 31, lh = lane >> 5;
    // staging role: 32 threads per row (8 columns = 16 bytes each), 16 rows per pass, 2 passes per operand and stage
    const int sc8 = tid & 31, sr = tid >> 5;
    const int gcol = tm0 + sc8 * 8, zcol = tk0 + sc8 * 8;
    const bool g_on = gcol < M, z_on = zcol < K;                 // M, K % 8 == 0: a piece is all in or all out
    float pa[8], pb[8];
    if (ZPRO && z_on) {
        ld8f(pscale + zcol, pa);
        ld8f(pshift + zcol, pb);
    }
    float ca[GDUAL ? 8 : 1], cb[GDUAL ? 8 : 1], ck1[GDUAL ? 8 : 1], ck0[GDUAL ? 8 : 1];
    if constexpr (GDUAL) {
        if (g_on) {
            ld8f(ga + gcol, ca);
            ld8f(gb + gcol, cb);
            ld8f(gk1 + gcol, ck1);
            ld8f(gk0 + gcol, ck0);
        }
    }
    uint4 rg[2][2], rz[2][2], rg2[GDUAL ? 2 : 1][2];             // [slot][pass]
    // Round 4: the loads are UNCONDITIONAL (columns beyond M / K are clamped to valid ones and zeroed at the store).  As
    // `on ? load : 0` every load sat in an exec-masked block of its own: no interleaving with the MFMAs and, since the compiler
    // cannot count loads across a block that may be skipped, vmcnt(0) before the uses (the float32 wgrad had the same disease:
    // profiles/r04_tn_ablation.txt).
    const int gcol_c = min(gcol, M - 8), zcol_c = min(zcol, K - 8);
    auto load = [&](int sl, int r0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int64_t row = min(r0 + p * 16 + sr, r_end - 1);
            rg[sl][p] = ld8b(G + row * ldg + gcol_c);
            if constexpr (GDUAL) rg2[sl][p] = ld8b(G2 + row * ldg2 + gcol_c);
            rz[sl][p] = ld8b(Z + row * ldz + zcol_c);
        }
    };
    auto store = [&](int buf, int sl, int r0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = p * 16 + sr;
            uint4 g = rg[sl][p], z = rz[sl][p];
            if constexpr (GDUAL) {
                float x[8], y[8];
                bf_unpack8(g, x);
                bf_unpack8(rg2[sl][p], y);
#pragma unroll
                for (int e = 0; e < 8; ++e)                      // = BwdApplyF (bn.hip), element for element
                    x[e] = fmaf(ca[e], x[e] * lrelu_grad(fmaf(y[e], ca[e], cb[e]), slope), fmaf(ck1[e], y[e], ck0[e]));
                g = g_on ? bf_pack8(x) : make_uint4(0, 0, 0, 0);
            }
            if (r0 + r >= r_end || !g_on) g = make_uint4(0, 0, 0, 0);     // rows beyond the split / columns beyond M contribute nothing
            if (ZPRO) {
                float f[8];
                bf_unpack8(z, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = lrelu(fmaf(f[e], pa[e], pb[e]), slope);
                z = bf_pack8(f);
            }
            if (!z_on) z = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&Gs[buf][tn_off(r, sc8 * 8)]) = g;
            *reinterpret_cast<uint4*>(&Zs[buf][tn_off(r, sc8 * 8)]) = z;
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // transpose reads: lane L = lane & 15 of a 16-lane group supplies the address of 4 consecutive columns of row
    // (L >> 2) and receives column (lane & 31) of the tile, rows 0..3 of the 4-row block
    const int L = lane & 15, gi = (lane >> 4) & 1;
    const bool wave_on = (tm0 + wr * 64 < M) && (tk0 + wc * 128 < K);
    auto frag = [&](const __bf16* img, int cbase, int ks) -> bf16x8 {
        const int nb = ks * 16 + lh * 8;
        const int col = cbase + 16 * gi + 4 * (L & 3);
        typedef __attribute__((address_space(3))) bf16x4* lp;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(img + tn_off(nb + (L >> 2), col)));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(img + tn_off(nb + 4 + (L >> 2), col)));
        bf16x8 o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
        o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
        return o;
    };
    auto compute = [&](int buf, auto on) __attribute__((always_inline)) {
        if (!decltype(on)::value) return;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2], bf[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = frag(Gs[buf], wr * 64 + i * 32, ks);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = frag(Zs[buf], wc * 128 + j * 32, ks);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_b16(af[i], bf[j], acc[i][j]);
        }
    };

    const int ns = (r_end - r_begin + kBK - 1) / kBK;
    if (ns > 0) {
        load(0, r_begin);
        load(1, r_begin + kBK);
        store(0, 0, r_begin);
        load(0, r_begin + 2 * kBK);
    }
    lds_barrier();
    // iteration s (LDS buffer s & 1): slot (s+1) & 1 holds stage s+1; it is stored into the other buffer and refilled with
    // stage s+3.  The loop body is ONE basic block (pairs of iterations, no test inside; `wave_on` -- a wave without a tile in
    // a narrow panel -- chosen outside): MFMAs, conversions and loads interleave and the vmcnt counts are exact.
    auto run = [&](auto on) __attribute__((always_inline)) {
        int s = 0;
#pragma unroll 1
        for (; s + 1 < ns; s += 2) {
            const int r1 = r_begin + (s + 1) * kBK;
            compute(0, on);
            store(1, 1, r1);
            load(1, r1 + 2 * kBK);
            lds_barrier();
            compute(1, on);
            store(0, 0, r1 + kBK);
            load(0, r1 + 3 * kBK);
            lds_barrier();
        }
        if (s < ns) {
            compute(0, on);
            lds_barrier();
        }
    };
    if (wave_on) run(std::true_type());
    else run(std::false_type());

    if (!wave_on) return;
    float* o = out + (int64_t)split * split_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = tk0 + wc * 128 + j * 32 + l31;
        if (k >= K) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = tm0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) o[(int64_t)m * ld_out + k] = acc[i][j][r];
            }
    }
}


// The same wgrad with both operands as they are in HBM (no prologue: the agg-first layers, dW = dY^T . P, and every wgrad
// whose Z is already activated): no registers on the way -- G and Z stages go global -> LDS by DMA (2 rows x 512 bytes per
// wave instruction; the SOURCE 64-byte chunk is pre-swizzled by row & 3, so the lane-linear LDS image is the swizzled one
// the transpose reads expect), three buffers, copies two stages ahead, one counted-vmcnt barrier per stage.  Every wave
// issues the same four copies per stage whatever the shape (columns beyond M / K and rows beyond the split are CLAMPED to
// valid addresses: harmless duplicates), so the in-order VMEM counter can be counted; the one ragged stage of the last
// split zeroes its surplus G rows in LDS after they have landed.
__global__ __launch_bounds__(512) void gemm_tn_b16_dma_kernel(
    const bf16_t* __restrict__ G, int64_t ldg, const bf16_t* __restrict__ Z, int64_t ldz, float* __restrict__ out,
    int64_t ld_out, int64_t split_stride, int n_rows, int M, int K, int rows_per_split, int n_tiles_m, int n_tiles_k,
    int n_splits) {
    constexpr int kStage = kBK * kTT;                            // elements of one operand stage (16 KB)
    constexpr int NV = 4;                                        // copies per wave and stage
    __shared__ __attribute__((aligned(16))) __bf16 smem[3 * 2 * kStage];

    const int n_tiles = n_tiles_m * n_tiles_k;
    const int xcd = blockIdx.x & (kXcd - 1), local = blockIdx.x >> 3;
    const int tile = local % n_tiles;
    const int split = (local / n_tiles) * kXcd + xcd;
    if (split >= n_splits) return;
    const int tm0 = (tile / n_tiles_k) * kTT, tk0 = (tile % n_tiles_k) * kTT;
    const int r_begin = split * rows_per_split;
    const int r_end = min(n_rows, r_begin + rows_per_split);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    // copy role: instruction q = 8 i + wave covers stage rows 2q, 2q+1; lane -> (row 2q + (lane >> 5), 16-byte piece lane & 31)
    const int piece = lane & 31;
    int64_t goff[2], zoff[2];
    int crow[2];
    bool gon[2], zon[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 2 * (8 * i + wave) + (lane >> 5);
        const int col = ((((piece >> 2) ^ (r & 3)) << 2) | (piece & 3)) * 8;      // logical column of this lane's piece
        crow[i] = r;
        gon[i] = tm0 + col < M;                                  // narrow operands: lanes beyond the width copy nothing.  Every
        zon[i] = tk0 + col < K;                                  // instruction keeps lanes that do (columns 0..7 of its two rows),
        goff[i] = min(tm0 + col, M - 8);                         // so each wave still issues exactly four copies per stage
        zoff[i] = min(tk0 + col, K - 8);
    }
    // running source pointers: the stages are copied in increasing row order, kBK rows apart (no 64-bit multiply per copy);
    // rows in [r_end, n_rows) are the next split's -- valid memory, their G rows are zeroed in LDS by the ragged stage --
    // and only rows past the matrix end are clamped
    const bf16_t* gsrc[2];
    const bf16_t* zsrc[2];
    int next_r0 = r_begin;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        gsrc[i] = G + (int64_t)(r_begin + crow[i]) * ldg + goff[i];
        zsrc[i] = Z + (int64_t)(r_begin + crow[i]) * ldz + zoff[i];
    }
    auto copy = [&](int, __bf16* buf) {                          // stage rows next_r0 .. next_r0 + 31 -> buf (G image, then Z image)
        if (next_r0 + kBK <= n_rows) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (gon[i]) dma16(gsrc[i], buf + (8 * i + wave) * 512);
                if (zon[i]) dma16(zsrc[i], buf + kStage + (8 * i + wave) * 512);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t back = (int64_t)(next_r0 + crow[i]) - min(next_r0 + crow[i], n_rows - 1);
                if (gon[i]) dma16(gsrc[i] - back * ldg, buf + (8 * i + wave) * 512);
                if (zon[i]) dma16(zsrc[i] - back * ldz, buf + kStage + (8 * i + wave) * 512);
            }
        }
        next_r0 += kBK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            gsrc[i] += (int64_t)kBK * ldg;
            zsrc[i] += (int64_t)kBK * ldz;
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int L = lane & 15, gi = (lane >> 4) & 1;
    const bool wave_on = (tm0 + wr * 64 < M) && (tk0 + wc * 128 < K);
    auto frag = [&](const __bf16* img, int cbase, int ks) -> bf16x8 {
        const int nb = ks * 16 + lh * 8;
        const int col = cbase + 16 * gi + 4 * (L & 3);
        typedef __attribute__((address_space(3))) bf16x4* lp;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(img + tn_off(nb + (L >> 2), col)));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(img + tn_off(nb + 4 + (L >> 2), col)));
        bf16x8 o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
        o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
        return o;
    };
    auto compute = [&](const __bf16* buf) {
        if (!wave_on) return;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2], bf[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = frag(buf, wr * 64 + i * 32, ks);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = frag(buf + kStage, wc * 128 + j * 32, ks);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_b16(af[i], bf[j], acc[i][j]);
        }
    };
    const int ns = (r_end - r_begin + kBK - 1) / kBK;
    auto stage = [&](int s, __bf16* cur, __bf16* nxt2) {
        copy(r_begin + (s + 2) * kBK, nxt2);                     // (past the end: clamped re-loads, never consumed)
        const int nvalid = r_end - (r_begin + s * kBK);
        if (nvalid < kBK) {                                      // the ragged last stage: surplus G rows contribute nothing
            for (int id = tid; id < kBK * 32; id += 512) {
                const int r = id >> 5;
                if (r >= nvalid) *reinterpret_cast<uint4*>(cur + r * kTT + (id & 31) * 8) = make_uint4(0, 0, 0, 0);
            }
            lds_barrier();
        }
        compute(cur);
        wait_barrier<NV>();                                      // stage s+1 has landed; stage s+2's copies stay in flight
    };
    copy(r_begin, smem);
    copy(r_begin + kBK, smem + 2 * kStage);
    wait_barrier<NV>();                                          // stage 0 landed
    for (int s = 0; s < ns; s += 3) {
        stage(s, smem, smem + 4 * kStage);
        if (s + 1 < ns) stage(s + 1, smem + 2 * kStage, smem);
        if (s + 2 < ns) stage(s + 2, smem + 4 * kStage, smem + 2 * kStage);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no copy outlives the workgroup's LDS

    if (!wave_on) return;
    float* o = out + (int64_t)split * split_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = tk0 + wc * 128 + j * 32 + l31;
        if (k >= K) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = tm0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) o[(int64_t)m * ld_out + k] = acc[i][j][r];
            }
    }
}

// dW = sum over splits of the partial panels: see reduce_splits_kernel (gemm.hip) -- 64 float4 columns per workgroup, the
// splits dealt to its four waves, float64 partial sums combined through LDS in wave order
__global__ __launch_bounds__(256) void reduce_splits_b16_kernel(const float* __restrict__ part, int64_t split_stride,
                                                                int n_splits, float* __restrict__ dW, int64_t lddw, int M,
                                                                int K) {
    __shared__ double sm[3][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;                        // float4 index (K % 4 == 0)
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
        const int idx = 4 * q, m = idx / K, k = idx % K;
        float* o = dW + (int64_t)m * lddw + k;
        o[0] = (float)s0; o[1] = (float)s1; o[2] = (float)s2; o[3] = (float)s3;
    }
}

#include "fpartials.inc"
#include "gemm_rr_b16.inc"

struct TnPlanB {
    int n_tiles_m, n_tiles_k, n_splits, rows_per_split;
};
TnPlanB tn_plan_b16(int64_t n_rows, int M, int K) {
    TnPlanB p;
    p.n_tiles_m = (int)cdiv(M, kTT);
    p.n_tiles_k = (int)cdiv(K, kTT);
    const int tiles = p.n_tiles_m * p.n_tiles_k;
    // one 512-thread workgroup per CU (two where the tiles are few); >= 256 rows per split; multiple of 8 splits
    int64_t want = std::max<int64_t>(1, (2 * kCu) / tiles);
    want = std::min<int64_t>(want, cdiv(n_rows, 256));
    want = std::max<int64_t>(kXcd, (want / kXcd) * kXcd);
    int64_t rps = cdiv(cdiv(n_rows, want), kBK) * kBK;
    p.rows_per_split = (int)rps;
    p.n_splits = (int)cdiv(n_rows, rps);
    return p;
}

int device_cus_b16() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = kCu;
    }
    return cus;
}

int planes_mp(int MD) { return MD > 256 ? 512 : MD > 128 ? 256 : MD > 64 ? 128 : MD > 32 ? 64 : 32; }

template <int WR, int WC, int NJ>
void launch_rows(const bf16_t* A, int64_t lda, const __bf16* planes, bf16_t* Y, int64_t ldy, int64_t n_rows, int KD, int MD,
                 const float* bias, const float* ps, const float* psh, float slope, hipStream_t st) {
    constexpr int BMR = 64 * WR;
    const int n_tiles = (int)cdiv(n_rows, BMR);
    const int grid = std::min(n_tiles, device_cus_b16());
    if (ps)
        hipLaunchKernelGGL((gemm_rows_b16_kernel<WR, WC, NJ, 1>), dim3(grid), dim3(512), 0, st, A, lda, planes, Y, ldy,
                           (int)n_rows, KD, MD, bias, ps, psh, slope, n_tiles);
    else
        hipLaunchKernelGGL((gemm_rows_b16_kernel<WR, WC, NJ, 0>), dim3(grid), dim3(512), 0, st, A, lda, planes, Y, ldy,
                           (int)n_rows, KD, MD, bias, ps, psh, slope, n_tiles);
}

// DDMP_GEMM_RR=0 keeps the row-panel kernel (A/B comparisons); DDMP_RR_MIN_ROWS: the row threshold
bool rr_b16_enabled() {
    static int e = -1;
    if (e < 0) {
        const char* v = getenv("DDMP_GEMM_RR");
        e = (v && atoi(v) == 0) ? 0 : 1;
    }
    return e == 1;
}
int64_t rr_b16_min_rows() {
    static int64_t n = -1;
    if (n < 0) {
        const char* v = getenv("DDMP_RR_MIN_ROWS");
        n = (v && atoll(v) > 0) ? atoll(v) : 20000;
    }
    return n;
}
bool rr_b16_ok(int64_t n_rows, int KD, int MD, int64_t lda, int64_t lda2) {
    return rr_b16_enabled() && n_rows >= rr_b16_min_rows() && MD >= 128 && MD <= 512 && MD % 8 == 0 && KD % kBK == 0 && KD >= kBK &&
           KD <= kMaxK && n_rows * lda * 2 < ((int64_t)1 << 32) && n_rows * lda2 * 2 < ((int64_t)1 << 32);
}
size_t rr_b16_stats_bytes(int64_t n_rows) {
    return (size_t)(cdiv(n_rows, 64) + 8) * 2 * 512 * sizeof(double) + 256 + fpartials_mid_bytes(512);
}

// row-register kernel (gemm_rr_b16.inc).  BWD: the operand is the BatchNorm backward of (A = dZ, A2 = Yb); stats_ws / sums:
// BatchNorm statistics of the output from the epilogue
template <bool BWD>
int launch_rr_b16(const bf16_t* A, int64_t lda, const bf16_t* A2, int64_t lda2, const float* W, int64_t ldw, int transpose,
                  bf16_t* Y, int64_t ldy, int64_t n_rows, int KD, int MD, const float* bias, const float* ps, const float* psh,
                  const float* pc1, const float* pc0, float slope, void* ws, size_t ws_bytes, void* stats_ws,
                  size_t stats_ws_bytes, double* sums, hipStream_t st) {
    const int MPW = MD > kRBCols ? 512 : 256;
    const size_t need = (size_t)KD * MPW * sizeof(uint16_t);
    if (!ws || ws_bytes < need || !b16_aligned(ws)) return DDMP_EWORKSPACE;
    if (sums && (!stats_ws || stats_ws_bytes < rr_b16_stats_bytes(n_rows))) return DDMP_EWORKSPACE;
    __bf16* planes = (__bf16*)ws;
    hipLaunchKernelGGL(w_planes_b16_kernel, dim3((unsigned)std::min<int64_t>(cdiv((int64_t)KD * MPW, 256), 1024)), dim3(256), 0,
                       st, W, ldw, MD, KD, transpose, MPW, planes);
    LAUNCH_TRY();
    const int n_halves = MPW / kRBCols;
    const int tiles = (int)cdiv(n_rows, kRBRows);
    const int slots = std::max(8, std::min((tiles + 7) / 8 * 8, 2 * device_cus_b16() / n_halves / 8 * 8));   // (one round when they fit)
    dim3 grid((unsigned)(slots * n_halves)), block(256);
    double* stats = (double*)stats_ws;
#define DDMP_RRB(PM_, ST_, NB_)                                                                                        \
    hipLaunchKernelGGL((gemm_rr_b16_kernel<PM_, ST_, NB_>), grid, block, 0, st, A, lda, A2, lda2, planes, MPW, Y, ldy,  \
                       (int)n_rows, KD, MD, bias, ps, psh, pc1, pc0, slope, tiles, stats)
    if (BWD) DDMP_RRB(2, 0, 2);
    else if (sums && ps) DDMP_RRB(1, 1, 3);
    else if (sums) DDMP_RRB(0, 1, 3);
    else if (ps) DDMP_RRB(1, 0, 3);
    else DDMP_RRB(0, 0, 3);
#undef DDMP_RRB
    LAUNCH_TRY();
    if (sums) {
        const size_t pbytes = ((size_t)tiles * 2 * 2 * MPW * sizeof(double) + 255) / 256 * 256;
        fpartials_reduce(stats, tiles * 2, MPW, MD, (double*)((char*)stats + pbytes), sums, st);
        LAUNCH_TRY();
    }
    return DDMP_OK;
}

// Y[n, MD] = f(A[n, KD]) . B  with B[k][m] = transpose ? W[k][m] : W[m][k]
int gemm_rows_b16(const bf16_t* A, int64_t lda, const float* W, int64_t ldw, int transpose, bf16_t* Y, int64_t ldy,
                  int64_t n_rows, int KD, int MD, const float* bias, const float* ps, const float* psh, float slope,
                  void* ws, size_t ws_bytes, hipStream_t st) {
    ARG_TRY(A && W && Y && n_rows > 0 && n_rows < INT32_MAX && KD > 0 && MD > 0 && MD <= 512 && MD % 8 == 0 && KD % 8 == 0);
    ARG_TRY(lda >= KD && ldy >= MD && lda % 8 == 0 && ldy % 8 == 0 && b16_aligned(A) && b16_aligned(Y));
    ARG_TRY((ps == nullptr) == (psh == nullptr));
    if (KD < kBK) {                                              // first layer
        ARG_TRY(!transpose && !ps && (KD == 8 || KD == 16));
        const int grid = (int)std::min<int64_t>(cdiv(n_rows * (MD / 8), 256), 256 * 8);
        const size_t sh = (size_t)(MD * KD + MD) * sizeof(float);
        if (KD == 8)
            hipLaunchKernelGGL((gemm_rows_smallk_b16_kernel<8>), dim3(grid), dim3(256), sh, st, A, lda, W, ldw, Y, ldy, (int)n_rows, MD, bias);
        else
            hipLaunchKernelGGL((gemm_rows_smallk_b16_kernel<16>), dim3(grid), dim3(256), sh, st, A, lda, W, ldw, Y, ldy, (int)n_rows, MD, bias);
        LAUNCH_TRY();
        return DDMP_OK;
    }
    ARG_TRY(KD % kBK == 0 && (!ps || KD <= kMaxK));
    // plain products: the row-register kernel where it measured faster than the row-panel kernel (1M-face step, round 3:
    // K = 512 -> 256 | 512 columns 288 vs 320 us and 261 vs 271 us; shorter contractions lose to its heavier epilogue)
    if (rr_b16_ok(n_rows, KD, MD, lda, 0) && KD >= 512 && MD > 128)
        return launch_rr_b16<false>(A, lda, nullptr, 0, W, ldw, transpose, Y, ldy, n_rows, KD, MD, bias, ps, psh, nullptr, nullptr,
                                    slope, ws, ws_bytes, nullptr, 0, nullptr, st);
    const int MP = planes_mp(MD);
    const size_t need = (size_t)KD * MP * sizeof(uint16_t);
    if (!ws || ws_bytes < need || !b16_aligned(ws)) return DDMP_EWORKSPACE;
    __bf16* planes = (__bf16*)ws;
    hipLaunchKernelGGL(w_planes_b16_kernel, dim3((unsigned)std::min<int64_t>(cdiv((int64_t)KD * MP, 256), 1024)), dim3(256), 0,
                       st, W, ldw, MD, KD, transpose, MP, planes);
    LAUNCH_TRY();
    if (MP == 512) launch_rows<2, 4, 4>(A, lda, planes, Y, ldy, n_rows, KD, MD, bias, ps, psh, slope, st);
    else if (MP == 256) launch_rows<4, 2, 4>(A, lda, planes, Y, ldy, n_rows, KD, MD, bias, ps, psh, slope, st);
    else if (MP == 128) launch_rows<4, 2, 2>(A, lda, planes, Y, ldy, n_rows, KD, MD, bias, ps, psh, slope, st);
    else if (MP == 64) launch_rows<4, 2, 1>(A, lda, planes, Y, ldy, n_rows, KD, MD, bias, ps, psh, slope, st);
    else launch_rows<8, 1, 1>(A, lda, planes, Y, ldy, n_rows, KD, MD, bias, ps, psh, slope, st);
    LAUNCH_TRY();
    return DDMP_OK;
}

}  // namespace

extern "C" size_t ddmp_gemm_rows_bf16_workspace_bytes(int K, int M) {
    if (K <= 0 || M <= 0) return 0;
    const int kp = (int)cdiv(K, kBK) * kBK, mp = (int)cdiv(M, kBK) * kBK;
    return (size_t)std::max(kp, mp) * 512 * sizeof(uint16_t);
}

extern "C" int ddmp_gemm_nt_bf16(const uint16_t* A, int64_t lda, const float* W, int64_t ldw, uint16_t* Y, int64_t ldy,
                                 int64_t n_rows, int K, int M, const float* bias, const float* pro_scale,
                                 const float* pro_shift, float slope, void* workspace, size_t workspace_bytes,
                                 ddmp_stream stream) {
    ARG_TRY(ldw >= K);
    return gemm_rows_b16(A, lda, W, ldw, 0, Y, ldy, n_rows, K, M, bias, pro_scale, pro_shift, slope, workspace,
                         workspace_bytes, (hipStream_t)stream);
}

extern "C" int ddmp_gemm_nn_bf16(const uint16_t* A, int64_t lda, const float* W, int64_t ldw, uint16_t* Y, int64_t ldy,
                                 int64_t n_rows, int M, int K, void* workspace, size_t workspace_bytes,
                                 ddmp_stream stream) {
    ARG_TRY(ldw >= K);
    return gemm_rows_b16(A, lda, W, ldw, 1, Y, ldy, n_rows, M, K, nullptr, nullptr, nullptr, 0.f, workspace,
                         workspace_bytes, (hipStream_t)stream);
}

extern "C" size_t ddmp_gemm_nt_stats_bf16_workspace_bytes(int64_t n_rows, int M) {
    if (n_rows <= 0 || M <= 0) return 0;
    return rr_b16_stats_bytes(n_rows);
}

// 1 = ddmp_gemm_nt_stats_bf16 takes its statistics from the GEMM epilogue and the two *_bnbwd_bf16 forms exist for this
// shape (row-register kernel); 0 = the caller runs the separate passes (bn_stats / bn_bwd_apply)
extern "C" int ddmp_gemm_fused_bf16_supported(int cout, int cin, int64_t n_rows) {
    // bit 0: ddmp_gemm_nt_stats_bf16 (measured against GEMM + bn_stats at 1M faces: 128|256 -> 256, 256 -> 512: 35-38 us
    //        per launch faster; 512 -> 512: a tie)
    // bit 1: the two *_bnbwd_bf16 forms (against bn_bwd_apply + the plain GEMMs: 512 <- 256: -157 us, 256 <- 256: -75 us,
    //        256 <- 128: -52 us per layer; 512 <- 512: +100 us -- the register-staged wgrad is 316 us slower than the all-DMA
    //        one it replaces -- so that shape keeps the separate pass)
    int r = 0;
    if (rr_b16_ok(n_rows, cin, cout, cin, 0) && cout > 128) r |= 1;
    if (rr_b16_ok(n_rows, cout, cin, cout, cout) && !(cout >= 512 && cin >= 512)) r |= 2;
    return r;
}

extern "C" int ddmp_gemm_nt_stats_bf16(const uint16_t* A, int64_t lda, const float* W, int64_t ldw, uint16_t* Y, int64_t ldy,
                                       int64_t n_rows, int K, int M, const float* bias, const float* pro_scale,
                                       const float* pro_shift, float slope, double* sums2, void* workspace,
                                       size_t workspace_bytes, void* stats_ws, size_t stats_ws_bytes, ddmp_stream stream) {
    ddmp::FinalizeScope fin_scope(sums2, stream, M);
    ARG_TRY(A && W && Y && sums2 && n_rows > 0 && n_rows < INT32_MAX && K > 0 && M > 0 && ldw >= K);
    ARG_TRY(lda >= K && ldy >= M && lda % 8 == 0 && ldy % 8 == 0 && b16_aligned(A) && b16_aligned(Y));
    ARG_TRY((pro_scale == nullptr) == (pro_shift == nullptr));
    if (!rr_b16_ok(n_rows, K, M, lda, 0)) return DDMP_EINVAL;
    return launch_rr_b16<false>(A, lda, nullptr, 0, W, ldw, 0, Y, ldy, n_rows, K, M, bias, pro_scale, pro_shift, nullptr, nullptr,
                                slope, workspace, workspace_bytes, stats_ws, stats_ws_bytes, sums2, (hipStream_t)stream);
}

// out[n, K] = bf16(dY) . W[M, K] with dY = BatchNorm+LeakyReLU backward of (dZ, Yb) rebuilt on the operand load
extern "C" int ddmp_gemm_nn_bnbwd_bf16(const uint16_t* dZ, int64_t lddz, const uint16_t* Yb, int64_t ldyb, const float* W,
                                       int64_t ldw, uint16_t* out, int64_t ld_out, int64_t n_rows, int M, int K,
                                       const float* a, const float* b, const float* c1, const float* c0, float slope,
                                       void* workspace, size_t workspace_bytes, ddmp_stream stream) {
    ARG_TRY(dZ && Yb && W && out && a && b && c1 && c0 && n_rows > 0 && n_rows < INT32_MAX && M > 0 && K > 0 && ldw >= K);
    ARG_TRY(lddz >= M && ldyb >= M && ld_out >= K && lddz % 8 == 0 && ldyb % 8 == 0 && ld_out % 8 == 0);
    ARG_TRY(b16_aligned(dZ) && b16_aligned(Yb) && b16_aligned(out) && b16_aligned(a) && b16_aligned(b) && b16_aligned(c1) && b16_aligned(c0));
    if (!rr_b16_ok(n_rows, M, K, lddz, ldyb)) return DDMP_EINVAL;
    return launch_rr_b16<true>(dZ, lddz, Yb, ldyb, W, ldw, 1, out, ld_out, n_rows, M, K, nullptr, a, b, c1, c0, slope, workspace,
                               workspace_bytes, nullptr, 0, nullptr, (hipStream_t)stream);
}

// dW[M, K] = bf16(dY)^T . f(Z) with dY as above
extern "C" int ddmp_gemm_tn_bnbwd_bf16(const uint16_t* dZ, int64_t lddz, const uint16_t* Yb, int64_t ldyb, const uint16_t* Z,
                                       int64_t ldz, float* dW, int64_t lddw, int64_t n_rows, int M, int K, const float* a,
                                       const float* b, const float* c1, const float* c0, const float* pro_scale,
                                       const float* pro_shift, float slope, void* workspace, size_t workspace_bytes,
                                       ddmp_stream stream) {
    ARG_TRY(dZ && Yb && Z && dW && a && b && c1 && c0 && n_rows > 0 && n_rows < INT32_MAX && M > 0 && K > 0 && M % 8 == 0 && K % 8 == 0);
    ARG_TRY(lddz >= M && ldyb >= M && ldz >= K && lddw >= K && lddz % 8 == 0 && ldyb % 8 == 0 && ldz % 8 == 0);
    ARG_TRY(b16_aligned(dZ) && b16_aligned(Yb) && b16_aligned(Z) && b16_aligned(a) && b16_aligned(b) && b16_aligned(c1) && b16_aligned(c0));
    ARG_TRY((pro_scale == nullptr) == (pro_shift == nullptr));
    ARG_TRY(!pro_scale || (b16_aligned(pro_scale) && b16_aligned(pro_shift)));
    hipStream_t st = (hipStream_t)stream;
    const TnPlanB p = tn_plan_b16(n_rows, M, K);
    const size_t need = (size_t)p.n_splits * (size_t)M * (size_t)K * sizeof(float);
    if (!workspace || workspace_bytes < need || !b16_aligned(workspace)) return DDMP_EWORKSPACE;
    float* part = (float*)workspace;
    const int64_t sstride = (int64_t)M * K;
    const int n_tiles = p.n_tiles_m * p.n_tiles_k;
    dim3 grid((unsigned)(cdiv(p.n_splits, kXcd) * kXcd * n_tiles)), block(512);
    if (pro_scale)
        hipLaunchKernelGGL((gemm_tn_b16_kernel<true, true>), grid, block, 0, st, dZ, lddz, Z, ldz, part, (int64_t)K, sstride,
                           (int)n_rows, M, K, p.rows_per_split, p.n_tiles_m, p.n_tiles_k, p.n_splits, pro_scale, pro_shift, slope,
                           Yb, ldyb, a, b, c1, c0);
    else
        hipLaunchKernelGGL((gemm_tn_b16_kernel<false, true>), grid, block, 0, st, dZ, lddz, Z, ldz, part, (int64_t)K, sstride,
                           (int)n_rows, M, K, p.rows_per_split, p.n_tiles_m, p.n_tiles_k, p.n_splits, pro_scale, pro_shift, slope,
                           Yb, ldyb, a, b, c1, c0);
    LAUNCH_TRY();
    hipLaunchKernelGGL(reduce_splits_b16_kernel, dim3((unsigned)cdiv((int64_t)M * K, 256)), dim3(256), 0, st, part, sstride,
                       p.n_splits, dW, lddw, M, K);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" size_t ddmp_gemm_tn_bf16_workspace_bytes(int64_t n_rows, int M, int K) {
    if (n_rows <= 0 || M <= 0 || K <= 0) return 0;
    const TnPlanB p = tn_plan_b16(n_rows, M, K);
    return (size_t)p.n_splits * (size_t)M * (size_t)K * sizeof(float);
}

extern "C" int ddmp_gemm_tn_bf16(const uint16_t* G, int64_t ldg, const uint16_t* Z, int64_t ldz, float* dW, int64_t lddw,
                                 int64_t n_rows, int M, int K, const float* pro_scale, const float* pro_shift,
                                 float slope, void* workspace, size_t workspace_bytes, ddmp_stream stream) {
    ARG_TRY(G && Z && dW && n_rows > 0 && n_rows < INT32_MAX && M > 0 && K > 0 && M % 8 == 0 && K % 8 == 0);
    ARG_TRY(ldg >= M && ldz >= K && lddw >= K && ldg % 8 == 0 && ldz % 8 == 0 && b16_aligned(G) && b16_aligned(Z));
    ARG_TRY((pro_scale == nullptr) == (pro_shift == nullptr));
    ARG_TRY(!pro_scale || (b16_aligned(pro_scale) && b16_aligned(pro_shift)));
    hipStream_t st = (hipStream_t)stream;
    const TnPlanB p = tn_plan_b16(n_rows, M, K);
    const size_t need = (size_t)p.n_splits * (size_t)M * (size_t)K * sizeof(float);
    if (!workspace || workspace_bytes < need || !b16_aligned(workspace)) return DDMP_EWORKSPACE;
    float* part = (float*)workspace;
    const int64_t sstride = (int64_t)M * K;
    const int n_tiles = p.n_tiles_m * p.n_tiles_k;
    dim3 grid((unsigned)(cdiv(p.n_splits, kXcd) * kXcd * n_tiles)), block(512);
    constexpr int tn_dma = 1;                                    // (0: the register-staged kernel for every wgrad -- an A/B of round 3)
    if (pro_scale)
        hipLaunchKernelGGL((gemm_tn_b16_kernel<true>), grid, block, 0, st, G, ldg, Z, ldz, part, (int64_t)K, sstride,
                           (int)n_rows, M, K, p.rows_per_split, p.n_tiles_m, p.n_tiles_k, p.n_splits, pro_scale, pro_shift, slope);
    else if (tn_dma)
        hipLaunchKernelGGL(gemm_tn_b16_dma_kernel, grid, block, 0, st, G, ldg, Z, ldz, part, (int64_t)K, sstride, (int)n_rows, M,
                           K, p.rows_per_split, p.n_tiles_m, p.n_tiles_k, p.n_splits);
    else
        hipLaunchKernelGGL((gemm_tn_b16_kernel<false>), grid, block, 0, st, G, ldg, Z, ldz, part, (int64_t)K, sstride,
                           (int)n_rows, M, K, p.rows_per_split, p.n_tiles_m, p.n_tiles_k, p.n_splits, pro_scale, pro_shift, slope);
    LAUNCH_TRY();
    hipLaunchKernelGGL(reduce_splits_b16_kernel, dim3((unsigned)cdiv((int64_t)M * K, 256)), dim3(256), 0, st, part, sstride,
                       p.n_splits, dW, lddw, M, K);
    LAUNCH_TRY();
    return DDMP_OK;
}
