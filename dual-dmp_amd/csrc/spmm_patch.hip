// GCN aggregation, LDS-patch form (float32 and bfloat16 features):  Y[i,:] = dinv[i] * sum_e dinv[col e] * f(X[col e,:]) (+ bias)
//
// Why a second SpMM kernel.  The slab kernel (spmm.hip / spmm_b16.hip) gathers every CSR entry through the vector L1.
// Measured on it: time = HBM streaming time + (entries per row) x (a fixed cost per gathered 128-byte
// line); the two do not overlap, and the vertex graph (7 entries per row) sits at 41-44 % of 8 TB/s where the face graph
// (4 entries) reaches 57-59 %.  The L1 returns data in request order, so the 6 of 7 gathers that HIT occupy the window
// of requests in flight while only the one that misses moves HBM bytes: the achievable miss-level parallelism is the
// window divided by the entries per row.
// Here the two kinds of traffic use different pipes:
//   * the DISTINCT rows a 64-row chunk references (its "patch": 100-120 rows on a Morton-ordered mesh, 1.6-1.9 per
//     output row instead of 4-7; tables built once per graph, graph.hip) are copied global -> LDS by
//     global_load_lds_dwordx4, one 128-byte slab of 8 rows per wave instruction, no VGPRs: every request in the L1 window
//     is a row fetch;
//   * the gathers read the patch from LDS (ds_read_b128, 256 B/clk/CU) through the entries' patch-local indices.
// What the first attempt at this (spmm_patch_dma_kernel, round 1: 2 buffers, a fifth wave copying one slab ahead, a
// vmcnt(0) + barrier per slab) got wrong was depth: one 14 KB slab in flight per workgroup.  Here every wave issues its
// share of the copies NB-1 slabs ahead (NB = 4 buffers), one barrier per slab, and the wait before it is a COUNTED
// vmcnt that leaves the younger slabs' copies and the output stores in flight.  Nothing in the slab loop returns data to
// a VGPR from memory: coefficients, dinv, CSR slice, patch list (and the Yp rows of the fused reduction) are in LDS, so
// the in-order VMEM counter carries only copies and stores and can be counted exactly.
#include "b16_common.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace {

using namespace ddmp;

constexpr int kRB = 64;            // rows per chunk (= ddmp::kChunkRows: the patch tables are per 64 rows)
constexpr int kMaxE = kRB * 16;    // CSR entries per chunk

template <typename T> struct Lane;                                  // one lane = 16 bytes of a row
template <> struct Lane<float> {
    static constexpr int VW = 4;
    static __device__ __forceinline__ void unpack(uint4 u, float (&f)[4]) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
    static __device__ __forceinline__ uint4 pack(const float (&f)[4]) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Lane<bf16_t> {
    static constexpr int VW = 8;
    static __device__ __forceinline__ void unpack(uint4 u, float (&f)[8]) { bf_unpack8(u, f); }
    static __device__ __forceinline__ uint4 pack(const float (&f)[8]) { return bf_pack8(f); }
};

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

struct RedArgs {                    // fused BatchNorm-backward column reductions of the output (see BnRed in spmm.hip)
    const void* Yp;
    int64_t ldyp;
    const float *scale, *shift, *mean, *rstd;
    float* part;
};

struct BwdArgs {                    // BatchNorm backward on the gather (see BnBwdGather in spmm.hip): the operand is rebuilt from (X = dZ, Yb)
    const void* Yb;
    int64_t ldyb;
    const float *c1, *c0;
};

// KD: copies per wave and slab of the patch (8 rows each, 4 waves): the patch buffer holds 32 KD rows
// NB: patch buffers (2: 41-57 KB of LDS per workgroup, two to three workgroups per CU hide each other's copy latency and
// barriers; 4: the round-2 form, one workgroup per CU pipelining three slabs ahead)
// NE: 0 = entries read from LDS per slab | 4, 8 = a lane keeps the patch offsets and weights of the FIRST NE entries of its two
// rows in REGISTERS for all slabs of the chunk (round 4: the index and weight reads were two of the three LDS instructions per
// gathered row; without them the 7-entry vertex graph went from 547 to 447 us at C = 512 in a timing-only build), the gather
// of a slab is then NE ds_read_b128 with immediate buffer offsets per row.  Round 5: rows LONGER than NE entries (irregular
// meshes: valence 3 ... 12+) continue from the LDS lists -- the same sums in the same order --, behind a wave-uniform test that
// a wave of short rows never enters; chunks the kernel cannot take at all (ddmp_graph: "heavy") return at once and are
// computed by the lean gather.  Rows without entries (csr_host graphs) are legal: weight 0 on patch row 0.
// RED: 0 | 1 the BatchNorm-backward reductions of the output (RedArgs; the chunk's own Yp rows ride the DMA ring) | 2 (round 5)
// the BatchNorm STATISTICS of the output around the per-column reference red.mean (see spmm_lean.inc): sum (y - ref), (y - ref)^2
// BWD (round 6; measured first in round 5, experiments/r05): the gathered operand is the BatchNorm + LeakyReLU BACKWARD of (X = dZ, Yb),
// rebuilt per entry as in spmm_lean_kernel<true, 0, true> (same expression, same order: bit-identical sums) -- the patch rows of Yb
// ride the DMA ring in a second half of every buffer (twice the copies, twice the LDS reads, no second trip through the L1 per
// entry); a, b = pscale, pshift.  No fused epilogue.
template <typename T, int KD, bool PRO, int RED, int NB, int NE, bool BWD = false>
__global__ __launch_bounds__(256) void spmm_patch2_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const unsigned short* __restrict__ lcol,
    const float* __restrict__ ew, const float* __restrict__ dinv, const float* __restrict__ dinv_r, const int* __restrict__ pl_ptr, const int* __restrict__ pl_col, const int* __restrict__ pl_split,
    const T* __restrict__ X, int64_t ldx, T* __restrict__ Y, int64_t ldy, int n_rows, int C,
    const float* __restrict__ bias, const float* __restrict__ pscale, const float* __restrict__ pshift, float slope,
    int chunks_per_xcd, int n_chunks, int emax, RedArgs red, BwdArgs bw = BwdArgs()) {
    static_assert(!BWD || (PRO && !RED), "BWD: coefficients a, b as the prologue's, no epilogue");
    constexpr int VW = Lane<T>::VW, CS = 8 * VW;                 // channels per 128-byte slab
    constexpr int PR = 32 * KD;                                  // patch rows per buffer
    constexpr int KX = BWD ? 2 * KD : KD;                        // patch copies per wave and slab (BWD: X and Yb)
    constexpr int KR = RED == 1 ? 2 : 0;                         // copies per wave and slab of the chunk's own Yp rows
    constexpr int NST = 2;                                       // output stores per lane and slab
    constexpr int NSR = RED ? 1 : 0;                             // + the wave's quarter of the previous slab's partial record
    constexpr int kBuf = PR * 128 * (BWD ? 2 : 1) + (RED == 1 ? kRB * 128 : 0);  // bytes per buffer
    constexpr int kNB = NB;
    constexpr int NWAIT = (KX + KR) * (kNB - 2) + (NST + NSR) * (kNB - 1);   // VMEM operations younger than the copies of slab s
    static_assert(NWAIT <= 63, "vmcnt range");
    static_assert(!RED || NB == 2, "the fused reduction's record stores are counted for two buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* bufs = smem;                                  // [kNB][kBuf]
    float* s_w = reinterpret_cast<float*>(smem + kNB * kBuf);    // [emax]: the most entries of a chunk of this graph (multiple of 64, <= kMaxE)
    float* s_dinv = s_w + emax;                                  // [kRB]
    int* s_rowptr = reinterpret_cast<int*>(s_dinv + kRB);        // [kRB + 1] (+3 pad)
    unsigned short* s_lc = reinterpret_cast<unsigned short*>(s_rowptr + kRB + 4);      // [emax]
    int* s_perm = reinterpret_cast<int*>(s_lc + emax);          // [kRB]: the chunk's rows, longest first (chunk_rank_desc)
    float* s_coef = reinterpret_cast<float*>(s_perm + kRB);      // [nco][C]: bias | pscale, pshift | scale, shift, mean, rstd (RED 2: ref)
    // RED: the four waves' partial sums of a slab, by slab parity: [2][4 waves][2 sums][CS]
    float* s_part = s_coef + (RED == 1 ? 7 : RED == 2 ? 4 : BWD ? 5 : PRO ? 3 : 1) * C;

    const int chunk = (blockIdx.x & (kXcd - 1)) * chunks_per_xcd + (blockIdx.x >> 3);
    if (chunk >= n_chunks) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = lane >> 3, sl = lane & 7;

    // Prologue (round 6): ONE round trip to the tables instead of four dependent ones.  The spans (patch list, CSR slice) come from
    // uniform loads; the lane's KD patch rows go to REGISTERS first in program order -- the copies of slab 0 need nothing else and
    // are issued as soon as those arrive --, every table of the chunk (row pointers, factors, the entries' patch indices and their
    // weights `ew` = dinv[col[e]], precomputed per graph: no col -> dinv chain) is requested before that wait and lands in LDS
    // while slab 0 travels.  (Before: row pointers + patch list -> barrier -> col -> dinv[col] -> barrier -> first copy.)
    const int pl0 = pl_ptr[chunk], pl1 = pl_ptr[chunk + 1];
    if (pl1 <= pl0) return;                                      // a heavy chunk: the lean gather's (uniform: before any barrier)
    // a SPLIT chunk (ddmp_graph::pl_split: its patch is too large, the patches of its 2 halves / 4 quarters fit) is walked in as many
    // passes, each a chunk of 32 / 16 rows with its own patch, entry indices and -- fused sums -- record slot
    int nh = 1, first = 0;
    if (pl_split) {
        first = pl_split[2 * chunk];
        nh = max(1, pl_split[2 * chunk + 1]);
    }
    const int rp = kRB / nh;
    const int rows_c = min(kRB, n_rows - chunk * kRB);
    for (int h = 0; h < nh; ++h) {
    const int r0 = chunk * kRB + rp * h;
    const int nr = min(rp, rows_c - rp * h);                     // (> 0: the table builder splits only chunks whose every part has rows)
    const int p0 = h ? pl_split[first + 2 * (h - 1)] : pl0;
    const int np = (h + 1 < nh ? pl_split[first + 2 * h] : pl1) - p0;
    const int rec = h ? pl_split[first + 2 * (h - 1) + 1] : chunk;
    const int e0 = rowptr[r0];
    const int ne = rowptr[r0 + nr] - e0;                         // <= kMaxE (graph.hip: a chunk with more entries is heavy)
    const T* xlane = X + sl * VW;
    const T* psrc[KD];                                           // the lane's patch rows (padded with the last row), + its 16 bytes
    const T* pysrc[BWD ? KD : 1];                                // BWD: the same rows of Yb
#pragma unroll
    for (int i = 0; i < KD; ++i) {
        const int prow = pl_col[p0 + min((4 * i + wave) * 8 + grp, np - 1)];
        psrc[i] = xlane + (int64_t)prow * ldx;
        if (BWD) pysrc[BWD ? i : 0] = static_cast<const T*>(bw.Yb) + sl * VW + (int64_t)prow * bw.ldyb;
    }
    const int rp_t = tid <= nr ? rowptr[r0 + tid] : 0;
    const int rp_n = (wave == 0 && lane < nr) ? rowptr[r0 + lane + 1] : 0;
    const float dv_t = tid < nr ? dinv_r[r0 + tid] : 0.f;
    unsigned short lc_t[4];
    float w_t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = tid + 256 * k;
        lc_t[k] = t < ne ? lcol[e0 + t] : (unsigned short)0;
        w_t[k] = t < ne ? ew[e0 + t] : 0.f;
    }
    float co_t[4][RED == 1 ? 7 : RED == 2 ? 4 : BWD ? 5 : PRO ? 3 : 1];      // coefficient columns tid, + 256, ... (C <= 1024)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k;
        const bool in = i < C;
        co_t[k][0] = (in && bias) ? bias[i] : 0.f;
        if (PRO) {
            co_t[k][1] = in ? pscale[i] : 0.f;
            co_t[k][2] = in ? pshift[i] : 0.f;
        }
        if (RED == 1) {
            co_t[k][3] = in ? red.scale[i] : 0.f;
            co_t[k][4] = in ? red.shift[i] : 0.f;
            co_t[k][5] = in ? red.mean[i] : 0.f;
            co_t[k][6] = in ? red.rstd[i] : 0.f;
        }
        if (RED == 2) co_t[k][3] = in ? red.mean[i] : 0.f;
        if (BWD) {
            co_t[k][BWD ? 3 : 0] = in ? bw.c1[i] : 0.f;
            co_t[k][BWD ? 4 : 0] = in ? bw.c0[i] : 0.f;
        }
    }
    const int n_slabs = C / CS;
    const T* yplane = RED == 1 ? static_cast<const T*>(red.Yp) + sl * VW : nullptr;
    // copies of slab s into buffer b: wave w, instruction i covers patch rows (4 i + w) * 8 .. + 7
    auto copy = [&](int s, unsigned char* dst) {
#pragma unroll
        for (int i = 0; i < KD; ++i) {
            dma16(psrc[i] + s * CS, dst + (4 * i + wave) * 8 * 128);
            if (BWD) dma16(pysrc[BWD ? i : 0] + s * CS, dst + PR * 128 + (4 * i + wave) * 8 * 128);
        }
        if (RED == 1) {
#pragma unroll
            for (int i = 0; i < KR; ++i) {
                const int j0 = (4 * i + wave) * 8;                // own rows j0 .. j0 + 7 of the chunk
                const int row = r0 + min(j0 + grp, nr - 1);
                dma16(yplane + (int64_t)row * red.ldyp + s * CS, dst + PR * 128 + j0 * 128);
            }
        }
    };
    // The slab loop is unrolled over the buffers with COMPILE-TIME buffer addresses: the compiler orders a ds_read
    // behind every LDS-DMA it cannot prove disjoint (a run-time buffer index costs a vmcnt(0) -- a full drain -- before
    // the first read of every slab; measured 2x on the whole kernel).
#pragma unroll
    for (int i = 0; i < kNB - 1; ++i)
        if (i < n_slabs) copy(i, bufs + i * kBuf);
    if (tid <= nr) s_rowptr[tid] = rp_t;
    if (tid < nr) s_dinv[tid] = dv_t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = tid + 256 * k;
        if (t < ne) {
            s_lc[t] = lc_t[k];
            s_w[t] = w_t[k];
        }
        if (t < C) {
            s_coef[t] = co_t[k][0];
            if (PRO) {
                s_coef[C + t] = co_t[k][1];
                s_coef[2 * C + t] = co_t[k][2];
            }
            if (RED == 1) {
                s_coef[3 * C + t] = co_t[k][3];
                s_coef[4 * C + t] = co_t[k][4];
                s_coef[5 * C + t] = co_t[k][5];
                s_coef[6 * C + t] = co_t[k][6];
            }
            if (RED == 2) s_coef[3 * C + t] = co_t[k][3];
            if (BWD) {
                s_coef[3 * C + t] = co_t[k][BWD ? 3 : 0];
                s_coef[4 * C + t] = co_t[k][BWD ? 4 : 0];
            }
        }
    }
    if (wave == 0) {                                             // rows of similar length share a wave (see chunk_rank_desc)
        const int nn_l = lane < nr ? rp_n - rp_t : -4;
        s_perm[chunk_rank_desc(min((nn_l + 3) >> 2, 31), lane)] = lane;
    }
    __syncthreads();                                             // (the LDS tables; the copies above are waited for in the slab loop)
    int rows[2];                                                 // this lane's two rows (local indices; clamped on a ragged chunk)
#pragma unroll
    for (int q = 0; q < 2; ++q) rows[q] = s_perm[min(wave * 16 + grp + 8 * q, nr - 1)];

    // NE > 0: this lane's rows (wave * 8 + grp, + 32) -- byte offset of the entry's patch row (+ the lane's 16 bytes) and weight;
    // a row's missing entries repeat its last one with weight 0 (no further distinct LDS row, the sum unchanged)
    constexpr int NEc = NE > 0 ? NE : 1;
    int lo[2][NEc];
    float wjr[2][NEc];
    int tail_es[2] = {0, 0}, tail_ee[2] = {0, 0};                // NE > 0: the entries behind the first NE of the lane's rows
    int nmax = 0;                                                // most entries of a row of this WAVE (uniform)
    if (NE > 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pos = wave * 16 + grp + 8 * q;             // (positions 16 w .. 16 w + 15 of the ordered rows: wave w)
            const int lrc = s_perm[min(pos, nr - 1)];
            const int es = s_rowptr[lrc] - e0;
            const int nn = pos < nr ? s_rowptr[lrc + 1] - s_rowptr[lrc] : 0;
            nmax = max(nmax, nn);
            tail_es[q] = es + min(nn, NEc);
            tail_ee[q] = es + nn;
#pragma unroll
            for (int k = 0; k < NEc; ++k) {
                const int ek = es + min(k, nn - 1);              // missing entries: the row's last one again, weight 0
                lo[q][k] = (nn > 0 ? (int)s_lc[ek] * 128 : 0) + sl * 16;     // (an empty row: patch row 0, weight 0)
                wjr[q][k] = k < nn ? s_w[ek] : 0.f;
            }
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
        nmax = __builtin_amdgcn_readfirstlane(nmax);
    }

    const bool full = nr == kRB;                                 // a ragged last chunk stores less: wait for everything

    // RED: ONE partial record per channel and chunk (as the lean gather writes them): the four waves' sums of slab s meet in
    // LDS, every wave adds them up for a quarter of the slab's 2 CS values (fixed order, float64, rounded once) and stores
    // that quarter -- one VMEM store per wave and slab, counted in the waits above
    auto record = [&](int s) {
        constexpr int Q = 2 * CS / 4;                            // values per wave
        const int i = wave * Q + (lane % Q);                     // (lanes >= Q repeat a value; only the first Q store)
        const int which = i / CS, c = i % CS;
        const float* sp = s_part + ((s & 1) * 4 * 2) * CS + which * CS + c;
        const double t = ((double)sp[0] + (double)sp[2 * CS]) + ((double)sp[4 * CS] + (double)sp[6 * CS]);
        if (lane < Q) red.part[((int64_t)rec * 2 + which) * C + s * CS + c] = (float)t;
    };
    auto slab = [&](int s, auto bc) {
        constexpr int B = decltype(bc)::value;
        // the copies of slab s have landed (mine: counted wait; everybody's: barrier); all waves are done with slab s-1
        // (counted: the kNB-2 younger slabs' copies and the stores issued since stay in flight; the first kNB-1 slabs
        //  have fewer stores behind them, the last ones fewer copies: exact counts or a full drain)
        constexpr int ND = (KX + KR) * (kNB - 2);
        if (!full || s + kNB - 1 > n_slabs) wait_vm_barrier<0>();
        else if (s == 0) wait_vm_barrier<ND>();
        else if (s == 1 && (kNB > 2 || RED)) wait_vm_barrier<ND + NST>();    // (RED: slab 0 had no record to store)
        else if (s == 2 && kNB > 3) wait_vm_barrier<ND + 2 * NST>();
        else wait_vm_barrier<NWAIT>();
        if (s + kNB - 1 < n_slabs) copy(s + kNB - 1, bufs + ((B + kNB - 1) % kNB) * kBuf);
        if (RED && s > 0) record(s - 1);                         // (every wave's partials of slab s-1 are in LDS: the barrier above)
        const unsigned char* pb = bufs + B * kBuf + sl * 16;
        const int off = s * CS + sl * VW;
        float pa[VW], psh[VW], bs[VW];
#pragma unroll
        for (int q = 0; q < VW / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(&s_coef[off + 4 * q]);
            bs[4 * q] = t.x; bs[4 * q + 1] = t.y; bs[4 * q + 2] = t.z; bs[4 * q + 3] = t.w;
            if (PRO) {
                const float4 a = *reinterpret_cast<const float4*>(&s_coef[C + off + 4 * q]);
                const float4 c = *reinterpret_cast<const float4*>(&s_coef[2 * C + off + 4 * q]);
                pa[4 * q] = a.x; pa[4 * q + 1] = a.y; pa[4 * q + 2] = a.z; pa[4 * q + 3] = a.w;
                psh[4 * q] = c.x; psh[4 * q + 1] = c.y; psh[4 * q + 2] = c.z; psh[4 * q + 3] = c.w;
            }
        }
        float kc1[VW], kc0[VW];
        if (BWD) {
#pragma unroll
            for (int q = 0; q < VW / 4; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(&s_coef[3 * C + off + 4 * q]);
                const float4 c = *reinterpret_cast<const float4*>(&s_coef[4 * C + off + 4 * q]);
                kc1[4 * q] = a.x; kc1[4 * q + 1] = a.y; kc1[4 * q + 2] = a.z; kc1[4 * q + 3] = a.w;
                kc0[4 * q] = c.x; kc0[4 * q + 1] = c.y; kc0[4 * q + 2] = c.z; kc0[4 * q + 3] = c.w;
            }
        }
        float q0[VW], q1[VW];
#pragma unroll
        for (int j = 0; j < VW; ++j) q0[j] = q1[j] = 0.f;
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int pos = wave * 16 + grp + 8 * q;
            const bool on = pos < nr;
            const int lrc = rows[q], lr = lrc;
            int es = NE > 0 ? 0 : s_rowptr[lrc] - e0;
            const int ee = NE > 0 ? tail_ee[q] : (on ? s_rowptr[lrc + 1] - e0 : es);
            float acc[VW];
#pragma unroll
            for (int j = 0; j < VW; ++j) acc[j] = 0.f;
            if (NE > 0) {                                        // entries from registers: the same sums in the same order
                const unsigned char* pq = bufs + B * kBuf;       // (compile-time buffer: immediate offsets)
#pragma unroll
                for (int k0 = 0; k0 < NEc; k0 += 4) {
                    if (k0 >= nmax) break;                       // (uniform)
                    // the quad's fourth entry is padding for EVERY row of the wave when nmax == k0 + 3 (the 7-entry vertex graph of
                    // a regular mesh, as in the lean kernel): its LDS read and its four FMAs with weight 0 are skipped wave-wide
                    const bool fourth = k0 + 3 < nmax;           // (uniform)
                    uint4 v[4], vy[BWD ? 4 : 1];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        v[k] = *reinterpret_cast<const uint4*>(pq + lo[q][k0 + k]);
                        if (BWD) vy[BWD ? k : 0] = *reinterpret_cast<const uint4*>(pq + PR * 128 + lo[q][k0 + k]);
                    }
                    if (fourth) {
                        v[3] = *reinterpret_cast<const uint4*>(pq + lo[q][k0 + 3]);
                        if (BWD) vy[BWD ? 3 : 0] = *reinterpret_cast<const uint4*>(pq + PR * 128 + lo[q][k0 + 3]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (k == 3 && !fourth) break;
                        float t[VW], y[VW];
                        Lane<T>::unpack(v[k], t);
                        if (BWD) Lane<T>::unpack(vy[BWD ? k : 0], y);
#pragma unroll
                        for (int j = 0; j < VW; ++j) {
                            const float x = BWD ? fmaf(pa[j], t[j] * lrelu_grad(fmaf(y[j], pa[j], psh[j]), slope), fmaf(kc1[j], y[j], kc0[j]))
                                                : PRO ? lrelu(fmaf(t[j], pa[j], psh[j]), slope) : t[j];
                            acc[j] = fmaf(wjr[q][k0 + k], x, acc[j]);
                        }
                    }
                }
                es = nmax > NEc ? tail_es[q] : ee;               // (uniform test: long rows continue from the LDS lists)
            }
            while (es < ee) {
                int li[4];
                float wj[4];
                uint4 v[4], vy[BWD ? 4 : 1];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ek = min(es + k, ee - 1);
                    li[k] = s_lc[ek];
                    wj[k] = (es + k < ee) ? s_w[ek] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = *reinterpret_cast<const uint4*>(pb + li[k] * 128);
                    if (BWD) vy[BWD ? k : 0] = *reinterpret_cast<const uint4*>(pb + PR * 128 + li[k] * 128);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t[VW], y[VW];
                    Lane<T>::unpack(v[k], t);
                    if (BWD) Lane<T>::unpack(vy[BWD ? k : 0], y);
#pragma unroll
                    for (int j = 0; j < VW; ++j) {
                        const float x = BWD ? fmaf(pa[j], t[j] * lrelu_grad(fmaf(y[j], pa[j], psh[j]), slope), fmaf(kc1[j], y[j], kc0[j]))
                                            : PRO ? lrelu(fmaf(t[j], pa[j], psh[j]), slope) : t[j];
                        acc[j] = fmaf(wj[k], x, acc[j]);
                    }
                }
                es += 4;
            }
            const float di = s_dinv[lrc];
            float o[VW];
#pragma unroll
            for (int j = 0; j < VW; ++j) o[j] = fmaf(acc[j], di, bs[j]);
            const uint4 ob = Lane<T>::pack(o);
            if (on) {
                typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
                nt_u4 ov = {ob.x, ob.y, ob.z, ob.w};
                __builtin_nontemporal_store(ov, reinterpret_cast<nt_u4*>(Y + (int64_t)(r0 + lr) * ldy + off));
            }
            if (RED == 2 && on) {                                // statistics of the values as stored, around the reference
                Lane<T>::unpack(ob, o);
#pragma unroll
                for (int j = 0; j < VW; ++j) {
                    const float d = o[j] - s_coef[3 * C + off + j];
                    q0[j] += d;
                    q1[j] = fmaf(d, d, q1[j]);
                }
            }
            if (RED == 1 && on) {                                // on the values as stored
                float y[VW];
                Lane<T>::unpack(ob, o);
                Lane<T>::unpack(*reinterpret_cast<const uint4*>(pb + PR * 128 + lr * 128), y);
#pragma unroll
                for (int j = 0; j < VW; ++j) {
                    const float g = o[j] * lrelu_grad(fmaf(y[j], s_coef[3 * C + off + j], s_coef[4 * C + off + j]), slope);
                    q0[j] += g;
                    q1[j] = fmaf(g, (y[j] - s_coef[5 * C + off + j]) * s_coef[6 * C + off + j], q1[j]);
                }
            }
        }
        if (RED) {                                               // this wave's 16 rows -> LDS; the record: next slab (record())
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                q0[j] = group8_sum(q0[j]);
                q1[j] = group8_sum(q1[j]);
            }
            if (grp == 0) {
                float* sp = s_part + (((s & 1) * 4 + wave) * 2) * CS + sl * VW;
#pragma unroll
                for (int j = 0; j < VW; ++j) {
                    sp[j] = q0[j];
                    sp[CS + j] = q1[j];
                }
            }
        }
    };
    for (int s = 0; s < n_slabs; s += kNB) {
        slab(s, std::integral_constant<int, 0>());
        if (s + 1 < n_slabs) slab(s + 1, std::integral_constant<int, 1>());
        if constexpr (kNB > 2) {
            if (s + 2 < n_slabs) slab(s + 2, std::integral_constant<int, 2>());
        }
        if constexpr (kNB > 3) {
            if (s + 3 < n_slabs) slab(s + 3, std::integral_constant<int, 3>());
        }
    }
    if (RED) {
        __syncthreads();
        record(n_slabs - 1);
    }
    if (h + 1 < nh) __syncthreads();                             // (every wave is done with the tables and buffers of this pass)
    }
    // (every copy was waited for by the slab that reads it; what is in flight here are stores: nothing to wait for)
}

// (two LDS buffers: three and four measured slower, profiles/r02_experiments)

template <typename T, int KD, bool PRO, int RED, int NB, bool BWD = false>
size_t patch2_lds(int C, int emax) {
    const int PR = 32 * KD;
    const size_t buf = (size_t)PR * 128 * (BWD ? 2 : 1) + (RED == 1 ? kRB * 128 : 0);
    const size_t cs = 128 / sizeof(T);
    return NB * buf + (size_t)emax * 4 + kRB * 4 + (kRB + 4) * 4 + (size_t)emax * 2 + kRB * 4 +
           (size_t)(RED == 1 ? 7 : RED == 2 ? 4 : BWD ? 5 : PRO ? 3 : 1) * C * 4 + (RED ? 2 * 4 * 2 * cs * 4 : 0);
}

template <typename T, int KD, bool PRO, int RED, int NB, int NE, bool BWD = false>
int launch_patch2nb(const ddmp_graph* g, const T* X, int64_t ldx, T* Y, int64_t ldy, int C, const float* bias, const float* ps,
                    const float* psh, float slope, hipStream_t st, RedArgs red, BwdArgs bw = BwdArgs()) {
    const int n = (int)g->n_rows;
    const int n_chunks = (int)cdiv(n, kRB);
    const int cpx = (int)cdiv(n_chunks, kXcd);
    // the entry tables hold the graph's largest taken chunk (face graph: 256 entries, not the 1024 a chunk may have at most: 4.5 KB
    // less, a third workgroup per CU for the fused reduction at C = 256)
    const int emax = std::min(kMaxE, std::max(64, (g->max_chunk_nnz + 63) / 64 * 64));
    const size_t lds = patch2_lds<T, KD, PRO, RED, NB, BWD>(C, emax);   // (the same for every NE)
    if (lds > 160 * 1024) return ddmp::kPatchNotApplicable;
    auto kern = spmm_patch2_kernel<T, KD, PRO, RED, NB, NE, BWD>;
    // > 64 KB of dynamic LDS needs the attribute, once per kernel AND device (a second device of the same process -- threaded
    // ranks -- has its own copy of the function)
    static std::atomic<uint32_t> attr_done{0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 32) return ddmp::kPatchNotApplicable;
    if (!(attr_done.load(std::memory_order_acquire) & (1u << dev))) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return ddmp::kPatchNotApplicable;
        }
        attr_done.fetch_or(1u << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(cpx * kXcd), dim3(256), lds, st, g->rowptr, g->col, g->lcol, g->ew, g->dinv, g->dinv_r, g->pl_ptr, g->pl_col, g->pl_split, X,
                       ldx, Y, ldy, n, C, bias, ps, psh, slope, cpx, n_chunks, emax, red, bw);
    LAUNCH_TRY();
    return DDMP_OK;
}

int patch_ne() {                                                  // DDMP_SPMM_PATCH_NE=0: entries from LDS per slab (A/B)
    static const int v = [] { const char* e = getenv("DDMP_SPMM_PATCH_NE"); return e ? atoi(e) : 1; }();
    return v;
}

template <typename T, int KD, bool PRO, int RED>
int launch_patch2(const ddmp_graph* g, const T* X, int64_t ldx, T* Y, int64_t ldy, int C, const float* bias, const float* ps,
                  const float* psh, float slope, hipStream_t st, RedArgs red) {
    // register entries: 4 where no row has more (face graphs), else 8 with the longer rows' tails from LDS (round 5)
    if (patch_ne() && g->max_row_nnz <= 4) return launch_patch2nb<T, KD, PRO, RED, 2, 4>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red);
    if (patch_ne()) return launch_patch2nb<T, KD, PRO, RED, 2, 8>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red);
    return launch_patch2nb<T, KD, PRO, RED, 2, 0>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red);
}

template <typename T, bool PRO, int RED>
int by_patch(const ddmp_graph* g, const T* X, int64_t ldx, T* Y, int64_t ldy, int C, const float* bias, const float* ps,
             const float* psh, float slope, hipStream_t st, RedArgs red) {
    switch (g->patch_kd) {                                       // (chosen per graph: ddmp_graph, graph.hip)
        case 3: return launch_patch2<T, 3, PRO, RED>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red);
        case 4: return launch_patch2<T, 4, PRO, RED>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red);
        case 5: return launch_patch2<T, 5, PRO, RED>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red);
        case 6: return launch_patch2<T, 6, PRO, RED>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red);
        default: return ddmp::kPatchNotApplicable;
    }
}

// Where this kernel runs (round 4: part of the default library).  Measured on MI355X, 1M-face mesh in Morton order, float32, two
// patch buffers, entries in registers (NE; profiles/r04_gather_patch_ab.txt), us per launch against the lean gather:
//   face graph (4 entries per row)    plain C = 512 797 vs 866, C = 256 408 vs 435; prologue C = 256 412 vs 446, C = 512 930 vs 900;
//                                     with the fused reduction 1342 vs 1271 / 664 vs 646; C = 128 227 vs 209
//   vertex graph (7 entries per row)  plain C = 512 468 vs 558, C = 256 251 vs 272; prologue C = 256 264 vs 289, C = 512 581 vs 597;
//                                     with the fused reduction 897 vs 755 / 367 vs 384; C = 128 152 vs 132
// (before the entries moved to registers the vertex graph lost everywhere: 583 / 277 us plain).  Round 5 (RCB numbering, patch_forms
// below): every form from C = 256 wins or ties.  DDMP_SPMM_PATCH: unset = the measured selection (ddmp_spmm_patch_selected), 0 =
// never, 1 = wherever it applies (A/B runs).
constexpr int patch_max_nnz() { return 1 << 30; }
int patch_mode() {
    static int m = -1;
    if (m < 0) {
        const char* e = getenv("DDMP_SPMM_PATCH");
        m = e ? atoi(e) : 3;
        if (m != 0 && m != 1) m = 3;
    }
    return m;
}

}  // namespace

// Forms taken (all four bits since round 5): bit 0 the prologue form at C >= 512, bit 1 the fused BatchNorm-backward reduction, bit 2
// the fused statistics form on the LDS-patch kernel (all from C = 256), bit 3 bfloat16 features on it (plain, prologue, statistics) -- round 5, measured with the RCB numbering (smaller
// patches: KD = 5 instead of 6, one more workgroup per CU), 1M faces, us per launch patch | lean (profiles/r05_gather_forms.txt):
//   prologue C = 512 face 812 | 935, vertex 498 | 627;  reduction C = 512 face 1246 | 1284, vertex 707 | 747, C = 256 face
//   639 | 646, vertex 370 | 394;  statistics C = 512 face 857 | 937, vertex 505 | 616, C = 256 face 432 | 469, vertex 264 | 308
constexpr int patch_forms() { return 15; }
// does ddmp_spmm* take the LDS-patch kernel for this graph and shape? (tests; the selection itself: patch_mode above)
// has_red: 0 | 1 the BatchNorm-backward reductions | 2 the statistics form | 3 the BatchNorm backward on the gather (ddmp_spmm_bnbwd_f32)
extern "C" int ddmp_spmm_patch_selected(const ddmp_graph* g, int C, int dtype, int has_pro, int has_red) {
    if (!g || !patch_mode() || g->max_patch <= 0 || g->max_patch > 192 || !g->lcol) return 0;
    if (has_red == 1 && has_pro) return 0;                       // (no such form)
    // the BatchNorm backward on the gather: patches of up to 128 rows (two tensors per buffer), C <= 512 (its five coefficient
    // rows + doubled buffers: 79 KB at C = 512 on the vertex graph, two workgroups per CU; beyond that one).  us per launch patch | lean,
    // 1M faces, RCB order, cold buffers (profiles/r06_gather_prologue_ab.txt E): face C = 512 1117 | 1345, 256 578 | 702, 128 299 | 345;
    // vertex 640 | 964, 335 | 473, 191 | 235
    if (has_red == 3 && (g->patch_kd > 4 || C > 512)) return 0;
    if (patch_mode() == 3) {
        // measured selection (by_patch's note, patch_forms); row lengths are not a condition since round 5
        // round 6: float32 from C = 128 -- with the one-round-trip set-up the kernel wins there too (us per launch patch | lean, 1M faces,
        // RCB order, cold buffers: face prologue 206 | 224, statistics 226 | 244, reduction 310 | 323; vertex 138 | 149, 158 | 167,
        // 197-206 | 200); at C = 64 it loses the prologue and reduction forms (profiles/r06_gather_prologue_ab.txt)
        // (bfloat16 features, round 6: plain / prologue / statistics from C = 128 too -- face prologue 120 | 146 us, statistics 139 | 172;
        //  vertex 96 | 105, 105 | 117, profiles/r06_gather_prologue_ab.txt F)
        if (g->max_row_nnz > patch_max_nnz() || C < 128) return 0;
        // bfloat16 features (round 5, 1M faces, RCB numbering, us per launch patch | slab kernel): plain C = 512 face 420 | 457,
        // vertex 251 | 311; prologue face 434 | 550, vertex 414 | 428; statistics face 546 | 617, vertex 377 | 425 (C = 256 alike);
        // the fused reduction LOSES there (face 819 | 730, vertex 493 | 448; round 6 with the new set-up: 703 | 681, 421 | 431 at C = 512,
        // 384 | 352, 235 | 224 at C = 256) and stays on the slab kernel
        if (dtype != DDMP_F32 && (has_red == 1 || !(patch_forms() & 8))) return 0;
        if (has_red == 1 && !(patch_forms() & 2)) return 0;
        if (has_red == 2 && !(patch_forms() & 4)) return 0;
        if (has_pro && C >= 512 && !(patch_forms() & 1)) return 0;
    }
    const int cs = dtype == DDMP_BF16 ? 64 : 32;
    return (C % cs == 0 && C >= 2 * cs && C <= 1024) ? 1 : 0;
}

namespace ddmp {

// -> DDMP_OK, an error, or kPatchNotApplicable (the caller takes its slab kernel).  Needs the graph's patch tables, C a multiple
// of the 128-byte slab and >= 2 slabs.  Fused epilogues: red_part + red_Yp = the BatchNorm-backward reductions; red_part without
// red_Yp = the statistics form around red_mean.  The caller processes g->heavy with its own kernel.
int spmm_patch(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype, const float* bias,
               const float* ps, const float* psh, float slope, const void* red_Yp, int64_t red_ldyp, const float* red_scale,
               const float* red_shift, const float* red_mean, const float* red_rstd, float* red_part, hipStream_t st) {
    const int kind = red_part ? (red_Yp ? 1 : 2) : 0;
    if (!ddmp_spmm_patch_selected(g, C, dtype, ps != nullptr, kind)) return kPatchNotApplicable;
    RedArgs red{red_Yp, red_ldyp, red_scale, red_shift, red_mean, red_rstd, red_part};
    if (dtype == DDMP_BF16) {
        auto x = static_cast<const bf16_t*>(X);
        auto y = static_cast<bf16_t*>(Y);
        if (kind == 1) return by_patch<bf16_t, false, 1>(g, x, ldx, y, ldy, C, bias, ps, psh, slope, st, red);
        if (kind == 2) return ps ? by_patch<bf16_t, true, 2>(g, x, ldx, y, ldy, C, bias, ps, psh, slope, st, red)
                                 : by_patch<bf16_t, false, 2>(g, x, ldx, y, ldy, C, bias, ps, psh, slope, st, red);
        if (ps) return by_patch<bf16_t, true, 0>(g, x, ldx, y, ldy, C, bias, ps, psh, slope, st, red);
        return by_patch<bf16_t, false, 0>(g, x, ldx, y, ldy, C, bias, ps, psh, slope, st, red);
    }
    auto x = static_cast<const float*>(X);
    auto y = static_cast<float*>(Y);
    if (kind == 1) return by_patch<float, false, 1>(g, x, ldx, y, ldy, C, bias, ps, psh, slope, st, red);
    if (kind == 2) return ps ? by_patch<float, true, 2>(g, x, ldx, y, ldy, C, bias, ps, psh, slope, st, red)
                             : by_patch<float, false, 2>(g, x, ldx, y, ldy, C, bias, ps, psh, slope, st, red);
    if (ps) return by_patch<float, true, 0>(g, x, ldx, y, ldy, C, bias, ps, psh, slope, st, red);
    return by_patch<float, false, 0>(g, x, ldx, y, ldy, C, bias, ps, psh, slope, st, red);
}

// out = A_hat . dY with dY the BatchNorm + LeakyReLU backward of (dZ, Yb) rebuilt on the gather: LDS-patch form.  The caller
// processes g->heavy with the lean kernel's BWD form.
int spmm_patch_bwd(const ddmp_graph* g, const void* dZ, int64_t lddz, const void* Yb, int64_t ldyb, void* out, int64_t ld_out, int C,
                   int dtype, const float* a, const float* b, const float* c1, const float* c0, float slope, hipStream_t st) {
    if (!ddmp_spmm_patch_selected(g, C, dtype, 1, 3)) return kPatchNotApplicable;
    BwdArgs bw{Yb, ldyb, c1, c0};
    RedArgs red{};
    const bool ne4 = g->max_row_nnz <= 4;
#define DDMP_PBWD(T_, KD_, NE_)                                                                                                  \
    launch_patch2nb<T_, KD_, true, 0, 2, NE_, true>(g, static_cast<const T_*>(dZ), lddz, static_cast<T_*>(out), ld_out, C, nullptr, a, b, \
                                                    slope, st, red, bw)
    if (dtype == DDMP_BF16) {
        switch (g->patch_kd) {
            case 3: return !patch_ne() ? DDMP_PBWD(bf16_t, 3, 0) : ne4 ? DDMP_PBWD(bf16_t, 3, 4) : DDMP_PBWD(bf16_t, 3, 8);
            case 4: return !patch_ne() ? DDMP_PBWD(bf16_t, 4, 0) : ne4 ? DDMP_PBWD(bf16_t, 4, 4) : DDMP_PBWD(bf16_t, 4, 8);
            default: return kPatchNotApplicable;
        }
    }
    switch (g->patch_kd) {
        case 3: return !patch_ne() ? DDMP_PBWD(float, 3, 0) : ne4 ? DDMP_PBWD(float, 3, 4) : DDMP_PBWD(float, 3, 8);
        case 4: return !patch_ne() ? DDMP_PBWD(float, 4, 0) : ne4 ? DDMP_PBWD(float, 4, 4) : DDMP_PBWD(float, 4, 8);
        default: return kPatchNotApplicable;
    }
#undef DDMP_PBWD
}

}  // namespace ddmp
