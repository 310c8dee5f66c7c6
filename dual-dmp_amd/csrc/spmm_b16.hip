// GCN aggregation on bfloat16 features:  Y[i,:] = bf16( dinv[i] * sum_{e in row i} dinv[col e] * f(X[col e,:]) (+ bias) )
//
// The bf16-feature form of spmm.hip's slab kernel (GCNConv.propagate of PyG 2.2.0, util/networks.py:51-62 of the
// reference).  Same work layout -- a workgroup owns 64 consecutive rows, its CSR slice and the dinv[col] weights are
// staged once in LDS, the rows are walked one 128-byte slab at a time, XCD-aware chunk mapping, no atomics -- but a lane
// now carries 8 channels (16 bytes of bf16), so a 128-byte slab is 64 channels and a gathered line feeds twice the
// channels of the float32 kernel.  Gathered elements are unpacked to float32, the prologue / BatchNorm-backward rebuild
// and the accumulation run in float32, the result is rounded to nearest-even once.
// Algorithmic bytes per call: 2*N*C*2 + 4*nnz + 4*(N+1) + 4*N  (SURVEY.md §8d with s = 2).
#include "b16_common.h"
#include "finalize.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

using namespace ddmp;

constexpr int kRB = 64;
constexpr int kMaxE = kRB * 16;

struct BnRedB {                    // see BnRed in spmm.hip
    const bf16_t* Yp;
    int64_t ldyp;
    const float *scale, *shift, *mean, *rstd;
    float* part;
};
struct BnBwdGatherB {              // see BnBwdGather in spmm.hip
    const bf16_t* Yb;
    int64_t ldyb;
    const float *c1, *c0;
};

// one lane = VW channels: 16 bytes (VW = 8) or 8 bytes (VW = 4) of a row
template <int VW> struct Piece;
template <> struct Piece<8> {
    typedef uint4 raw;
    static __device__ __forceinline__ raw ld(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
    static __device__ __forceinline__ void unpack(raw u, float (&f)[8]) { bf_unpack8(u, f); }
    static __device__ __forceinline__ raw pack(const float (&f)[8]) { return bf_pack8(f); }
    static __device__ __forceinline__ void st(bf16_t* p, raw v) { nt_st8b(p, v); }
};
template <> struct Piece<4> {
    typedef uint2 raw;
    static __device__ __forceinline__ raw ld(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
    static __device__ __forceinline__ void unpack(raw u, float (&f)[4]) {
        f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
    }
    static __device__ __forceinline__ raw pack(const float (&f)[4]) { return make_uint2(bf_pack(f[0], f[1]), bf_pack(f[2], f[3])); }
    static __device__ __forceinline__ void st(bf16_t* p, raw v) {
        typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
        nt_u2 o = {v.x, v.y};
        __builtin_nontemporal_store(o, reinterpret_cast<nt_u2*>(p));
    }
};
template <int VW> __device__ __forceinline__ void ldcf(const float* p, float (&f)[VW]) {
#pragma unroll
    for (int q = 0; q < VW / 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
        f[4 * q] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w;
    }
}

// LANES lanes x VW channels per row and slab (128-byte slabs: LANES x VW = 64; narrower rows: fewer lanes).
// VW = 8 (16-byte pieces) by default; VW = 4 exists for A/B runs of the fused forms (see dispatch_b16).
// LEAN (default; see spmm_lean.inc for the measurements): everything that does not depend on the slab is done once per chunk
// at staging time -- an entry is (row offset in 16-byte units, weight), rows own fixed slots of `stride` entries padded to
// whole quads with weight-0 copies of their last entry -- so the address of a gathered piece is one v_lshl_add_u64 on the
// lane's slab base and the inner loop reads four entries with two ds_read_b128, without clamping or branches.
// RED: 0 | 1 BatchNorm-backward reductions of the output | 2 BatchNorm statistics of the (rounded) output around red.mean
// (see spmm_lean.inc).  Round 5: a row's slots are its own quads, packed per chunk (rows of any length; a chunk whose packed
// slots exceed the LDS array reads its entries from global memory), `chunk_list` = the LDS-patch kernel's heavy chunks.
template <int VW, int LANES, int U, bool PRO, int RED, bool BWD, bool LEAN>
__global__ __launch_bounds__(256) void spmm_slab_b16_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ dinv, const float* __restrict__ dinv_r,
    const bf16_t* __restrict__ X, int64_t ldx, bf16_t* __restrict__ Y, int64_t ldy, int n_rows, int C,
    const float* __restrict__ bias, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float slope, int chunks_per_xcd, int n_chunks, const int* __restrict__ chunk_list, BnRedB red, BnBwdGatherB bwd) {
    static_assert(!BWD || (PRO && !RED), "BWD: the coefficients a, b come as the prologue's");
    static_assert(!LEAN || U == 4, "quads");
    typedef Piece<VW> PC;
    constexpr int CS = LANES * VW;
    constexpr int RPW = 64 / LANES;
    constexpr int RPB = 4 * RPW;
    __shared__ int s_rowptr[kRB + 1];
    __shared__ int s_slot[kRB];                                  // LEAN: (first quad of the row's slots) * 512 + its quad count
    __shared__ int s_perm[kRB];                                  // the chunk's rows, longest first (LEAN; else the identity)
    __shared__ __attribute__((aligned(16))) uint2 s_ent[kMaxE];  // LEAN: (offset, weight) slots; else: s_col | s_w
    int* s_col = reinterpret_cast<int*>(s_ent);
    float* s_w = reinterpret_cast<float*>(s_ent) + kMaxE;
    // RED: the previous layer's BatchNorm coefficients, per channel (scale, shift, rstd, -mean * rstd).  In LDS and read per
    // row: as four per-lane vectors they cost 32 registers (76 in all: 6 waves per SIMD instead of 8) -- the reason this
    // fused form measured slower than SpMM + reduction pass (round 2: 939 vs 457 + 390 us at C = 512)
    constexpr int kRedC = 1024;
    __shared__ __attribute__((aligned(16))) float s_red[RED == 1 ? 4 * kRedC : RED == 2 ? kRedC : 4];
    __shared__ float s_part[RED ? 2 : 1][RED ? 4 : 1][RED ? 64 : 1];    // RED: the waves' partials of a slab (see the epilogue)

    int chunk;
    if (chunk_list) {
        if ((int)blockIdx.x >= n_chunks) return;
        chunk = chunk_list[blockIdx.x];
    } else {
        chunk = (blockIdx.x & (kXcd - 1)) * chunks_per_xcd + (blockIdx.x >> 3);
        if (chunk >= n_chunks) return;
    }
    const int r0 = chunk * kRB;
    const int nr = min(kRB, n_rows - r0);
    const int tid = threadIdx.x;
    for (int i = tid; i <= nr; i += 256) s_rowptr[i] = rowptr[r0 + i];
    __syncthreads();
    const int e0 = s_rowptr[0];
    const int ne = s_rowptr[nr] - e0;
    bool staged = ne <= kMaxE;
    if (RED == 1) {
        for (int i = tid; i < C; i += 256) {
            const float rs = red.rstd[i];
            s_red[i] = red.scale[i];
            s_red[kRedC + i] = red.shift[i];
            s_red[2 * kRedC + i] = rs;
            s_red[3 * kRedC + i] = -red.mean[i] * rs;
        }
    }
    if (RED == 2)
        for (int i = tid; i < C; i += 256) s_red[i] = red.mean[i];
    const unsigned ld16 = (unsigned)(ldx >> 3);                  // row pitch in 16-byte units
    if (LEAN) {
        // every wave: quad counts of the chunk's rows (lane = row), their exclusive prefix sum; thread t stages the slots
        // k = t & 3, + 4, ... of row t >> 2 (see spmm_lean.inc)
        const int ln = tid & 63;
        const int nn_l = ln < nr ? s_rowptr[ln + 1] - s_rowptr[ln] : 0;
        const int nq_l = (nn_l + 3) >> 2;
        const int nq0 = __builtin_amdgcn_readfirstlane(nq_l);   // (regular meshes: one ballot instead of the scan, spmm_lean.inc)
        const bool uni = __ballot(nq_l != nq0) == 0ull;
        int incl = (ln + 1) * nq0;
        if (!uni) {
            incl = nq_l;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (ln >= o) incl += t;
            }
        }
        const int qoff_l = incl - nq_l;
        staged = (uni ? 64 * nq0 : __shfl(incl, 63, 64)) * 4 <= kMaxE;   // (the same in every wave)
        const int rank_l = chunk_rank_desc(ln < nr ? min(nq_l, 31) : -1, ln);
        if (tid < 64) s_perm[rank_l] = ln;
        if (tid < nr) s_slot[tid] = staged ? qoff_l * 512 + nq_l : 0;
        if (staged) {
            const int lr = tid >> 2;
            const int rb = s_rowptr[min(lr, nr)];
            const int nn = lr < nr ? s_rowptr[lr + 1] - rb : 0, nq = (nn + 3) >> 2;
            const int qo = uni ? lr * nq0 : __shfl(qoff_l, lr, 64);
            for (int k = tid & 3; k < 4 * nq; k += 4) {          // padding: the row's last entry again, weight 0
                const int c = col[rb + min(k, nn - 1)];
                s_ent[4 * qo + k] = make_uint2((unsigned)c * ld16, k < nn ? __float_as_uint(dinv[c]) : 0u);
            }
        }
    } else {
        if (tid < 64) s_perm[tid] = tid;
        if (staged)
            for (int t = tid; t < ne; t += 256) {
                const int c = col[e0 + t];
                s_col[t] = c;
                s_w[t] = dinv[c];
            }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int grp = lane / LANES, sl = lane % LANES;
    // slab groups (gridDim.y > 1: the heavy-chunk launches): this workgroup walks only its share of the C / CS slabs
    const int n_slabs_all = (C + CS - 1) / CS;
    const int slabs_per_group = (n_slabs_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int c_begin = (int)blockIdx.y * slabs_per_group * CS, c_end = min(C, c_begin + slabs_per_group * CS);
    auto slabs = [&](auto slots_tag) {
    constexpr bool SLOTS = decltype(slots_tag)::value;           // LEAN and the chunk's slots fit: entries from s_ent
    for (int c0 = c_begin; c0 < c_end; c0 += CS) {
        const int off = c0 + sl * VW;
        float pa[VW], pb[VW], k1[VW], k0[VW];
        float q0[VW], q1[VW];
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            pa[j] = 1.f; pb[j] = 0.f; k1[j] = 0.f; k0[j] = 0.f;
            q0[j] = 0.f; q1[j] = 0.f;
        }
        if (PRO) {
            ldcf<VW>(pscale + off, pa);
            ldcf<VW>(pshift + off, pb);
        }
        if (BWD) {
            ldcf<VW>(bwd.c1 + off, k1);
            ldcf<VW>(bwd.c0 + off, k0);
        }
        const bf16_t* xc = X + off;
        const bf16_t* yc = BWD ? bwd.Yb + off : nullptr;
        for (int pos = wave * RPW + grp; pos < nr; pos += RPB) {
            const int lr = s_perm[pos];
            const int sq = SLOTS ? s_slot[lr] : 0;
            int es = SLOTS ? 0 : s_rowptr[lr] - e0;
            const int ee = SLOTS ? 4 * (sq & 511) : s_rowptr[lr + 1] - e0;
            const uint4* ep = reinterpret_cast<const uint4*>(&s_ent[SLOTS ? 4 * (sq >> 9) : 0]);
            float acc[VW];
#pragma unroll
            for (int j = 0; j < VW; ++j) acc[j] = 0.f;
            while (es < ee) {
                int cj[U];
                float wj[U];
                typename PC::raw v[U], vy[BWD ? U : 1];
                if (SLOTS) {
                    const uint4 ea = ep[es >> 1], eb = ep[(es >> 1) + 1];
                    const unsigned o[4] = {ea.x, ea.z, eb.x, eb.z};
                    wj[0] = __uint_as_float(ea.y);
                    wj[1] = __uint_as_float(ea.w);
                    wj[2] = __uint_as_float(eb.y);
                    wj[U - 1] = __uint_as_float(eb.w);
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        const uint64_t bo = (uint64_t)o[k & 3] << 4;
                        v[k] = PC::ld(reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(xc) + bo));
                        if (BWD) vy[BWD ? k : 0] = PC::ld(reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(yc) + bo));
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        const int ek = min(es + k, ee - 1);
                        if (!LEAN && staged) {
                            cj[k] = s_col[ek];
                            wj[k] = s_w[ek];
                        } else {
                            cj[k] = col[e0 + ek];
                            wj[k] = dinv[cj[k]];
                        }
                        if (es + k >= ee) wj[k] = 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        v[k] = PC::ld(xc + (int64_t)cj[k] * ldx);
                        if (BWD) vy[BWD ? k : 0] = PC::ld(yc + (int64_t)cj[k] * bwd.ldyb);
                    }
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    float t[VW];
                    PC::unpack(v[k], t);
                    if (BWD) {
                        float y[VW];
                        PC::unpack(vy[BWD ? k : 0], y);
#pragma unroll
                        for (int j = 0; j < VW; ++j)
                            t[j] = fmaf(pa[j], t[j] * lrelu_grad(fmaf(y[j], pa[j], pb[j]), slope), fmaf(k1[j], y[j], k0[j]));
                    } else if (PRO) {
#pragma unroll
                        for (int j = 0; j < VW; ++j) t[j] = lrelu(fmaf(t[j], pa[j], pb[j]), slope);
                    }
#pragma unroll
                    for (int j = 0; j < VW; ++j) acc[j] = fmaf(wj[k], t[j], acc[j]);
                }
                es += U;
            }
            const int row = r0 + lr;
            const float di = dinv_r[row];
            float o[VW], bs[VW];
#pragma unroll
            for (int j = 0; j < VW; ++j) bs[j] = 0.f;
            if (bias) ldcf<VW>(bias + off, bs);                  // (per row, from the L1: not worth registers across the loop)
#pragma unroll
            for (int j = 0; j < VW; ++j) o[j] = fmaf(acc[j], di, bs[j]);
            const typename PC::raw ob = PC::pack(o);
            PC::st(Y + (int64_t)row * ldy + off, ob);
            if (RED == 2) {                                      // statistics of the values as stored (rounded), around ref
                PC::unpack(ob, o);
#pragma unroll
                for (int h = 0; h < VW / 4; ++h) {
                    int roff = off + 4 * h;
                    asm volatile("" : "+v"(roff));
                    const float4 rf = *reinterpret_cast<const float4*>(s_red + roff);
                    const float r4[4] = {rf.x, rf.y, rf.z, rf.w};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const float d = o[4 * h + jj] - r4[jj];
                        q0[4 * h + jj] += d;
                        q1[4 * h + jj] = fmaf(d, d, q1[4 * h + jj]);
                    }
                }
            }
            if (RED == 1) {                                      // on the values as stored (rounded)
                float y[VW];
                PC::unpack(ob, o);
                PC::unpack(PC::ld(red.Yp + (int64_t)row * red.ldyp + off), y);
                // four channels at a time, coefficients straight from LDS (opaque offsets: read per row, not hoisted)
#pragma unroll
                for (int h = 0; h < VW / 4; ++h) {
                    int roff = off + 4 * h;
                    asm volatile("" : "+v"(roff));
                    const float4 ra = *reinterpret_cast<const float4*>(s_red + roff);
                    const float4 rb = *reinterpret_cast<const float4*>(s_red + kRedC + roff);
                    const float4 rrs = *reinterpret_cast<const float4*>(s_red + 2 * kRedC + roff);
                    const float4 rm = *reinterpret_cast<const float4*>(s_red + 3 * kRedC + roff);
                    const float a4[4] = {ra.x, ra.y, ra.z, ra.w}, b4[4] = {rb.x, rb.y, rb.z, rb.w};
                    const float s4[4] = {rrs.x, rrs.y, rrs.z, rrs.w}, m4[4] = {rm.x, rm.y, rm.z, rm.w};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = 4 * h + jj;
                        const float g = o[j] * lrelu_grad(fmaf(y[j], a4[jj], b4[jj]), slope);
                        q0[j] += g;
                        q1[j] = fmaf(g, fmaf(y[j], s4[jj], m4[jj]), q1[j]);      // yhat = (y - mean) rstd
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (RED) {                                               // per chunk: ONE partial record per channel (round 4: the four
#pragma unroll                                                   // waves' sums meet in LDS -- a quarter of the partial traffic,
            for (int j = 0; j < VW; ++j) {                       // which was 17 % of the tensor bytes at C = 512 in bf16)
                if (LANES == 8) {                                // (DPP + lane swaps instead of the LDS crossbar: ddmp_common.h)
                    q0[j] = group8_sum(q0[j]);
                    q1[j] = group8_sum(q1[j]);
                } else {
#pragma unroll
                    for (int o = LANES; o < 64; o <<= 1) {
                        q0[j] += __shfl_xor(q0[j], o, 64);
                        q1[j] += __shfl_xor(q1[j], o, 64);
                    }
                }
            }
            if (grp == 0) {
#pragma unroll
                for (int j = 0; j < VW; ++j) {
                    s_part[0][RED ? wave : 0][RED ? sl * VW + j : 0] = q0[j];
                    s_part[RED ? 1 : 0][RED ? wave : 0][RED ? sl * VW + j : 0] = q1[j];
                }
            }
            __syncthreads();
            if (tid < 2 * CS) {                                  // (waves in a fixed order, float64: the sums do not depend on timing)
                const int which = tid / CS, c = tid % CS;
                const double t = ((double)s_part[RED ? which : 0][0][RED ? c : 0] + (double)s_part[RED ? which : 0][RED ? 1 : 0][RED ? c : 0]) +
                                 ((double)s_part[RED ? which : 0][RED ? 2 : 0][RED ? c : 0] + (double)s_part[RED ? which : 0][RED ? 3 : 0][RED ? c : 0]);
                if (c0 + c < C) red.part[((int64_t)chunk * 2 + which) * C + c0 + c] = (float)t;
            }
            __syncthreads();
        }
    }
    };
    if (LEAN && staged) slabs(std::true_type());
    else slabs(std::false_type());
}

#include "fpartials.inc"

bool lean_b16_ok(const ddmp_graph* g, int64_t ldx) {             // offsets in 16-byte units fit 32 bits
    return (ldx & 7) == 0 && (uint64_t)g->n_cols * (uint64_t)(ldx >> 3) < (1ull << 32);
}

// heavy / n_heavy: only these chunks (the LDS-patch kernel's heavy list), one workgroup each
template <int VW, int LANES, bool PRO, int RED, bool BWD>
int launch_b16(const ddmp_graph* g, const bf16_t* X, int64_t ldx, bf16_t* Y, int64_t ldy, int C, const float* bias,
               const float* ps, const float* psh, float slope, hipStream_t st, BnRedB red, BnBwdGatherB bwd,
               const int* heavy = nullptr, int n_heavy = 0) {
    const int n = (int)g->n_rows;
    const int n_chunks = heavy ? n_heavy : (int)cdiv(n, kRB);
    const int cpx = (int)cdiv(n_chunks, kXcd);
    if (heavy && n_heavy <= 0) return DDMP_OK;
    // (a chunk list: the slabs of a chunk are split over workgroups, ~2048 in all -- see launch_lean in spmm.hip)
    const int n_sl = std::max((C + LANES * VW - 1) / (LANES * VW), 1);
    const int want = heavy ? std::min(n_sl, std::max(1, 2048 / n_heavy)) : 1;
    const int per = (n_sl + want - 1) / want;
    const dim3 grid(heavy ? n_heavy : cpx * kXcd, heavy ? (n_sl + per - 1) / per : 1);
    // lean staging where the rows fit their slots and the offsets (16-byte units) fit 32 bits; DDMP_SPMM_LEAN=0 for A/B
    static int lean_on = -1;
    if (lean_on < 0) {
        const char* e = getenv("DDMP_SPMM_LEAN");
        lean_on = (e && atoi(e) == 0) ? 0 : 1;
    }
    const bool lean = lean_on && lean_b16_ok(g, ldx) && (!BWD || bwd.ldyb == ldx);
    if (lean)
        hipLaunchKernelGGL((spmm_slab_b16_kernel<VW, LANES, 4, PRO, RED, BWD, true>), grid, dim3(256), 0, st, g->rowptr,
                           g->col, g->dinv, g->dinv_r, X, ldx, Y, ldy, n, C, bias, ps, psh, slope, cpx, n_chunks, heavy, red, bwd);
    else
        hipLaunchKernelGGL((spmm_slab_b16_kernel<VW, LANES, 4, PRO, RED, BWD, false>), grid, dim3(256), 0, st, g->rowptr,
                           g->col, g->dinv, g->dinv_r, X, ldx, Y, ldy, n, C, bias, ps, psh, slope, cpx, n_chunks, heavy, red, bwd);
    LAUNCH_TRY();
    return DDMP_OK;
}

template <bool PRO, int RED, bool BWD>
int dispatch_b16(const ddmp_graph* g, const bf16_t* X, int64_t ldx, bf16_t* Y, int64_t ldy, int C, const float* bias,
                 const float* ps, const float* psh, float slope, hipStream_t st, BnRedB red = BnRedB(),
                 BnBwdGatherB bwd = BnBwdGatherB(), const int* heavy = nullptr, int n_heavy = 0) {
    // (8-byte pieces in the fused forms measured SLOWER although they restore full occupancy -- 1M-face graph, C = 512: prologue form
    // 771 vs 613 us, reduction form 1043 vs 999 us: these forms are bound by instructions issued per gathered byte.)
    if (C % 64 == 0) return launch_b16<8, 8, PRO, RED, BWD>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red, bwd, heavy, n_heavy);
    if (C % 32 == 0) return launch_b16<8, 4, PRO, RED, BWD>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red, bwd, heavy, n_heavy);
    if (C % 16 == 0) return launch_b16<8, 2, PRO, RED, BWD>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red, bwd, heavy, n_heavy);
    return launch_b16<8, 1, PRO, RED, BWD>(g, X, ldx, Y, ldy, C, bias, ps, psh, slope, st, red, bwd, heavy, n_heavy);
}

bool shape_ok(const void* X, int64_t ldx, const void* Y, int64_t ldy, int C) {
    return C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C && b16_aligned(X) && b16_aligned(Y);
}
bool coef_ok(const float* p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int ddmp_spmm_bf16(const ddmp_graph* g, const uint16_t* X, int64_t ldx, uint16_t* Y, int64_t ldy, int C,
                              const float* bias, const float* pro_scale, const float* pro_shift, float slope,
                              ddmp_stream stream) {
    ARG_TRY(g && X && Y && X != Y && shape_ok(X, ldx, Y, ldy, C));
    ARG_TRY((pro_scale == nullptr) == (pro_shift == nullptr));
    ARG_TRY(coef_ok(bias) && coef_ok(pro_scale) && coef_ok(pro_shift));
    hipStream_t st = (hipStream_t)stream;
    {
        const int rc = ddmp::spmm_patch(g, X, ldx, Y, ldy, C, DDMP_BF16, bias, pro_scale, pro_shift, slope, nullptr, 0, nullptr,
                                        nullptr, nullptr, nullptr, nullptr, st);
        if (rc == DDMP_OK && g->n_heavy > 0) {                   // its heavy chunks: the slab kernel over the list
            if (pro_scale) return dispatch_b16<true, 0, false>(g, X, ldx, Y, ldy, C, bias, pro_scale, pro_shift, slope, st, BnRedB(),
                                                               BnBwdGatherB(), g->heavy, g->n_heavy);
            return dispatch_b16<false, 0, false>(g, X, ldx, Y, ldy, C, bias, nullptr, nullptr, slope, st, BnRedB(), BnBwdGatherB(),
                                                 g->heavy, g->n_heavy);
        }
        if (rc != ddmp::kPatchNotApplicable) return rc;
    }
    if (pro_scale) return dispatch_b16<true, 0, false>(g, X, ldx, Y, ldy, C, bias, pro_scale, pro_shift, slope, st);
    return dispatch_b16<false, 0, false>(g, X, ldx, Y, ldy, C, bias, nullptr, nullptr, slope, st);
}

extern "C" size_t ddmp_spmm_bnred_bf16_workspace_bytes(int64_t n_rows, int C) {
    if (n_rows <= 0 || C <= 0) return 0;
    return (size_t)ddmp::cdiv(n_rows, kRB) * 4 * 2 * (size_t)C * sizeof(float) + 256 + fpartials_mid_bytes(C);
}

extern "C" int ddmp_spmm_bnred_bf16(const ddmp_graph* g, const uint16_t* X, int64_t ldx, uint16_t* Y, int64_t ldy, int C,
                                    const uint16_t* Yp, int64_t ldyp, const float* scale, const float* shift,
                                    const float* mean, const float* rstd, float slope, double* sums2, void* ws,
                                    size_t ws_bytes, ddmp_stream stream) {
    ddmp::FinalizeScope fin_scope(sums2, stream, C);
    ARG_TRY(g && X && Y && Yp && scale && shift && mean && rstd && sums2 && ws && X != Y && shape_ok(X, ldx, Y, ldy, C));
    ARG_TRY(ldyp >= C && ldyp % 8 == 0 && b16_aligned(Yp) && b16_aligned(ws));
    ARG_TRY(coef_ok(scale) && coef_ok(shift) && coef_ok(mean) && coef_ok(rstd));
    if (ws_bytes < ddmp_spmm_bnred_bf16_workspace_bytes(g->n_rows, C)) return DDMP_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int n_chunks = (int)cdiv(g->n_rows, kRB);
    BnRedB red{Yp, ldyp, scale, shift, mean, rstd, (float*)ws};
    int rc = ddmp::spmm_patch(g, X, ldx, Y, ldy, C, DDMP_BF16, nullptr, nullptr, nullptr, slope, Yp, ldyp, scale, shift, mean, rstd,
                              (float*)ws, st);
    const bool patched = rc == DDMP_OK;
    if (rc == DDMP_OK && g->n_heavy > 0)
        rc = dispatch_b16<false, 1, false>(g, X, ldx, Y, ldy, C, nullptr, nullptr, nullptr, slope, st, red, BnBwdGatherB(), g->heavy,
                                           g->n_heavy);
    if (rc == ddmp::kPatchNotApplicable)
        rc = dispatch_b16<false, 1, false>(g, X, ldx, Y, ldy, C, nullptr, nullptr, nullptr, slope, st, red);
    if (rc != DDMP_OK) return rc;
    const size_t pbytes = ((size_t)n_chunks * 4 * 2 * (size_t)C * sizeof(float) + 255) / 256 * 256;
    fpartials_reduce((const float*)ws, n_chunks + (patched ? g->n_split : 0), C, C, (double*)((char*)ws + pbytes), sums2, st);    // (one record per chunk, both kernels; + the LDS-patch kernel's split chunks)
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_spmm_stats_bf16(const ddmp_graph* g, const uint16_t* X, int64_t ldx, uint16_t* Y, int64_t ldy, int C,
                                    const float* bias, const float* pro_scale, const float* pro_shift, float slope,
                                    const float* ref, double* sums2, void* ws, size_t ws_bytes, ddmp_stream stream) {
    // = ddmp_spmm_bf16 + ddmp_bn_stats_bf16 of the stored output, the statistics from the gather's epilogue around `ref`
    ddmp::FinalizeScope fin_scope(sums2, stream, C);
    ARG_TRY(g && X && Y && sums2 && ws && X != Y && shape_ok(X, ldx, Y, ldy, C));
    ARG_TRY((pro_scale == nullptr) == (pro_shift == nullptr));
    ARG_TRY(coef_ok(bias) && coef_ok(pro_scale) && coef_ok(pro_shift) && coef_ok(ref) && b16_aligned(ws));
    if (ws_bytes < ddmp_spmm_bnred_bf16_workspace_bytes(g->n_rows, C)) return DDMP_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!ref || C % 16 != 0 || C > 1024) {
        int rc = ddmp_spmm_bf16(g, X, ldx, Y, ldy, C, bias, pro_scale, pro_shift, slope, stream);
        if (rc != DDMP_OK) return rc;
        return ddmp_bn_stats_bf16(Y, ldy, g->n_rows, C, sums2, ws, ws_bytes, stream);
    }
    const int n_chunks = (int)cdiv(g->n_rows, kRB);
    BnRedB red{nullptr, 0, nullptr, nullptr, ref, nullptr, (float*)ws};
    int rc = ddmp::spmm_patch(g, X, ldx, Y, ldy, C, DDMP_BF16, bias, pro_scale, pro_shift, slope, nullptr, 0, nullptr, nullptr, ref,
                              nullptr, (float*)ws, st);
    const bool patched = rc == DDMP_OK;
    if (rc != DDMP_OK && rc != ddmp::kPatchNotApplicable) return rc;
    const int* list = patched ? g->heavy : nullptr;
    if (!patched || g->n_heavy > 0)
        rc = pro_scale ? dispatch_b16<true, 2, false>(g, X, ldx, Y, ldy, C, bias, pro_scale, pro_shift, slope, st, red, BnBwdGatherB(),
                                                      list, g->n_heavy)
                       : dispatch_b16<false, 2, false>(g, X, ldx, Y, ldy, C, bias, nullptr, nullptr, slope, st, red, BnBwdGatherB(),
                                                       list, g->n_heavy);
    if (rc != DDMP_OK) return rc;
    const size_t pbytes = ((size_t)n_chunks * 4 * 2 * (size_t)C * sizeof(float) + 255) / 256 * 256;
    fpartials_reduce((const float*)ws, n_chunks + (patched ? g->n_split : 0), C, C, (double*)((char*)ws + pbytes), sums2, st, ref,
                     (double)g->n_rows);    // (one record per chunk; + the LDS-patch kernel's split chunks)
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_spmm_bnbwd_bf16(const ddmp_graph* g, const uint16_t* dZ, int64_t lddz, const uint16_t* Yb,
                                    int64_t ldyb, uint16_t* out, int64_t ld_out, int C, const float* a, const float* b,
                                    const float* c1, const float* c0, float slope, ddmp_stream stream) {
    ARG_TRY(g && dZ && Yb && out && a && b && c1 && c0 && dZ != out && Yb != out && shape_ok(dZ, lddz, out, ld_out, C));
    ARG_TRY(ldyb >= C && ldyb % 8 == 0 && b16_aligned(Yb) && coef_ok(a) && coef_ok(b) && coef_ok(c1) && coef_ok(c0));
    BnBwdGatherB bwd{Yb, ldyb, c1, c0};
    {   // LDS-patch form where it applies (round 6); its heavy chunks: the slab kernel over the list
        const int rc = ddmp::spmm_patch_bwd(g, dZ, lddz, Yb, ldyb, out, ld_out, C, DDMP_BF16, a, b, c1, c0, slope, (hipStream_t)stream);
        if (rc == DDMP_OK && g->n_heavy > 0)
            return dispatch_b16<true, 0, true>(g, dZ, lddz, out, ld_out, C, nullptr, a, b, slope, (hipStream_t)stream, BnRedB(), bwd, g->heavy,
                                               g->n_heavy);
        if (rc != ddmp::kPatchNotApplicable) return rc;
    }
    return dispatch_b16<true, 0, true>(g, dZ, lddz, out, ld_out, C, nullptr, a, b, slope, (hipStream_t)stream, BnRedB(), bwd);
}
