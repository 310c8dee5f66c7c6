// Wide f16x3 weight gradient, round 4:   partial[split][m][k] = sum_{rows of the split} g(G)[row][m] * f(Z)[row][k]
//
//   GCNConv.lin weight gradient (util/networks.py:51-62 under main.py:107) of the layers with C_in, C_out >= 256.
//
// What was wrong with the round-1..3 kernel (gemm_tn_panel_kernel<13, ..>, gemm_panel.inc) -- measured by removing parts of
// it (profiles/r04_tn_ablation.txt, 1M rows, 512 x 512): data path alone 778 us (plain) / 1358 us (BatchNorm backward on the G
// load: three operand streams) = the HBM rate; matrix pipe alone 779-859 us; together 1745 / 2366 us -- the SUM.  Its loads
// were 8 (12) dword loads per thread and operand inside `if (row + 8 <= n_rows) .. else ..`: exec-masked side blocks.  The
// compiler cannot count younger loads across a block that may be skipped, so every use waited for vmcnt(0) -- including the
// loads issued one phase before --, and sched_group_barrier does not interleave across blocks.  Made unconditional, the 24
// per-element addresses overflowed the register file (scratch reloads inside the loop).
//
// This kernel keeps the 256 x 256 panel per workgroup, 8 waves as 4 (m) x 2 (k) of 64 x 128, three products per 16-row stage,
// and changes the data path:
//   * a wave loads WHOLE ROWS: lane l = columns 4l..4l+3 of stage rows (wave, wave + 8) -- one global_load_dwordx4 per row and
//     operand, 4 (6) per thread and stage instead of 16 (24); the row is wave-uniform, so its clamp to the matrix end and its
//     address are scalar arithmetic (SGPR base + one constant VGPR offset), every load is unconditional straight-line code
//     and the compiler's vmcnt counts are exact: the loads of stage s + 3 are issued in iteration s and waited for in s + 2;
//   * prologue, scale, split as before, but the two f16 planes are written ROW-major (ds_write_b64: one 512-byte row per wave
//     instruction; 64-byte chunk c of row r at chunk c ^ (r & 3)) and the MFMA fragments -- 8 consecutive rows of one column --
//     come out of ds_read_b64_tr_b16, the layout gemm_tn_b16_kernel (gemm_b16.hip) uses for bf16 features;
//   * rows beyond the split read as zeros through the descriptor of a plain operand (buffer range = the split), the BatchNorm
//     coefficients sit in LDS.
// The values that reach the MFMAs, their k order and the product order are those of the old kernel: same results bit for bit.
#include "gemm_tn_rm.h"

#include <cstdlib>
#include <type_traits>

namespace {

using namespace ddmp;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#include "gemm_f16s.inc"

constexpr int kRows = 16;                                        // rows per stage = one MFMA k-step
constexpr int kT = 256;                                          // panel edge

template <int T>
__device__ __forceinline__ int rm_off(int row, int col) {         // element offset in a [16][T] plane (T = 256 | 128)
    return row * T + ((((col >> 5) ^ (row & 3)) << 5) | (col & 31));
}

__device__ __forceinline__ void lds_barrier() {                  // LDS traffic only; register loads stay in flight
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef DDMP_TN_ABLATE                                            // timing-only diagnostic builds (results invalid), bits:
#define DDMP_TA_(bit, stmt) do { if (!((DDMP_TN_ABLATE) & (bit))) { stmt; } } while (0)     // 1 MFMA  2 staging (prologue,
#else                                                            // split, ds_write)  4 global loads  8 fragment reads
#define DDMP_TA_(bit, stmt) do { stmt; } while (0)
#endif
#define DDMP_FENCE_() __builtin_amdgcn_sched_barrier(0)

// PP = 1: the two waves of a SIMD run the segments of an iteration in opposite order (see run_dm / run_md)
// TM x TK: the panel of dW a workgroup owns -- 256 x 256, or (the 128 <-> 256 layers) 256 x 128 / 128 x 256: the 8 waves then
// own 64 x 64 each, a 128-wide operand is staged as ONE load per thread and stage (a wave = two rows of 32 lanes)
template <bool PRO, bool GDUAL, int PP, int TM = 256, int TK = 256>
__global__ __launch_bounds__(512) void gemm_tn_rm_kernel(const TnRmArgs a) {
    static_assert((TM == 256 || TM == 128) && (TK == 256 || TK == 128) && TM + TK >= 384, "panel shapes");
    __shared__ __attribute__((aligned(16))) _Float16 Gs[2][2][kRows * TM];      // [buffer][term]
    __shared__ __attribute__((aligned(16))) _Float16 Zs[2][2][kRows * TK];
    __shared__ __attribute__((aligned(16))) float s_co[(GDUAL ? 4 : 0) + (PRO ? 2 : 0) + 1][kT];

    const int n_tiles = a.n_tiles_m * a.n_tiles_k;
    const int xcd = blockIdx.x & (kXcd - 1), local = blockIdx.x >> 3;
    const int tile = local % n_tiles;
    const int split = (local / n_tiles) * kXcd + xcd;
    if (split >= a.n_splits) return;
    const int tm0 = (tile / a.n_tiles_k) * TM, tk0 = (tile % a.n_tiles_k) * TK;
    const int n_rows = a.n_rows, M = a.M, K = a.K;
    const int r_begin = split * a.rows_per_split;
    const int r_end = min(n_rows, r_begin + a.rows_per_split);
    const float slope = a.slope;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int WK = TM == 256 ? 2 : 4, WTK = TK / WK, NJ = WTK / 32;      // waves: (8 / WK) x WK of 64 (m) x WTK (k)
    const int wr = wave / WK, wc = wave % WK;
    const int l31 = lane & 31, lh = lane >> 5;

    float sg = 1.f, sz = 1.f, gmax = 0.f, zmax = 0.f;
    {
        float g_use = a.gslot[0], z_use = a.zslot[0];
        int target = a.target;
        if (a.heal) {                                            // kernel-uniform: nobody writes the slots during this launch
            if ((reinterpret_cast<const unsigned*>(a.gslot)[2] | reinterpret_cast<const unsigned*>(a.zslot)[2]) == 0u) return;
            g_use = a.gslot[1];
            z_use = a.zslot[1];
            target = kF16TargetExact;
        }
        sg = f16s_scale(g_use, target);
        sz = f16s_scale(z_use, target);
    }
    // staging role.  256-wide operand: lane = 4 columns, wave = stage rows (wave, wave + 8): two loads.  128-wide operand: a wave
    // = stage rows (2 wave, 2 wave + 1) of 32 lanes each: one load.  Columns beyond M / K are clamped to valid ones (duplicates
    // that only reach entries never stored).
    constexpr int GP = TM / 128, ZP = TK / 128;                  // loads per thread and stage
    const int c4 = lane * 4;
    const int gc4 = TM == 256 ? c4 : (lane & 31) * 4, zc4 = TK == 256 ? c4 : (lane & 31) * 4;
    const float sgs = sg * slope;
    const int gcol = min(tm0 + gc4, M - 4), zcol = min(tk0 + zc4, K - 4);
    // Coefficient tables in LDS (read per stage: as loop invariants in registers they spill).  The operand scales (powers of
    // two: exact) are folded in -- G: a, b as they are (the gate multiplies by sg or sg * slope), k1 sg, k0 sg; Z: pscale sz, pshift sz (LeakyReLU
    // is positively homogeneous) -- so the conversion needs no separate multiply.
    constexpr int kCoZ = GDUAL ? 4 : 0;
    if (GDUAL && tid < TM / 4) {                                 // (tid = lane here: c4 = the thread's columns of the table)
        const int tcol = min(tm0 + c4, M - 4);
        const float4 ca = *reinterpret_cast<const float4*>(a.ga + tcol), k1 = *reinterpret_cast<const float4*>(a.gk1 + tcol);
        const float4 k0 = *reinterpret_cast<const float4*>(a.gk0 + tcol);
        *reinterpret_cast<float4*>(&s_co[0][c4]) = ca;
        *reinterpret_cast<float4*>(&s_co[1][c4]) = *reinterpret_cast<const float4*>(a.gb + tcol);
        *reinterpret_cast<float4*>(&s_co[GDUAL ? 2 : 0][c4]) = make_float4(k1.x * sg, k1.y * sg, k1.z * sg, k1.w * sg);
        *reinterpret_cast<float4*>(&s_co[GDUAL ? 3 : 0][c4]) = make_float4(k0.x * sg, k0.y * sg, k0.z * sg, k0.w * sg);
    }
    if (PRO && tid >= 64 && tid < 64 + TK / 4) {
        const int tcol = min(tk0 + c4, K - 4);
        const float4 sc = *reinterpret_cast<const float4*>(a.pscale + tcol), sh = *reinterpret_cast<const float4*>(a.pshift + tcol);
        *reinterpret_cast<float4*>(&s_co[kCoZ][c4]) = make_float4(sc.x * sz, sc.y * sz, sc.z * sz, sc.w * sz);
        *reinterpret_cast<float4*>(&s_co[kCoZ + (PRO ? 1 : 0)][c4]) = make_float4(sh.x * sz, sh.y * sz, sh.z * sz, sh.w * sz);
    }
    __syncthreads();

    // Operand rows through buffer loads: descriptor (SGPRs) = the operand from the split's first row to its LAST row, voffset =
    // this lane's column bytes + the wave's row bytes (one v_add per load: the row part is scalar).  The row goes through
    // voffset, not soffset, because only voffset is range-checked: a load beyond the descriptor's range returns 0 without a
    // fault, so rows past the split (and past the matrix) contribute nothing -- zero times anything finite is zero, the other
    // operand needs no masking -- and no 64-bit VGPR address is formed or rewritten while loads are in flight.
    // Only the G operand with the BatchNorm-backward prologue (it does not map 0 to 0) is zeroed by select (MASK), which also
    // keeps its published maximum exact.
    const unsigned ldg4 = (unsigned)(a.ldg * 4), ldg24 = (unsigned)(a.ldg2 * 4), ldz4 = (unsigned)(a.ldz * 4);
    // byte offsets of the lane inside the stage: its columns (+ its row of the wave's two for a 128-wide operand)
    const int voff_g = gcol * 4 + (TM == 128 ? (int)((lane >> 5) * ldg4) : 0), voff_g2 = gcol * 4 + (TM == 128 ? (int)((lane >> 5) * ldg24) : 0);
    const int voff_z = zcol * 4 + (TK == 128 ? (int)((lane >> 5) * ldz4) : 0);
    // (scalar part of the stage row of load p) and the lane's row inside the stage
    auto grow_s = [&](int p) __attribute__((always_inline)) { return TM == 256 ? wave + 8 * p : 2 * wave; };
    auto zrow_s = [&](int p) __attribute__((always_inline)) { return TK == 256 ? wave + 8 * p : 2 * wave; };
    const int grow_l = TM == 256 ? 0 : (lane >> 5), zrow_l = TK == 256 ? 0 : (lane >> 5);
    constexpr bool MASK = GDUAL;
    auto rsrc = [&](const float* base, int64_t ld) {
        const int64_t bytes = (int64_t)(r_end - r_begin) * ld * 4;                       // (< 2^31: launch_tn_rm checks)
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (int64_t)r_begin * ld), 0, (int)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t Gd = rsrc(a.G, a.ldg), Zd = rsrc(a.Z, a.ldz);
    const __amdgpu_buffer_rsrc_t G2d = rsrc(GDUAL ? a.G2 : a.G, GDUAL ? a.ldg2 : a.ldg);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto ldrow = [](__amdgpu_buffer_rsrc_t d, int voff, unsigned row_off) __attribute__((always_inline)) -> float4 {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(d, voff + (int)row_off, 0, 0);
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };

    // Register slots of raw rows.  G (and G2) run TWO stages ahead of their conversion; Z too, except in the three-stream
    // form (ZL = 1: one stage ahead, 8 registers fewer -- with all three streams two stages ahead hipcc runs out of registers
    // in the rotated loop and copies slot registers whose loads are in flight, behind a vmcnt(0)).
    constexpr int ZL = GDUAL ? 1 : 2;
    struct Slot {
        float4 g[GP], z[ZL == 2 ? ZP : 1], g2[GDUAL ? GP : 1];
    };
    Slot S[2];
    float4 Z1[ZP];                                               // ZL = 1: the one Z slot
    auto load_z = [&](float4 (&z)[ZP], int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < ZP; ++p) {
            const unsigned rc = (unsigned)(zrow_s(p) + stage * kRows);
            z[p] = ldrow(Zd, voff_z, rc * ldz4);
            DDMP_FENCE_();
        }
    };
    auto load = [&](Slot& sl, int stage) __attribute__((always_inline)) {
        // (ZL = 1: Z of stage - 1, FIRST -- it is needed one iteration from now, before this call's G rows: the in-order
        //  counter then lets those stay in flight)
        if (ZL == 1) load_z(Z1, max(stage - 1, 0));
        // (fences: the loads keep THIS order everywhere -- g [g2] z of load 0, then of load 1)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (p < GP) {
                const unsigned rc = (unsigned)(grow_s(p) + stage * kRows);                  // scalar
                sl.g[p < GP ? p : 0] = ldrow(Gd, voff_g, rc * ldg4);
                DDMP_FENCE_();
                if (GDUAL) {
                    sl.g2[GDUAL && p < GP ? p : 0] = ldrow(G2d, voff_g2, rc * ldg24);
                    DDMP_FENCE_();
                }
            }
            if (ZL == 2 && p < ZP) {
                const unsigned rc = (unsigned)(zrow_s(p) + stage * kRows);
                sl.z[ZL == 2 && p < ZP ? p : 0] = ldrow(Zd, voff_z, rc * ldz4);
                DDMP_FENCE_();
            }
        }
    };
    // v = the operand times its scale.  No clamp to the f16 range: with a stale scale an element may overflow to inf, but then
    // the wave that saw it raises the slot's flag (f16s_publish) and the heal launch, which follows every stale-scale launch,
    // redoes the whole product with the exact scale and overwrites these partials (gemm_f16s.inc).
    auto split_store = [&](float4 v, _Float16* p0, _Float16* p1) __attribute__((always_inline)) {
        uint2 h, m;                                              // (f16_split_quad: cvt_pk + one v_fma_mix per element)
        f16_split_quad(v, h, m);
        *reinterpret_cast<uint2*>(p0) = h;
        *reinterpret_cast<uint2*>(p1) = m;
    };
    // a plain operand: the scale rides in the conversion instructions (f16_split_quad_scaled), the maximum is taken raw
    auto split_store_scaled = [&](float4 x, float s, _Float16* p0, _Float16* p1) __attribute__((always_inline)) {
        uint2 h, m;
        f16_split_quad_scaled(x, s, h, m);
        *reinterpret_cast<uint2*>(p0) = h;
        *reinterpret_cast<uint2*>(p1) = m;
    };
    // (maxima of the operands as they are converted: SCALED where a prologue carries the scale -- f16s_publish gets those un-scaled,
    //  exactly, the scales are powers of two -- raw for a plain operand)
    auto amax4 = [](float m, float4 v) { return fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w)))); };
    auto store_g = [&](int buf, const Slot& sl, int stage) __attribute__((always_inline)) {
        int co = gc4;
        asm volatile("" : "+v"(co));                             // (keeps the coefficient reads inside the loop)
        __builtin_assume((co & 3) == 0);                         // (... as 16-byte reads: ds_read2_b32 pairs conflict 4-way)
#pragma unroll
        for (int p = 0; p < GP; ++p) {
            float4 x = sl.g[p];
            if (GDUAL) {                                         // = sg * BwdApplyF (bn.hip), element for element
                const float4 y = sl.g2[GDUAL ? p : 0];
                const float4 ca = *reinterpret_cast<const float4*>(&s_co[0][co]);
                const float4 cb = *reinterpret_cast<const float4*>(&s_co[1][co]);
                const float4 k1 = *reinterpret_cast<const float4*>(&s_co[GDUAL ? 2 : 0][co]);
                const float4 k0 = *reinterpret_cast<const float4*>(&s_co[GDUAL ? 3 : 0][co]);
                // sg (a dz lrelu'(a y + b) + k1 y + k0): the gate picks sg or sg * slope (scalars), k1, k0 come scaled
                x.x = fmaf(ca.x, x.x * (fmaf(y.x, ca.x, cb.x) > 0.f ? sg : sgs), fmaf(k1.x, y.x, k0.x));
                x.y = fmaf(ca.y, x.y * (fmaf(y.y, ca.y, cb.y) > 0.f ? sg : sgs), fmaf(k1.y, y.y, k0.y));
                x.z = fmaf(ca.z, x.z * (fmaf(y.z, ca.z, cb.z) > 0.f ? sg : sgs), fmaf(k1.z, y.z, k0.z));
                x.w = fmaf(ca.w, x.w * (fmaf(y.w, ca.w, cb.w) > 0.f ? sg : sgs), fmaf(k1.w, y.w, k0.w));
            }
            if (MASK) {
                const int row = r_begin + stage * kRows + grow_s(p) + grow_l;      // (scalar for a 256-wide operand)
                if (row >= r_end) x = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            gmax = amax4(gmax, x);
            const int o = rm_off<TM>(grow_s(p) + grow_l, gc4);
            if (GDUAL) split_store(x, &Gs[buf][0][o], &Gs[buf][1][o]);
            else split_store_scaled(x, sg, &Gs[buf][0][o], &Gs[buf][1][o]);
        }
    };
    // (PRO: rows past the split read as 0 through the descriptor and the prologue maps them to lrelu(shift) != 0 -- harmless for
    //  the product (their G rows ARE 0) but folded into zmax: with a Z prologue the published Z maximum is an UPPER BOUND that
    //  may include max_k |lrelu(shift_k)| when the split's last stage is ragged; never smaller than the true maximum, so a scale
    //  derived from it cannot overflow -- at worst it is one binade coarser.  Masking here costs 8 VALU per piece in a loop that
    //  is issue-bound, for a bound that is already safe: not done.)
    auto store_z = [&](int buf, const Slot& sl) __attribute__((always_inline)) {
        const float4* zsrc = ZL == 2 ? sl.z : Z1;
        int co = zc4;
        asm volatile("" : "+v"(co));
        __builtin_assume((co & 3) == 0);
#pragma unroll
        for (int p = 0; p < ZP; ++p) {
            float4 x = zsrc[p];
            if (PRO) {
                const float4 sc = *reinterpret_cast<const float4*>(&s_co[kCoZ][co]);
                const float4 sh = *reinterpret_cast<const float4*>(&s_co[kCoZ + (PRO ? 1 : 0)][co]);
                x = f4_affine_lrelu(x, sc, sh, slope);
            }
            zmax = amax4(zmax, x);
            const int o = rm_off<TK>(zrow_s(p) + zrow_l, zc4);
            if (PRO) split_store(x, &Zs[buf][0][o], &Zs[buf][1][o]);
            else split_store_scaled(x, sz, &Zs[buf][0][o], &Zs[buf][1][o]);
        }
    };

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // transpose reads: lane L = lane & 15 of a 16-lane group supplies the address of 4 consecutive columns of row (L >> 2)
    // and receives column (lane & 31) of the tile, rows 0..3 of the 4-row block (gemm_b16.hip)
    const int L = lane & 15, gi = (lane >> 4) & 1;
    auto frag = [&](const _Float16* plane, int cbase, auto tw) __attribute__((always_inline)) -> f16x8 {
        constexpr int T = decltype(tw)::value;
        typedef __fp16 hf4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
        typedef __attribute__((address_space(3))) hf4* lp;
        const int r_lo = lh * 8 + (L >> 2), r_hi = r_lo + 4;
        const int col = cbase + 16 * gi + 4 * (L & 3);
        const f16x4 lo = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(plane + rm_off<T>(r_lo, col))));
        const f16x4 hi = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(plane + rm_off<T>(r_hi, col))));
        f16x8 o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
        o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
        return o;
    };
    // Matrix segment of one stage: the G fragments (2 tiles x 2 terms) stay in registers, the Z fragments are read tile by tile
    // (8 registers live + 8 being read), each feeding its six MFMAs: products a0 b1, a1 b0, a0 b0 (the old kernel's order per
    // accumulator), the two G tiles alternating so that consecutive MFMAs hit different accumulators.
    // NBF = 2: the next Z tile's fragments are requested before this tile's MFMAs (8 more registers); the three-stream form has
    // none to spare (with them hipcc copies slot registers whose loads are in flight -- behind a vmcnt(0)).
    constexpr int NBF = GDUAL ? 1 : 2;
    auto seg_m = [&](int buf) __attribute__((always_inline)) {
        constexpr std::integral_constant<int, TM> twm{};
        constexpr std::integral_constant<int, TK> twk{};
        f16x8 af[2][2], bf[NBF][2];                              // [tile][term]
        if (NBF == 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) DDMP_TA_(8, bf[0][t] = frag(Zs[buf][t], wc * WTK, twk));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) DDMP_TA_(8, af[i][t] = frag(Gs[buf][t], wr * 64 + i * 32, twm));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (NBF == 1 || j < NJ - 1) {
                const int jn = NBF == 1 ? j : j + 1;
#pragma unroll
                for (int t = 0; t < 2; ++t) DDMP_TA_(8, bf[jn % NBF][t] = frag(Zs[buf][t], wc * WTK + jn * 32, twk));
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ta = q == 1 ? 1 : 0, tb = q == 0 ? 1 : 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    DDMP_TA_(1, acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][ta], bf[j % NBF][tb], acc[i][j], 0, 0, 0));
            }
        }
    };

    const int ns = (r_end - r_begin + kRows - 1) / kRows;
    // products in the old kernel's order: a0 b1, a1 b0, a0 b0
    // iteration s (LDS buffer s & 1): slot (s + 1) & 1 holds stage s + 1; it is converted into the other buffer while the
    // products of stage s run, and refilled with stage s + 3
    // The two waves of a SIMD (waves w and w + 4 of the workgroup) run the two segments of an iteration in OPPOSITE order: while
    // one issues its 24 MFMAs back to back (matrix segment: fragments of stage s), the other converts stage s + 1 into the other
    // buffer and requests stage s + 3 (data segment: VALU, LDS writes, loads) -- the segments touch different buffers, both waves
    // meet at the barrier.  In one order for all, both waves of a SIMD want the matrix pipe at the same time and leave it idle
    // together afterwards.
    auto seg_d = [&](int s, int buf, Slot& sl) __attribute__((always_inline)) {
        DDMP_TA_(2, store_g(buf ^ 1, sl, s + 1));
        DDMP_TA_(2, store_z(buf ^ 1, sl));
        DDMP_TA_(4, load(sl, s + 3));
    };
    // Between two barriers every wave runs one matrix segment (stage s) and one data segment (stage s + 1 -> the other buffer).
    // PP: waves 0-3 run them in the order M, D and waves 4-7 (their SIMD partners) in the order D, M.  Both are written as the
    // SAME loop body -- data segment, then matrix segment --: the first group's loop is rotated by half an iteration (its
    // barrier sits between the two segments and M(0) is peeled off in front).  A loop body that STARTS with the fragment reads
    // made hipcc wait vmcnt(0) there, i.e. for the loads issued just before: the one thing this pipeline must not do.
    auto run_dm = [&]() __attribute__((always_inline)) {            // [D(s+1) M(s) | barrier]
        int s = 0;
#pragma unroll 1
        for (; s + 1 < ns; s += 2) {
            seg_d(s, 0, S[1]);
            DDMP_FENCE_();
            seg_m(0);
            lds_barrier();
            seg_d(s + 1, 1, S[0]);
            DDMP_FENCE_();
            seg_m(1);
            lds_barrier();
        }
        if (s < ns) {
            seg_d(s, 0, S[1]);
            DDMP_FENCE_();
            seg_m(0);
            lds_barrier();
        }
    };
    auto run_md = [&]() __attribute__((always_inline)) {            // M(0); [D(s+1) | barrier | M(s+1)]
        seg_m(0);
        int s = 0;
#pragma unroll 1
        for (; s + 2 < ns; s += 2) {
            seg_d(s, 0, S[1]);
            lds_barrier();
            seg_m(1);
            DDMP_FENCE_();
            seg_d(s + 1, 1, S[0]);
            lds_barrier();
            seg_m(0);
            DDMP_FENCE_();
        }
        // one or two stages left: M(s) is done; (the data segments past the end convert rows beyond the split: zeros)
        seg_d(s, 0, S[1]);
        lds_barrier();
        if (s + 1 < ns) {
            seg_m(1);
            DDMP_FENCE_();
            seg_d(s + 1, 1, S[0]);
            lds_barrier();
        }
    };
    // (reads every register of a slot: the compiler waits for its loads here)
    auto settle = [](const Slot& sl) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (p < GP) asm volatile("" ::"v"(sl.g[p < GP ? p : 0].x), "v"(sl.g[p < GP ? p : 0].y), "v"(sl.g[p < GP ? p : 0].z), "v"(sl.g[p < GP ? p : 0].w));
            if (ZL == 2 && p < ZP) asm volatile("" ::"v"(sl.z[ZL == 2 && p < ZP ? p : 0].x), "v"(sl.z[ZL == 2 && p < ZP ? p : 0].y), "v"(sl.z[ZL == 2 && p < ZP ? p : 0].z), "v"(sl.z[ZL == 2 && p < ZP ? p : 0].w));
            if (GDUAL && p < GP) asm volatile("" ::"v"(sl.g2[GDUAL && p < GP ? p : 0].x), "v"(sl.g2[GDUAL && p < GP ? p : 0].y), "v"(sl.g2[GDUAL && p < GP ? p : 0].z), "v"(sl.g2[GDUAL && p < GP ? p : 0].w));
        }
    };
    if (ns > 0) {
        load(S[0], 0);
        if (ZL == 2) {
            load(S[1], 1);
            store_g(0, S[0], 0);
            store_z(0, S[0]);
            load(S[0], 2);
        } else {                                                 // (load(S, stage) also requests Z of stage - 1 into Z1)
            store_g(0, S[0], 0);
            store_z(0, S[0]);                                    // Z1 = stage 0
            load(S[1], 1);                                       // G of stage 1; Z1 <- stage 0 again (harmless)
            load(S[0], 2);                                       // G of stage 2; Z1 <- stage 1
        }
        // The loops are entered with NO load pending as far as the compiler knows: the wait it puts in front of a slot's first
        // use has to hold on every path into the loop, and against the prologue's pending loads (other registers, other order)
        // it came out as vmcnt(0) in every iteration.  One drain per kernel instead.
        settle(S[1]);
        settle(S[0]);
        lds_barrier();
        if (PP == 1 && wave < 4) run_md();                       // (uniform; waves w and w + 4 share a SIMD)
        else run_dm();
    }

    float* o = a.part + (int64_t)split * a.split_stride;
    const float out_scale = (1.f / sg) * (1.f / sz);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = tk0 + wc * WTK + j * 32 + l31;
        if (k >= K) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = tm0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) o[(int64_t)m * a.ld_out + k] = acc[i][j][r] * out_scale;
            }
    }
    if (!a.heal) {
        f16s_publish(a.gslot, GDUAL ? gmax * (1.f / sg) : gmax, sg);
        f16s_publish(a.zslot, PRO ? zmax * (1.f / sz) : zmax, sz);
    }
}

}  // namespace

namespace ddmp {

void launch_tn_rm(const TnRmArgs& a, hipStream_t st) {
    const int n_tiles = a.n_tiles_m * a.n_tiles_k;
    dim3 grid((unsigned)(cdiv(a.n_splits, kXcd) * kXcd * n_tiles)), block(512);
    const bool pro = a.pscale != nullptr, gdual = a.G2 != nullptr;
    // (pp = 0: the same segment order on all waves; measured 6-9 % slower: profiles/r04_tn_kernel_ab.txt)
    constexpr int pp = 1;
#define DDMP_L3_(P_, G_, Q_, TM_, TK_) hipLaunchKernelGGL((gemm_tn_rm_kernel<P_, G_, Q_, TM_, TK_>), grid, block, 0, st, a)
#define DDMP_L_(P_, G_, Q_)                                                                       \
    do {                                                                                          \
        if (a.tm == 256 && a.tk == 256) DDMP_L3_(P_, G_, Q_, 256, 256);                           \
        else if (a.tm == 256) DDMP_L3_(P_, G_, Q_, 256, 128);                                     \
        else DDMP_L3_(P_, G_, Q_, 128, 256);                                                      \
    } while (0)
    // (three operand streams: always the common order -- in the rotated loop hipcc puts a full vmcnt drain in front of the Z
    //  conversion, and measured the two forms tie: profiles/r04_tn_kernel_ab.txt)
    if (gdual) { if (pro) DDMP_L_(true, true, 0); else DDMP_L_(false, true, 0); }
    else if (pp == 0) { if (pro) DDMP_L_(true, false, 0); else DDMP_L_(false, false, 0); }
    else { if (pro) DDMP_L_(true, false, 1); else DDMP_L_(false, false, 1); }
#undef DDMP_L3_
#undef DDMP_L_
}

}  // namespace ddmp
