// Shared definitions for libddmp_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/ddmp_hip.h"
#include "ddmp_internal.h"
#ifdef __cplusplus
#include <cstdlib>
#include <string>
namespace ddmp {
// DDMP_UNFUSE=name[,name...]: fused routes switched OFF for A/B runs and the fused-vs-composed identity tests (read once per
// process; the host mirror reads the same variable: dual-dmp_amd/ops.py `unfused`).  Names used in the library: bnbwd_narrow,
// dgrad_red_narrow; in the engine: stats, gather_bwd, bnbwd_l0, dgrad_red, tail, wprep, bf16_gemm, bf16_spmm_red, equal_width.
inline bool unfused(const char* name) {
    static const std::string v = [] { const char* e = getenv("DDMP_UNFUSE"); return std::string(",") + (e ? e : "") + ","; }();
    return v.size() > 2 && v.find(std::string(",") + name + ",") != std::string::npos;
}
}  // namespace ddmp
#endif

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e__ = (expr);                       \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)

#define LAUNCH_TRY()                                   \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)

#define ARG_TRY(cond)                                  \
    do {                                               \
        if (!(cond)) return DDMP_EINVAL;               \
    } while (0)

namespace ddmp {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kXcd = 8;            // MI355X: 8 XCDs, block b is dispatched to XCD b % 8
constexpr int kCu = 256;

__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// LeakyReLU with the slope as a runtime value (reference: nn.LeakyReLU() = 0.01)
// for 0 <= slope <= 1 (the reference: 0.01) LeakyReLU(x) == max(x, slope*x): one mul + one max, no compare/select
__device__ __forceinline__ float lrelu(float x, float slope) { return fmaxf(x, slope * x); }
__device__ __forceinline__ float lrelu_grad(float x, float slope) { return x > 0.f ? 1.f : slope; }

// Streaming stores: [N, C] outputs of 1-2 GB that nobody reads before they have left every cache.  The nontemporal
// hint measured 2-7 % on the SpMM kernels (1M x 512 face graph 1004 -> 986 us, 0.5M x 512 vertex graph 635 -> 591 us);
// full 16-byte-per-lane row segments only: dword-granular nontemporal stores in the GEMM panel epilogue were 10-20 % slower.
__device__ __forceinline__ void nt_store4(float* p, float4 v) {
#ifdef DDMP_NO_NT                                                // (A/B builds)
    *reinterpret_cast<float4*>(p) = v;
    return;
#endif
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    nt_f4 o = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(o, reinterpret_cast<nt_f4*>(p));
}

__device__ __forceinline__ float4 f4_affine_lrelu(float4 v, float4 a, float4 b, float slope) {
    v.x = lrelu(fmaf(v.x, a.x, b.x), slope);
    v.y = lrelu(fmaf(v.y, a.y, b.y), slope);
    v.z = lrelu(fmaf(v.z, a.z, b.z), slope);
    v.w = lrelu(fmaf(v.w, a.w, b.w), slope);
    return v;
}

// wave-level sum (all 64 lanes end with the total)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Position of this lane's row when the 64 rows of a chunk (lane = row) are ordered by DESCENDING key (stable; key in
// -1 .. 31, -1 = "not a row": last).  The gather kernels process 8 rows per wave step in lockstep, so rows of similar length
// belong in the same step (irregular meshes: a step of 8 rows costs its LONGEST row; sorted, the long rows share a few steps).
// All keys equal (regular meshes): the identity, at the price of two wave reductions.
__device__ __forceinline__ int chunk_rank_desc(int key, int lane) {
    // regular meshes: every key equal -- decided with one ballot, no cross-lane traffic
    if (__ballot(key != __builtin_amdgcn_readfirstlane(key)) == 0ull) return lane;
    int kmax = key, kmin = key;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        kmax = max(kmax, __shfl_xor(kmax, o, 64));
        kmin = min(kmin, __shfl_xor(kmin, o, 64));
    }
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    kmin = __builtin_amdgcn_readfirstlane(kmin);
    int rank = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int v = kmax; v >= kmin; --v) {                         // (uniform: at most 33 rounds)
        const unsigned long long m = __ballot(key == v);
        if (key < v) rank += __popcll(m);
        else if (key == v) rank += __popcll(m & below);
    }
    return rank;
}

// Two-term _Float16 split of (already scaled) floats for the f16x3 GEMMs: h = f16(v) packed by v_cvt_pk_f16_f32, m = f16(v - h) by
// ONE v_fma_mix per element -- the instruction reads the f16 half of h in place, subtracts in f32 (the difference of a float and
// its own f16 rounding is exact) and rounds once: the same bits as (_Float16)(v - (float)h), without the conversion of h back to
// f32 and without the separate packing of m.  Left to itself hipcc builds half of the elements that way and the other half as
// cvt_f32_f16 + pk_fma + cvt_pk: 3.7 VALU operations per element in the wide wgrad's loop instead of 2.5, in kernels that are
// short of issue slots.  Four elements per asm statement, the two halves of a register written by instructions that
// are NOT adjacent: a partial (op_sel) write followed at once by a read of the same register costs a wait state (hipcc pads
// such pairs with s_nop when it sees them; inside one statement nobody does, so the order below keeps them apart and the
// statement ends with the wait state its last write may owe the next reader).
__device__ __forceinline__ void f16_split_quad(float4 v, uint2& h, uint2& m) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    h2_t a, b;
    a[0] = (_Float16)v.x; a[1] = (_Float16)v.y;
    b[0] = (_Float16)v.z; b[1] = (_Float16)v.w;
    h.x = __builtin_bit_cast(unsigned, a);
    h.y = __builtin_bit_cast(unsigned, b);
    asm("v_fma_mixlo_f16 %0, %2, 1.0, -%6 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %4, 1.0, -%7 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, 1.0, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %5, 1.0, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 0"
        : "=&v"(m.x), "=&v"(m.y) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(h.x), "v"(h.y));
}
// The same split of x * s for a power-of-two scale s (the products are exact): the scale rides in the instructions' second
// source and no scaled copy of the operand is ever formed -- two v_fma_mix per element.
__device__ __forceinline__ void f16_split_quad_scaled(float4 x, float s, uint2& h, uint2& m) {
    asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
        "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
        "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
        "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 0"
        : "=&v"(h.x), "=&v"(h.y), "=&v"(m.x), "=&v"(m.y) : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w), "v"(s));
}

// Sum over the 8 lanes l, l ^ 8, l ^ 16, l ^ 32 ... that share (lane & 7) -- the 8 row groups of a wave in the gather kernels'
// fused-reduction epilogues.  __shfl_xor compiles to ds_bpermute_b32 (the LDS crossbar: 24 of them per wave and 128-byte slab
// were a third of the statistics epilogue's cost on the 7-entry vertex graph); here one DPP add within the 16-lane row and the
// two gfx950 lane-swap instructions.  Same pairings in the same order as the shuffle tree: bit-identical sums.
__device__ __forceinline__ float group8_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /*row_ror:8*/, 0xf, 0xf, false));
    unsigned a = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);     // rows 0 <-> 1, 2 <-> 3 (16-lane rows)
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    a = __float_as_uint(v);
    const auto t = __builtin_amdgcn_permlane32_swap(a, a, false, false);     // lower <-> upper half
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

// block-level sum of one double per thread; result valid in thread 0. `sm` holds >= blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sm[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) t += sm[i];
    }
    return t;
}

}  // namespace ddmp

// the CSR graph handle (opaque in the C ABI)
struct ddmp_graph {
    int64_t n_rows;     // rows that are aggregated (owned nodes)
    int64_t n_cols;     // nodes that can be referenced by col (owned + halo); >= n_rows
    int64_t nnz;        // entries, self loops included
    int32_t* rowptr;    // device [n_rows + 1]
    int32_t* col;       // device [nnz]
    float* dinv;        // device [n_cols]   deg^-1/2 (deg counts the self loop)
    float* ew;          // device [nnz]: dinv[col[e]], the entry's weight as the gather kernels stage it (no col -> dinv chain per chunk)
    float* dinv_r;      // = dinv + row0: the factor of output row i (row0 > 0: a row SLICE of a local graph, see ddmp_graph_create_csr_rows_host)
    int max_row_nnz;
    // LDS-patch gather (spmm_patch.hip).  Per 64-row chunk: the sorted unique column ids it references ("patch") and, per CSR
    // entry, the index of its column in that list.  Selection is PER CHUNK: a chunk whose patch does not fit the kernel's
    // buffers (more than 32 * patch_kd distinct columns), whose CSR slice does not fit its LDS tables, or that has no entries
    // at all is "heavy": its pl_ptr span is empty, the patch kernel skips it and the lean gather processes the `heavy` list
    // in a second (small) launch -- one hub vertex does not move a whole graph to another kernel.
    int32_t* pl_ptr;    // device [n_chunks + 1]
    int32_t* pl_col;    // device [pl_ptr[n_chunks]]
    uint16_t* lcol;     // device [nnz]
    int max_chunk_nnz;  // most CSR entries of a chunk the patch kernel takes (its LDS entry tables are sized for it)
    int max_patch;      // largest patch among the chunks the patch kernel takes (0: tables not built)
    int patch_kd;       // patch rows per LDS buffer / 32 (3..6), chosen so that at most ~1 % of the chunks are heavy
    // round 6: a chunk whose patch is too large but whose 2 halves (32 rows) or 4 quarters (16 rows) each fit is "split": the patch
    // kernel walks it in as many passes.  pl_split[2c + 1] = parts (0: not split), pl_split[2c] = index IN pl_split of the chunk's
    // (patch start in pl_col, record slot of the part's fused sums) pairs for parts 1 ..; record slots are n_chunks + 0 .. n_split - 1.
    // On a regular mesh that leaves no heavy chunk, i.e. no second launch per gather
    int32_t* pl_split;  // device [2 * n_chunks + 2 * n_split] (nullptr: no split chunk)
    int n_split;        // extra record slots = sum over the split chunks of (parts - 1)
    int32_t* heavy;     // device [n_heavy]: the heavy chunks, ascending
    int n_heavy;
};
namespace ddmp {
constexpr int kChunkRows = 64;
// LDS-patch SpMM (spmm_patch.hip), shared by the float32 and bfloat16 entry points: DDMP_OK / an error / "take the slab kernel"
constexpr int kPatchNotApplicable = -100;
// spmm_patch.hip: the caller takes its slab / lean kernel when this returns kPatchNotApplicable (selection: patch_mode there)
int spmm_patch(const ddmp_graph* g, const void* X, int64_t ldx, void* Y, int64_t ldy, int C, int dtype, const float* bias,
               const float* ps, const float* psh, float slope, const void* red_Yp, int64_t red_ldyp, const float* red_scale,
               const float* red_shift, const float* red_mean, const float* red_rstd, float* red_part, hipStream_t st);
// the BatchNorm backward on the gather (ddmp_spmm_bnbwd_f32), LDS-patch form; heavy chunks are the caller's
int spmm_patch_bwd(const ddmp_graph* g, const void* dZ, int64_t lddz, const void* Yb, int64_t ldyb, void* out, int64_t ld_out, int C,
                   int dtype, const float* a, const float* b, const float* c1, const float* c0, float slope, hipStream_t st);
}
