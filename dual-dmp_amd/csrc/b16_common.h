// bf16-feature mode (BASELINE.json configs[1], SURVEY.md §8b `dtype`): node features [N, C] are stored as bfloat16
// (row = 2C bytes, 16-byte aligned: C % 8 == 0), every kernel unpacks to float32, does its arithmetic and its
// accumulation in float32 (BatchNorm statistics in float64) and rounds to nearest-even once, on the store.
// Parameters, their gradients and the optimizer state stay float32.
#pragma once
#include "ddmp_common.h"

namespace ddmp {

typedef uint16_t bf16_t;                                              // raw bits in the C ABI
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// two floats -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned bf_pack(float lo, float hi) {
    bf16x2 p;
    p[0] = (__bf16)lo;
    p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf_round(float x) { return (float)(__bf16)x; }

__device__ __forceinline__ void bf_unpack8(uint4 u, float (&f)[8]) {
    f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
    f[4] = bf_lo(u.z); f[5] = bf_hi(u.z); f[6] = bf_lo(u.w); f[7] = bf_hi(u.w);
}
__device__ __forceinline__ uint4 bf_pack8(const float (&f)[8]) {
    return make_uint4(bf_pack(f[0], f[1]), bf_pack(f[2], f[3]), bf_pack(f[4], f[5]), bf_pack(f[6], f[7]));
}
__device__ __forceinline__ void ld8f(const float* p, float (&f)[8]) {           // 8 float32 coefficients (32-byte aligned)
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ uint4 ld8b(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st8b(bf16_t* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ void nt_st8b(bf16_t* p, uint4 v) {                   // streaming store (see nt_store4)
    typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
    nt_u4 o = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(o, reinterpret_cast<nt_u4*>(p));
}

// global -> LDS copy of 16 bytes per lane (LDS destination = wave-uniform base + lane * 16) issued from inline asm.
// The builtin (__builtin_amdgcn_global_load_lds) is tracked by hipcc's waitcnt pass as a pending LDS write: every later
// ds_read / ds_write it cannot prove disjoint from the destination -- any run-time LDS offset -- is preceded by a
// vmcnt(0), i.e. a drain of the whole copy pipeline (measured: 2x on spmm_patch2_kernel).  From asm the copy is invisible
// to that pass; the kernels that use it order their LDS reads behind the copies themselves: counted s_waitcnt vmcnt(N),
// then s_barrier, then the reads (cdna_hip_programming.md §5.7).  M0 (the destination base) is saved and restored
// inside the statement.
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
    const unsigned dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds_dst);
    const unsigned dst_u = __builtin_amdgcn_readfirstlane(dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst_u)
                 : "memory");
}

__host__ inline bool b16_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace ddmp
