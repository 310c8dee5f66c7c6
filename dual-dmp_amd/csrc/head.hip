// Output heads of PosNet / NormalNet (util/networks.py:64-67 and :125-129 of the reference):
//   z  = LeakyReLU(bn12(conv12))            [n,32]   (applied here as a load prologue: scale/shift)
//   t  = LeakyReLU(linear1(z))              [n,16]
//   u  = linear2(t)                         [n,3]
//   PosNet   : out = x_pos + u
//   NormalNet: v = tanh(u);  out = v * 1/(||v||_2 + 1e-12)
// One thread per node; weights are wave-uniform scalar loads.  Backward recomputes the forward from
// the saved conv12 output, writes dZ [n,32] and reduces the four parameter gradients through LDS
// tiles of 256 nodes (no atomics; per-block partials + a final reduction -> deterministic).
#include "b16_common.h"

#include <algorithm>

namespace {

using namespace ddmp;

constexpr int H0 = 32, H1 = 16, H2 = 3;
constexpr int kNPar = H1 * H0 + H1 + H2 * H1 + H2;   // 579: dW1 | db1 | dW2 | db2

struct HeadW {
    const float *W1, *b1, *W2, *b2;
};

// the conv12 output row: float32 or bfloat16 features (b16_common.h)
__device__ __forceinline__ void load_row32(const float* __restrict__ yrow, float (&y)[H0]) {
#pragma unroll
    for (int q = 0; q < H0 / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(yrow + q * 4);
        y[q * 4 + 0] = v.x; y[q * 4 + 1] = v.y; y[q * 4 + 2] = v.z; y[q * 4 + 3] = v.w;
    }
}
__device__ __forceinline__ void load_row32(const bf16_t* __restrict__ yrow, float (&y)[H0]) {
#pragma unroll
    for (int q = 0; q < H0 / 8; ++q) {
        float t[8];
        bf_unpack8(ld8b(yrow + q * 8), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[q * 8 + e] = t[e];
    }
}
__device__ __forceinline__ void store_quad(float* p, const float (&d)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]);
}
__device__ __forceinline__ void store_quad(bf16_t* p, const float (&d)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(bf_pack(d[0], d[1]), bf_pack(d[2], d[3]));
}

template <typename T>
__device__ __forceinline__ void head_forward_row(const T* __restrict__ yrow, const float* __restrict__ scale,
                                                 const float* __restrict__ shift, float slope, const HeadW& w,
                                                 float (&z)[H0], float (&tp)[H1], float (&u)[H2]) {
    load_row32(yrow, z);
#pragma unroll
    for (int j = 0; j < H0; ++j) z[j] = lrelu(fmaf(z[j], scale[j], shift[j]), slope);
#pragma unroll
    for (int i = 0; i < H1; ++i) {
        float s = w.b1[i];
#pragma unroll
        for (int j = 0; j < H0; ++j) s = fmaf(w.W1[i * H0 + j], z[j], s);
        tp[i] = s;   // pre-activation
    }
#pragma unroll
    for (int o = 0; o < H2; ++o) {
        float s = w.b2[o];
#pragma unroll
        for (int i = 0; i < H1; ++i) s = fmaf(w.W2[o * H1 + i], lrelu(tp[i], slope), s);
        u[o] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ Y, int64_t ldy, int n_rows,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float slope, HeadW w,
                                                       int kind, const float* __restrict__ x_pos,
                                                       float* __restrict__ out) {
    for (int row = blockIdx.x * 256 + threadIdx.x; row < n_rows; row += gridDim.x * 256) {
        float z[H0], tp[H1], u[H2];
        head_forward_row(Y + (int64_t)row * ldy, scale, shift, slope, w, z, tp, u);
        float* o = out + (int64_t)row * 3;
        if (kind == 0) {
            const float* xp = x_pos + (int64_t)row * 3;
            o[0] = xp[0] + u[0];
            o[1] = xp[1] + u[1];
            o[2] = xp[2] + u[2];
        } else {
            const float v0 = tanhf(u[0]), v1 = tanhf(u[1]), v2 = tanhf(u[2]);
            const float r = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
            const float s = 1.0f / (r + 1.0e-12f);
            o[0] = v0 * s;
            o[1] = v1 * s;
            o[2] = v2 * s;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ Y, int64_t ldy, int n_rows,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float slope, HeadW w,
                                                       int kind, const float* __restrict__ dout,
                                                       T* __restrict__ dZ, int64_t lddz,
                                                       float* __restrict__ partial /*[grid][kNPar]*/) {
    __shared__ float zs[256][H0 + 1];
    __shared__ float dtps[256][H1 + 1];
    __shared__ float ts[256][H1 + 1];
    __shared__ float dus[256][4];
    const int tid = threadIdx.x;
    // reduction roles
    const int w1_j = tid & 31, w1_i = tid >> 5;            // entries (w1_i, w1_j) and (w1_i + 8, w1_j)
    float accA = 0.f, accB = 0.f;                          // dW1 pair
    float accC = 0.f;                                      // dW2 (tid < 48) | db1 (64..79) | db2 (128..130)

    const int n_tiles = (n_rows + 255) / 256;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row = tile * 256 + tid;
        float z[H0], tp[H1], u[H2], du[H2], dtp[H1];
        if (row < n_rows) {
            head_forward_row(Y + (int64_t)row * ldy, scale, shift, slope, w, z, tp, u);
            const float* g = dout + (int64_t)row * 3;
            if (kind == 0) {
                du[0] = g[0]; du[1] = g[1]; du[2] = g[2];
            } else {
                const float v0 = tanhf(u[0]), v1 = tanhf(u[1]), v2 = tanhf(u[2]);
                const float r = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
                const float s = 1.0f / (r + 1.0e-12f);
                const float gv = g[0] * v0 + g[1] * v1 + g[2] * v2;
                const float k = r > 0.f ? s * s * gv / r : 0.f;
                du[0] = (s * g[0] - k * v0) * (1.f - v0 * v0);
                du[1] = (s * g[1] - k * v1) * (1.f - v1 * v1);
                du[2] = (s * g[2] - k * v2) * (1.f - v2 * v2);
            }
#pragma unroll
            for (int i = 0; i < H1; ++i) {
                float s = 0.f;
#pragma unroll
                for (int o = 0; o < H2; ++o) s = fmaf(w.W2[o * H1 + i], du[o], s);
                dtp[i] = s * lrelu_grad(tp[i], slope);
            }
            T* dz = dZ + (int64_t)row * lddz;
#pragma unroll
            for (int q = 0; q < H0 / 4; ++q) {
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < H1; ++i) s = fmaf(w.W1[i * H0 + q * 4 + e], dtp[i], s);
                    d[e] = s;
                }
                store_quad(dz + q * 4, d);
            }
        } else {
#pragma unroll
            for (int j = 0; j < H0; ++j) z[j] = 0.f;
#pragma unroll
            for (int i = 0; i < H1; ++i) { tp[i] = 0.f; dtp[i] = 0.f; }
            du[0] = du[1] = du[2] = 0.f;
        }
        __syncthreads();   // previous tile's reduction reads are done
#pragma unroll
        for (int j = 0; j < H0; ++j) zs[tid][j] = z[j];
#pragma unroll
        for (int i = 0; i < H1; ++i) {
            dtps[tid][i] = dtp[i];
            ts[tid][i] = row < n_rows ? lrelu(tp[i], slope) : 0.f;
        }
        dus[tid][0] = du[0]; dus[tid][1] = du[1]; dus[tid][2] = du[2];
        __syncthreads();
        for (int r = 0; r < 256; ++r) {
            const float zj = zs[r][w1_j];
            accA = fmaf(dtps[r][w1_i], zj, accA);
            accB = fmaf(dtps[r][w1_i + 8], zj, accB);
        }
        if (tid < H2 * H1) {
            const int o = tid / H1, i = tid % H1;
            for (int r = 0; r < 256; ++r) accC = fmaf(dus[r][o], ts[r][i], accC);
        } else if (tid >= 64 && tid < 64 + H1) {
            for (int r = 0; r < 256; ++r) accC += dtps[r][tid - 64];
        } else if (tid >= 128 && tid < 128 + H2) {
            for (int r = 0; r < 256; ++r) accC += dus[r][tid - 128];
        }
    }
    float* p = partial + (int64_t)blockIdx.x * kNPar;
    p[w1_i * H0 + w1_j] = accA;
    p[(w1_i + 8) * H0 + w1_j] = accB;
    if (tid < H2 * H1) p[H1 * H0 + H1 + tid] = accC;
    else if (tid >= 64 && tid < 64 + H1) p[H1 * H0 + (tid - 64)] = accC;
    else if (tid >= 128 && tid < 128 + H2) p[H1 * H0 + H1 + H2 * H1 + (tid - 128)] = accC;
}

// 32 parameters x 8 block-slices per workgroup
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ partial, int nblk,
                                                          float* __restrict__ dW1, float* __restrict__ db1,
                                                          float* __restrict__ dW2, float* __restrict__ db2) {
    __shared__ double sm[256];
    const int i = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
    double s = 0.0;
    if (i < kNPar)
        for (int b = sl; b < nblk; b += 8) s += (double)partial[(int64_t)b * kNPar + i];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (sl != 0 || i >= kNPar) return;
    double r = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) r += sm[threadIdx.x + 32 * u];
    const float v = (float)r;
    if (i < H1 * H0) dW1[i] = v;
    else if (i < H1 * H0 + H1) db1[i - H1 * H0] = v;
    else if (i < H1 * H0 + H1 + H2 * H1) dW2[i - H1 * H0 - H1] = v;
    else db2[i - H1 * H0 - H1 - H2 * H1] = v;
}

int head_bwd_blocks(int64_t n_rows) { return (int)std::max<int64_t>(1, std::min<int64_t>(512, cdiv(n_rows, 256))); }

}  // namespace

extern "C" int ddmp_head_fwd_f32(const float* Y, int64_t ldy, int64_t n_rows, const float* scale,
                                 const float* shift, float slope, const float* W1, const float* b1,
                                 const float* W2, const float* b2, int kind, const float* x_pos, float* out,
                                 ddmp_stream stream) {
    ARG_TRY(Y && scale && shift && W1 && b1 && W2 && b2 && out && n_rows > 0 && n_rows < INT32_MAX);
    ARG_TRY(ldy >= H0 && ldy % 4 == 0 && (kind == 0 || kind == 1) && (kind == 1 || x_pos));
    const int grid = (int)std::min<int64_t>(cdiv(n_rows, 256), 256 * 8);
    HeadW w{W1, b1, W2, b2};
    hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, Y, ldy, (int)n_rows, scale,
                       shift, slope, w, kind, x_pos, out);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" size_t ddmp_head_bwd_workspace_bytes(int64_t n_rows) {
    if (n_rows <= 0) return 0;
    return (size_t)head_bwd_blocks(n_rows) * kNPar * sizeof(float);
}

extern "C" int ddmp_head_bwd_f32(const float* Y, int64_t ldy, int64_t n_rows, const float* scale,
                                 const float* shift, float slope, const float* W1, const float* b1,
                                 const float* W2, const float* b2, int kind, const float* dout, float* dZ,
                                 int64_t lddz, float* dW1, float* db1, float* dW2, float* db2, void* ws,
                                 size_t ws_bytes, ddmp_stream stream) {
    ARG_TRY(Y && scale && shift && W1 && b1 && W2 && b2 && dout && dZ && dW1 && db1 && dW2 && db2);
    ARG_TRY(n_rows > 0 && n_rows < INT32_MAX && ldy >= H0 && ldy % 4 == 0 && lddz >= H0 && lddz % 4 == 0);
    ARG_TRY(kind == 0 || kind == 1);
    const int nblk = head_bwd_blocks(n_rows);
    if (!ws || ws_bytes < (size_t)nblk * kNPar * sizeof(float)) return DDMP_EWORKSPACE;
    HeadW w{W1, b1, W2, b2};
    hipLaunchKernelGGL(head_bwd_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, Y, ldy, (int)n_rows, scale,
                       shift, slope, w, kind, dout, dZ, lddz, (float*)ws);
    LAUNCH_TRY();
    hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)cdiv(kNPar, 32)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, nblk, dW1, db1, dW2, db2);
    LAUNCH_TRY();
    return DDMP_OK;
}


// ---- bfloat16 features: Y (conv12 output) and dZ are bf16 [n,32]; out / dout / parameters stay float32
extern "C" int ddmp_head_fwd_bf16(const uint16_t* Y, int64_t ldy, int64_t n_rows, const float* scale, const float* shift,
                                  float slope, const float* W1, const float* b1, const float* W2, const float* b2,
                                  int kind, const float* x_pos, float* out, ddmp_stream stream) {
    ARG_TRY(Y && scale && shift && W1 && b1 && W2 && b2 && out && n_rows > 0 && n_rows < INT32_MAX);
    ARG_TRY(ldy >= H0 && ldy % 8 == 0 && b16_aligned(Y) && (kind == 0 || kind == 1) && (kind == 1 || x_pos));
    const int grid = (int)std::min<int64_t>(cdiv(n_rows, 256), 256 * 8);
    HeadW w{W1, b1, W2, b2};
    hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, Y, ldy, (int)n_rows, scale,
                       shift, slope, w, kind, x_pos, out);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_head_bwd_bf16(const uint16_t* Y, int64_t ldy, int64_t n_rows, const float* scale, const float* shift,
                                  float slope, const float* W1, const float* b1, const float* W2, const float* b2,
                                  int kind, const float* dout, uint16_t* dZ, int64_t lddz, float* dW1, float* db1,
                                  float* dW2, float* db2, void* ws, size_t ws_bytes, ddmp_stream stream) {
    ARG_TRY(Y && scale && shift && W1 && b1 && W2 && b2 && dout && dZ && dW1 && db1 && dW2 && db2);
    ARG_TRY(n_rows > 0 && n_rows < INT32_MAX && ldy >= H0 && ldy % 8 == 0 && lddz >= H0 && lddz % 8 == 0);
    ARG_TRY((kind == 0 || kind == 1) && b16_aligned(Y) && b16_aligned(dZ));
    const int nblk = head_bwd_blocks(n_rows);
    if (!ws || ws_bytes < (size_t)nblk * kNPar * sizeof(float)) return DDMP_EWORKSPACE;
    HeadW w{W1, b1, W2, b2};
    hipLaunchKernelGGL(head_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, Y, ldy, (int)n_rows, scale,
                       shift, slope, w, kind, dout, dZ, lddz, (float*)ws);
    LAUNCH_TRY();
    hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)cdiv(kNPar, 32)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, nblk, dW1, db1, dW2, db2);
    LAUNCH_TRY();
    return DDMP_OK;
}
