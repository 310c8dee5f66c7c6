// On-device evaluation block (main.py:117-123 of the reference): face normals of the predicted positions
// (util/mesh.py:87-92; the reference feeds float32 positions, so the cross product / normalisation run in
// float32 there too) and the mean angular difference to ground-truth normals (util/loss.py:261-272: inner
// product, clip, arccos, degrees, mean -- float64).  Replaces a D2H copy + numpy pass every 10 iterations.
#include "ddmp_common.h"

#include <algorithm>

namespace {
using namespace ddmp;
constexpr int kNB = 256;

__global__ __launch_bounds__(256) void face_normals_kernel(int F, const float* __restrict__ pos,
                                                           const int* __restrict__ faces, float* __restrict__ fn,
                                                           float* __restrict__ fa) {
    for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += gridDim.x * 256) {
        const int i0 = faces[3 * (int64_t)f], i1 = faces[3 * (int64_t)f + 1], i2 = faces[3 * (int64_t)f + 2];
        const float ax = pos[3 * (int64_t)i1] - pos[3 * (int64_t)i0], ay = pos[3 * (int64_t)i1 + 1] - pos[3 * (int64_t)i0 + 1],
                    az = pos[3 * (int64_t)i1 + 2] - pos[3 * (int64_t)i0 + 2];
        const float bx = pos[3 * (int64_t)i2] - pos[3 * (int64_t)i0], by = pos[3 * (int64_t)i2 + 1] - pos[3 * (int64_t)i0 + 1],
                    bz = pos[3 * (int64_t)i2 + 2] - pos[3 * (int64_t)i0 + 2];
        const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
        const float len = sqrtf(cx * cx + cy * cy + cz * cz);
        const float nrm = len + 1e-24f;
        fn[3 * (int64_t)f] = cx / nrm;
        fn[3 * (int64_t)f + 1] = cy / nrm;
        fn[3 * (int64_t)f + 2] = cz / nrm;
        if (fa) fa[f] = 0.5f * len;
    }
}

__global__ __launch_bounds__(256) void mad_kernel(int F, const float* __restrict__ n1, const double* __restrict__ n2,
                                                  double* __restrict__ partial) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += kNB * 256) {
        double inner = (double)n1[3 * (int64_t)f] * n2[3 * (int64_t)f] + (double)n1[3 * (int64_t)f + 1] * n2[3 * (int64_t)f + 1] +
                       (double)n1[3 * (int64_t)f + 2] * n2[3 * (int64_t)f + 2];
        inner = inner < -1.0 ? -1.0 : (inner > 1.0 ? 1.0 : inner);
        s += acos(inner) * (180.0 / 3.14159265358979323846);
    }
    const double t = block_sum(s, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void mad_final_kernel(const double* __restrict__ partial, int F, double* __restrict__ out) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < kNB; i += blockDim.x) s += partial[i];
    const double t = block_sum(s, sm);
    if (threadIdx.x == 0) out[0] = t / (double)F;
}
}  // namespace

extern "C" int ddmp_face_normals_f32(int64_t F, const float* pos, const int32_t* faces, float* fn, float* fa,
                                     ddmp_stream stream) {
    ARG_TRY(F > 0 && F < INT32_MAX / 3 && pos && faces && fn);
    const int grid = (int)std::min<int64_t>(cdiv(F, 256), 2048);
    hipLaunchKernelGGL(face_normals_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (int)F, pos, faces, fn, fa);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" size_t ddmp_mad_workspace_bytes(void) { return sizeof(double) * kNB; }

extern "C" int ddmp_mad_f64(int64_t F, const float* n1, const double* n2, double* out, void* ws, size_t ws_bytes,
                            ddmp_stream stream) {
    ARG_TRY(F > 0 && F < INT32_MAX / 3 && n1 && n2 && out);
    if (!ws || ws_bytes < sizeof(double) * kNB) return DDMP_EWORKSPACE;
    hipLaunchKernelGGL(mad_kernel, dim3(kNB), dim3(256), 0, (hipStream_t)stream, (int)F, n1, n2, (double*)ws);
    LAUNCH_TRY();
    hipLaunchKernelGGL(mad_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, (int)F, out);
    LAUNCH_TRY();
    return DDMP_OK;
}
