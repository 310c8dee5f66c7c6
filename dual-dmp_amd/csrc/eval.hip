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
// ---- vertex update from predicted normals (util/models.py:31-44).  Per sweep: c_f = centroid of the current
// positions; p_v += (1/|F(v)|) sum_{f in F(v)} ((c_f - p_v) . n_f) n_f.  The reference updates vertices one by one
// in a Python loop, but each update reads only the centroids computed BEFORE the loop and the vertex's own
// position, so the sweep is order-independent and a parallel pass is the same computation.
__global__ __launch_bounds__(256) void face_center_kernel(int F, const float* __restrict__ pos,
                                                          const int* __restrict__ faces, float* __restrict__ fc) {
    for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += gridDim.x * 256) {
        const int i0 = faces[3 * (int64_t)f], i1 = faces[3 * (int64_t)f + 1], i2 = faces[3 * (int64_t)f + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            fc[3 * (int64_t)f + c] = (pos[3 * (int64_t)i0 + c] + pos[3 * (int64_t)i1 + c] + pos[3 * (int64_t)i2 + c]) / 3.0f;
    }
}

__global__ __launch_bounds__(256) void vertex_update_kernel(int V, float* __restrict__ pos, const float* __restrict__ fc,
                                                            const float* __restrict__ nrm, const int* __restrict__ vf_ptr,
                                                            const int* __restrict__ vf_corner) {
    for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
        const float px = pos[3 * (int64_t)v], py = pos[3 * (int64_t)v + 1], pz = pos[3 * (int64_t)v + 2];
        float dx = 0.f, dy = 0.f, dz = 0.f;
        const int b = vf_ptr[v], e = vf_ptr[v + 1];
        for (int k = b; k < e; ++k) {
            const int f = vf_corner[k] / 3;
            const float nx = nrm[3 * (int64_t)f], ny = nrm[3 * (int64_t)f + 1], nz = nrm[3 * (int64_t)f + 2];
            const float d = nx * (fc[3 * (int64_t)f] - px) + ny * (fc[3 * (int64_t)f + 1] - py) + nz * (fc[3 * (int64_t)f + 2] - pz);
            dx += d * nx; dy += d * ny; dz += d * nz;
        }
        const float cnt = (float)(e - b);
        pos[3 * (int64_t)v] = px + dx / cnt;
        pos[3 * (int64_t)v + 1] = py + dy / cnt;
        pos[3 * (int64_t)v + 2] = pz + dz / cnt;
    }
}
}  // namespace

extern "C" int ddmp_vertex_update_f32(int64_t V, int64_t F, float* pos, const float* norm, const int32_t* faces,
                                      const int32_t* vf_ptr, const int32_t* vf_corner, float* fc_scratch, int loop,
                                      ddmp_stream stream) {
    ARG_TRY(V > 0 && V < INT32_MAX / 3 && F > 0 && F < INT32_MAX / 3 && pos && norm && faces && vf_ptr && vf_corner);
    ARG_TRY(fc_scratch && loop >= 0);
    hipStream_t st = (hipStream_t)stream;
    const int gf = (int)std::min<int64_t>(cdiv(F, 256), 2048), gv = (int)std::min<int64_t>(cdiv(V, 256), 2048);
    for (int it = 0; it < loop; ++it) {
        hipLaunchKernelGGL(face_center_kernel, dim3(gf), dim3(256), 0, st, (int)F, pos, faces, fc_scratch);
        LAUNCH_TRY();
        hipLaunchKernelGGL(vertex_update_kernel, dim3(gv), dim3(256), 0, st, (int)V, pos, fc_scratch, norm, vf_ptr,
                           vf_corner);
        LAUNCH_TRY();
    }
    return DDMP_OK;
}

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
