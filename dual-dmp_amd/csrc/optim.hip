// clip_grad_norm_ + Adam over a flat float32 parameter arena (main.py:108-110 of the reference:
// nn.utils.clip_grad_norm_(normnet.parameters(), 0.8) then two torch.optim.Adam(lr=.01).step()).
// All parameters of a net live in ONE contiguous buffer (same for grad, exp_avg, exp_avg_sq), so the
// reference's ~100 small per-tensor kernels become one reduction and one streaming update.
#include "ddmp_common.h"

#include <algorithm>
#include <cmath>

namespace {

using namespace ddmp;

constexpr int kNB = 256;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n,
                                                    double* __restrict__ partial) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)kNB * 256) {
        const double v = g[i];
        s = fma(v, v, s);
    }
    const double t = block_sum(s, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void sumsq_final_kernel(const double* __restrict__ partial, double* __restrict__ out) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < kNB; i += blockDim.x) s += partial[i];
    const double t = block_sum(s, sm);
    if (threadIdx.x == 0) out[0] = t;
}

__device__ __forceinline__ float clip_coef(const double* sumsq, float max_norm) {
    if (!sumsq) return 1.f;
    const float total = (float)sqrt(sumsq[0]);
    const float c = max_norm / (total + 1.0e-6f);     // torch: clip_coef = max_norm / (total_norm + 1e-6)
    return c < 1.f ? c : 1.f;                          //        clamped to <= 1
}

__global__ __launch_bounds__(256) void scale_by_clip_kernel(float* __restrict__ g, int64_t n,
                                                            const double* __restrict__ sumsq, float max_norm) {
    const float c = clip_coef(sumsq, max_norm);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) g[i] *= c;
}

// device-side step counter (hipGraph replay: the launch arguments are frozen, the step count is not):
// ++counter, coef = { lr / (1 - b1^t), sqrt(1 - b2^t) }
__global__ void adam_prepare_kernel(int* __restrict__ counter, float lr, float b1, float b2, float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int t = counter[0] + 1;
    counter[0] = t;
    coef[0] = (float)((double)lr / (1.0 - pow((double)b1, (double)t)));
    coef[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float b1, float b2, float eps, float step_size,
                                                   float bc2_sqrt, const double* __restrict__ sumsq,
                                                   float max_norm, const float* __restrict__ coef_dev) {
    if (coef_dev) {
        step_size = coef_dev[0];
        bc2_sqrt = coef_dev[1];
    }
    const float c = clip_coef(sumsq, max_norm);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * c;
        const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

}  // namespace

extern "C" size_t ddmp_sumsq_workspace_bytes(void) { return sizeof(double) * kNB; }

extern "C" int ddmp_grad_sumsq_f32(const float* g, int64_t n, double* out, void* ws, size_t ws_bytes,
                                   ddmp_stream stream) {
    ARG_TRY(g && out && n > 0);
    if (!ws || ws_bytes < sizeof(double) * kNB) return DDMP_EWORKSPACE;
    hipLaunchKernelGGL(sumsq_kernel, dim3(kNB), dim3(256), 0, (hipStream_t)stream, g, n, (double*)ws);
    LAUNCH_TRY();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, out);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_grad_clip_f32(float* g, int64_t n, const double* sumsq, float max_norm, ddmp_stream stream) {
    ARG_TRY(g && sumsq && n > 0 && max_norm > 0.f);
    const int grid = (int)std::min<int64_t>(cdiv(n, 256), 2048);
    hipLaunchKernelGGL(scale_by_clip_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, sumsq, max_norm);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                  float beta2, float eps, int step, const double* clip_sumsq, float max_norm,
                                  ddmp_stream stream) {
    ARG_TRY(p && g && m && v && n > 0 && step >= 1 && lr >= 0.f);
    ARG_TRY(!clip_sumsq || max_norm > 0.f);
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)std::sqrt(bc2);
    const int grid = (int)std::min<int64_t>(cdiv(n, 256), 2048);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2, eps,
                       step_size, bc2_sqrt, clip_sumsq, max_norm, (const float*)nullptr);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_adam_prepare(int32_t* step_counter, float lr, float beta1, float beta2, float* coef,
                                 ddmp_stream stream) {
    ARG_TRY(step_counter && coef && lr >= 0.f);
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_counter, lr, beta1, beta2, coef);
    LAUNCH_TRY();
    return DDMP_OK;
}

extern "C" int ddmp_adam_step_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, float beta1,
                                      float beta2, float eps, const float* coef, const double* clip_sumsq,
                                      float max_norm, ddmp_stream stream) {
    ARG_TRY(p && g && m && v && n > 0 && coef);
    ARG_TRY(!clip_sumsq || max_norm > 0.f);
    const int grid = (int)std::min<int64_t>(cdiv(n, 256), 2048);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2, eps,
                       0.f, 1.f, clip_sumsq, max_norm, coef);
    LAUNCH_TRY();
    return DDMP_OK;
}
