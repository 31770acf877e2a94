// Adam over one flat fp32 parameter buffer: the optimizer.step() of
// scripts/train_nerf.py:336 (torch.optim.Adam defaults: no weight decay, no amsgrad) as a
// single bandwidth-bound launch instead of one small launch per parameter tensor.
#include "common.hpp"

namespace nerf {

__global__ __launch_bounds__(256) void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float step_size,
                                                   float beta1, float beta2, float eps, float inv_sqrt_bc2,
                                                   float grad_scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);          // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;      // mul_(beta2).addcmul_(g, g, 1 - beta2)
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// Device-resident hyper-parameters, for callers that replay the step from a HIP graph (scalars passed
// by value would be frozen into the graph).  state = {lr, beta1, beta2, eps, grad_scale, step, -, -};
// the prepare kernel advances `step` and derives the bias-corrected factors into state[6..7].
__global__ void adam_prepare_kernel(float* __restrict__ state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double step = (double)state[5] + 1.0;
        state[5] = (float)step;
        state[6] = (float)((double)state[0] / (1.0 - pow((double)state[1], step)));
        state[7] = (float)(1.0 / sqrt(1.0 - pow((double)state[2], step)));
    }
}

__global__ __launch_bounds__(256) void adam_dev_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       const float* __restrict__ state) {
    const float beta1 = state[1], beta2 = state[2], eps = state[3], grad_scale = state[4];
    const float step_size = state[6], inv_sqrt_bc2 = state[7];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);
        const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

}  // namespace nerf

extern "C" int nerf_adam_step_dev(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                  float* state, void* stream) {
    NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq && state, NERF_E_NULL);
    NERF_REQUIRE(n >= 0, NERF_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nerf::adam_prepare_kernel, dim3(1), dim3(64), 0, st, state);
    int rc = nerf_launch_status();
    if (rc || n == 0) return rc;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(nerf::adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, params, grads, exp_avg,
                       exp_avg_sq, state);
    return nerf_launch_status();
}

extern "C" int nerf_adam_step(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                              float lr, float beta1, float beta2, float eps, int64_t step, float grad_scale,
                              void* stream) {
    NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq, NERF_E_NULL);
    NERF_REQUIRE(n >= 0 && step >= 1, NERF_E_SHAPE);
    if (n == 0) return NERF_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(nerf::adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, params,
                       grads, exp_avg, exp_avg_sq, step_size, beta1, beta2, eps, inv_sqrt_bc2, grad_scale);
    return nerf_launch_status();
}
