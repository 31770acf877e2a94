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

}  // namespace nerf

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
