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
// by value would be frozen into the graph).  state = {lr, beta1, beta2, eps, grad_scale, step, ticket, -}.
// One launch: every workgroup derives the bias-corrected factors of step + 1 itself (thread 0, fp64, shared
// through LDS); the workgroup that finishes last -- told by a ticket counter in state[6], which it resets --
// advances `step`.  Every workgroup has read `step` before it takes its ticket, so none sees the new value.
// The ticket must be 0 when the first call is made (the caller zero-initialises `state`); every call leaves it 0.
__global__ __launch_bounds__(256) void adam_dev_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       float* __restrict__ state) {
    __shared__ float factors[2];
    const float beta1 = state[1], beta2 = state[2], eps = state[3], grad_scale = state[4];
    if (threadIdx.x == 0) {
        // beta^step by repeated squaring (the step count is an integer; libm's pow costs this one thread tens of
        // microseconds while the other 255 wait)
        unsigned k = (unsigned)(state[5] + 1.0f);
        double p1 = 1.0, p2 = 1.0, b1 = (double)beta1, b2 = (double)beta2;
        for (; k != 0; k >>= 1) {
            if (k & 1) { p1 *= b1; p2 *= b2; }
            b1 *= b1; b2 *= b2;
        }
        factors[0] = (float)((double)state[0] / (1.0 - p1));
        factors[1] = (float)(1.0 / sqrt(1.0 - p2));
    }
    __syncthreads();
    const float step_size = factors[0], inv_sqrt_bc2 = factors[1];
    auto update = [&](float& pi, float gi, float& mi, float& vi) {
        gi *= grad_scale;
        mi = mi + (gi - mi) * (1.0f - beta1);
        vi = vi * beta2 + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        pi = pi - step_size * (mi / denom);
    };
    // four elements per thread and iteration where the buffers allow 16-byte accesses
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    const int64_t n4 = vec ? n / 4 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 p4 = reinterpret_cast<float4*>(p)[i], m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
        const float4 g4 = reinterpret_cast<const float4*>(g)[i];
        update(p4.x, g4.x, m4.x, v4.x); update(p4.y, g4.y, m4.y, v4.y);
        update(p4.z, g4.z, m4.z, v4.z); update(p4.w, g4.w, m4.w, v4.w);
        reinterpret_cast<float4*>(p)[i] = p4; reinterpret_cast<float4*>(m)[i] = m4; reinterpret_cast<float4*>(v)[i] = v4;
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        update(p[i], g[i], m[i], v[i]);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* ticket = reinterpret_cast<unsigned*>(state + 6);
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {      // the last workgroup
            *ticket = 0u;
            state[5] = state[5] + 1.0f;
        }
    }
}

}  // namespace nerf

extern "C" int nerf_adam_step_dev(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                  float* state, void* stream) {
    NERF_REQUIRE(params && grads && exp_avg && exp_avg_sq && state, NERF_E_NULL);
    NERF_REQUIRE(n >= 0, NERF_E_SHAPE);
    hipStream_t st = (hipStream_t)stream;
    // few workgroups: each ends with one atomic on the same word, which serialise at about 90 per microsecond
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;            // n = 0 still counts as a step
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
