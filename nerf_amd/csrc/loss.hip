// The training loss of scripts/train_nerf.py:299 and :330-331 in one launch:
//   loss_a = mean((a - t)^2), loss_b = mean((b - t)^2), total = loss_a + loss_b,
// together with the gradients of the two means, 2 (x - t) / n_elements.  The reference spells this as
// seven small ATen kernels forward and about ten backward; for a 1024-ray batch the data is 36 KiB.
// One workgroup, fp64 partial sums, fixed reduction order: the result does not depend on timing.
#include "common.hpp"

namespace nerf {

__global__ __launch_bounds__(1024) void mse_pair_kernel(int64_t n_el, const float* __restrict__ a,
                                                        const float* __restrict__ b, const float* __restrict__ t,
                                                        float* __restrict__ losses, float* __restrict__ da,
                                                        float* __restrict__ db) {
    __shared__ double red[2][16];
    const float k = 2.0f / (float)n_el;
    double sa = 0.0, sb = 0.0;
    for (int64_t i = threadIdx.x; i < n_el; i += blockDim.x) {
        const float ti = t[i];
        const float ea = a[i] - ti;
        sa += (double)(ea * ea);
        if (da) da[i] = k * ea;
        if (b) {
            const float eb = b[i] - ti;
            sb += (double)(eb * eb);
            if (db) db[i] = k * eb;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sa += __shfl_down(sa, off, 64);
        sb += __shfl_down(sb, off, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = sa; red[1][wave] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 0; w < nw; ++w) { ta += red[0][w]; tb += red[1][w]; }
        const float la = (float)(ta / (double)n_el), lb = (float)(tb / (double)n_el);
        losses[0] = la + lb; losses[1] = la; losses[2] = lb;
    }
}

}  // namespace nerf

extern "C" int nerf_mse_pair(int64_t n_elements, const float* a, const float* b, const float* target, float* losses,
                             float* d_a, float* d_b, void* stream) {
    NERF_REQUIRE(a && target && losses, NERF_E_NULL);
    NERF_REQUIRE(n_elements >= 1, NERF_E_SHAPE);
    NERF_REQUIRE(b || !d_b, NERF_E_NULL);
    hipLaunchKernelGGL(nerf::mse_pair_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_elements, a, b, target,
                       losses, d_a, d_b);
    return nerf_launch_status();
}
