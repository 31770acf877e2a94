// Shared host/device helpers for libnerf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nerf_hip.h"

#define NERF_REQUIRE(cond, code) \
    do {                         \
        if (!(cond)) return (code); \
    } while (0)

// Launch epilogue: report the launch error of the call just made (sticky errors
// from earlier asynchronous work are not consumed here).
static inline int nerf_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NERF_OK : (int)e;
}

static inline bool nerf_aligned(const void* p, uintptr_t a) { return ((uintptr_t)p & (a - 1)) == 0; }

namespace nerf {

constexpr int WAVE = 64;

// torch.linspace(0, 1, n)[k] on the CPU: step*k in the first half and a single-rounding
// 1 - step*(n-1-k) in the second half (checked bit-for-bit against ATen for n = 5..192).
__device__ __forceinline__ float linspace01(int k, int n) {
    const float step = n > 1 ? 1.0f / (float)(n - 1) : 0.0f;
    return (k < n / 2) ? step * (float)k : fmaf(-step, (float)(n - 1 - k), 1.0f);
}

// Orders LDS traffic of ONE wave: the LDS unit serves a wave's accesses in issue order, so
// only the compiler has to be kept from moving reads above earlier writes.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- wave-level primitives (64 lanes) ------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive prefix product over the 64 lanes (Hillis-Steele on shuffles)
__device__ __forceinline__ double wave_incl_prod(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}

__device__ __forceinline__ double wave_incl_sum(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// inclusive SUFFIX sum: lane l gets sum_{j >= l} v_j
__device__ __forceinline__ float wave_incl_suffix_sum(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_down(v, o, 64);
        if (lane + o < 64) v += t;
    }
    return v;
}

}  // namespace nerf
