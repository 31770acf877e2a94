// Shared host/device helpers for libnerf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nerf_hip.h"

#define NERF_REQUIRE(cond, code) \
    do {                         \
        if (!(cond)) return (code); \
    } while (0)

// Launch epilogue: hipGetLastError() returns the last error recorded on this host thread and clears it.
// Right after a launch that is the launch's own error (bad configuration, missing code object); an error
// recorded by an earlier call that nobody collected is reported, and consumed, here as well.
static inline int nerf_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NERF_OK : (int)e;
}

// `precision` of the MLP entry points: the operand type, optionally with the architecture flag
static inline bool nerf_prec_ok(int precision) {
    const int p = precision & ~NERF_ARCH_NOVIEW;
    return p == NERF_PREC_BF16 || p == NERF_PREC_FP16;
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

// Cross-lane moves on the VALU's DPP path (no LDS traffic).  CTRL is the dpp_ctrl code: 0x110+n = row_shr:n,
// 0x100+n = row_shl:n (rows are 16 lanes), 0x130 / 0x138 = wave_shl:1 / wave_shr:1, 0x142 / 0x143 =
// row_bcast:15 / row_bcast:31 (last lane of a row, or of the lower half, to every lane above it).  A lane
// whose source lies outside its row (or the wave), or whose row ROWS masks out, receives `old`.
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ float dpp_move(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, ROWS, 0xf, false));
}

template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ double dpp_move(double old, double v) {
    const long long o = __builtin_bit_cast(long long, old), x = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)o, (int)x, CTRL, ROWS, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(o >> 32), (int)(x >> 32), CTRL, ROWS, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ float read_lane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__device__ __forceinline__ double read_lane(double v, int l) {
    const long long x = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)x, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(x >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Inclusive prefix scans over the 64 lanes: Kogge-Stone inside each 16-lane row, then the row totals
// carried upwards with the two row broadcasts.
__device__ __forceinline__ float wave_incl_sum(float v) {
    v += dpp_move<0x111>(0.0f, v); v += dpp_move<0x112>(0.0f, v);
    v += dpp_move<0x114>(0.0f, v); v += dpp_move<0x118>(0.0f, v);
    v += dpp_move<0x142, 0xa>(0.0f, v);
    v += dpp_move<0x143, 0xc>(0.0f, v);
    return v;
}

__device__ __forceinline__ double wave_incl_sum(double v) {
    v += dpp_move<0x111>(0.0, v); v += dpp_move<0x112>(0.0, v);
    v += dpp_move<0x114>(0.0, v); v += dpp_move<0x118>(0.0, v);
    v += dpp_move<0x142, 0xa>(0.0, v);
    v += dpp_move<0x143, 0xc>(0.0, v);
    return v;
}

__device__ __forceinline__ double wave_incl_prod(double v) {
    v *= dpp_move<0x111>(1.0, v); v *= dpp_move<0x112>(1.0, v);
    v *= dpp_move<0x114>(1.0, v); v *= dpp_move<0x118>(1.0, v);
    v *= dpp_move<0x142, 0xa>(1.0, v);
    v *= dpp_move<0x143, 0xc>(1.0, v);
    return v;
}

// sum over the wave, the same value on every lane
__device__ __forceinline__ float wave_sum(float v) { return read_lane(wave_incl_sum(v), 63); }
__device__ __forceinline__ double wave_sum(double v) { return read_lane(wave_incl_sum(v), 63); }

// inclusive SUFFIX sum: lane l gets sum_{j >= l} v_j (suffix scan inside each row, then the totals of the
// rows above, read from their first lanes)
__device__ __forceinline__ float wave_incl_suffix_sum(float v, int lane) {
    v += dpp_move<0x101>(0.0f, v); v += dpp_move<0x102>(0.0f, v);
    v += dpp_move<0x104>(0.0f, v); v += dpp_move<0x108>(0.0f, v);
    const float t1 = read_lane(v, 16), t2 = read_lane(v, 32), t3 = read_lane(v, 48);
    const int row = lane >> 4;
    const float above = row == 0 ? t1 + (t2 + t3) : (row == 1 ? t2 + t3 : (row == 2 ? t3 : 0.0f));
    return v + above;
}

}  // namespace nerf
