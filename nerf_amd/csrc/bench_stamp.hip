// Time-stamp kernel for bench.py (not part of the C ABI of libnerf_hip.so; built into libnerf_bench.so).
// One lane writes the constant-rate 100 MHz counter (s_memrealtime) into a slot.  Enqueued in front of and
// behind a call while a training step is captured into a HIP graph, the difference of the two slots after a
// replay is that call's duration inside the replayed step, kernels back to back, plus the two launch
// boundaries around the stamps (about a microsecond each).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void stamp_kernel(unsigned long long* slot) {
    if (threadIdx.x == 0) *slot = wall_clock64();
}

extern "C" int nerf_bench_stamp(unsigned long long* slot, void* stream) {
    if (!slot) return -1;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// ticks of that counter per second
extern "C" int64_t nerf_bench_stamp_rate(void) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess || khz <= 0) return 100000000;
    return (int64_t)khz * 1000;
}
