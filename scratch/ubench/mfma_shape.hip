// Bare MFMA loops on random data, A operand re-read from LDS each step, two waves per SIMD:
// v_mfma_f32_32x32x16_bf16 (1 per 1 KiB A block) vs v_mfma_f32_16x16x32_bf16 (2 per 1 KiB A block, two
// column tiles).  Prints TFLOP/s of each.  hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool USE_LDS>
__global__ __launch_bounds__(512, 2) void k(const uint4* __restrict__ w, const uint4* __restrict__ x, float* out, int iters) {
    __shared__ uint4 lds[64 * 64];       // 64 blocks of 1 KiB
    for (int i = threadIdx.x; i < 64 * 64; i += 512) lds[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 b[16];
    for (int i = 0; i < 16; ++i) b[i] = __builtin_bit_cast(bf16x8, x[(blockIdx.x * 16 + i) * 64 + lane]);
    bf16x8 areg[8];
    for (int i = 0; i < 8; ++i) areg[i] = __builtin_bit_cast(bf16x8, lds[i * 64 + lane]);
    f32x16 acc = {0}, acc2 = {0};
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 64; ++blk) {
            const bf16x8 a = USE_LDS ? __builtin_bit_cast(bf16x8, lds[blk * 64 + lane]) : areg[blk & 7];
            if (SHAPE == 32) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[blk & 15], acc, 0, 0, 0);
            } else if (SHAPE == 33) {      // two independent 32x32 accumulators, alternating
                if (blk & 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[blk & 15], acc, 0, 0, 0);
                else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[blk & 15], acc2, 0, 0, 0);
            } else {
                if (blk & 1) {
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[blk & 15], c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[(blk + 8) & 15], c3, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[blk & 15], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[(blk + 8) & 15], c1, 0, 0, 0);
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const int nblk = 256;
    std::vector<unsigned short> hw(64 * 64 * 8), hx((size_t)nblk * 16 * 64 * 8);
    srand(1);
    auto rb = []() { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.25f; unsigned u; std::memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (auto& v : hw) v = rb();
    for (auto& v : hx) v = rb();
    uint4 *w, *x; float* out;
    hipMalloc(&w, hw.size() * 2); hipMalloc(&x, hx.size() * 2); hipMalloc(&out, nblk * 512 * 4);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        for (int shape : {32, 33, 16}) {
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) {
                if (shape == 32) hipLaunchKernelGGL((k<32, false>), dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
                else if (shape == 33) hipLaunchKernelGGL((k<33, false>), dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
                else hipLaunchKernelGGL((k<16, false>), dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per A block: 32x32x16 -> 32768 flop; 2 x 16x16x32 -> 32768 flop
            double flop = 5.0 * nblk * 8 * (double)iters * 64 * 32768.0;
            printf("shape %2d: %.2f ms  %.1f TFLOP/s\n", shape, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
