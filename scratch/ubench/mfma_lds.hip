// MFMA loops on random data with the A operand read from LDS through an explicit software pipeline
// (inline-asm ds_read_b128, counted lgkmcnt waits, PF blocks ahead), two waves per SIMD.
//   mode 0: 32x32x16, one A block per MFMA, two alternating accumulators  (the chain kernels' structure)
//   mode 1: 32x32x16, one A block feeds two MFMAs (two sample tiles per wave: half the LDS bytes per FLOP)
//   mode 2: 16x16x32, one A block feeds two MFMAs (two 16-sample column tiles)
//   mode 3: 16x16x32, one A block feeds four MFMAs (four column tiles = 64 samples per wave)
//   modes 4-7: as 0-3, but the reads run PF blocks ahead across the 32-block boundary as well (the kernels stop
//              their read-ahead at every chunk entry and layer start: the data is only known to have landed there)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int PF = 4;

template <int OFF> __device__ __forceinline__ i32x4 rd(uint32_t addr) {
    i32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); return v;
}
template <int N> __device__ __forceinline__ void wt(i32x4& v) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N)); }

template <int MODE_>
__global__ __launch_bounds__(512, 2) void k(const uint4* __restrict__ w, const uint4* __restrict__ x, float* out, int iters) {
    __shared__ uint4 lds[32 * 64];       // 32 blocks of 1 KiB
    for (int i = threadIdx.x; i < 32 * 64; i += 512) lds[i] = w[i];
    __syncthreads();
    constexpr int MODE = MODE_ & 3;
    constexpr bool WRAP = MODE_ >= 4;
    const int lane = threadIdx.x & 63;
    const uint32_t base = (uint32_t)(uintptr_t)lds + lane * 16;
    bf16x8 b[16];
    for (int i = 0; i < 16; ++i) b[i] = __builtin_bit_cast(bf16x8, x[(blockIdx.x * 16 + i) * 64 + lane]);
    f32x16 A0 = {0}, A1 = {0};
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    i32x4 a[32];
    if (WRAP) { a[0] = rd<0>(base); a[1] = rd<1024>(base); a[2] = rd<2048>(base); a[3] = rd<3072>(base); }
    for (int it = 0; it < iters; ++it) {
        if (!WRAP) { a[0] = rd<0>(base); a[1] = rd<1024>(base); a[2] = rd<2048>(base); a[3] = rd<3072>(base); }
#define STEP(I)                                                                                         \
        if (WRAP || (I) + PF < 32) a[((I) + PF) & 31] = rd<(((I) + PF) & 31) * 1024>(base);             \
        if (WRAP || (I) + PF < 32) wt<PF>(a[I]); else wt<31 - (I)>(a[I]);                               \
        {                                                                                               \
            const bf16x8 av = __builtin_bit_cast(bf16x8, a[I]);                                         \
            if (MODE == 0) {                                                                            \
                if ((I) & 1) A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[(I) & 15], A1, 0, 0, 0); \
                else A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[(I) & 15], A0, 0, 0, 0);        \
            } else if (MODE == 1) {                                                                     \
                A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[(I) & 7], A0, 0, 0, 0);              \
                A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[8 + ((I) & 7)], A1, 0, 0, 0);        \
            } else if (MODE == 2) {                                                                     \
                if ((I) & 1) {                                                                          \
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(I) & 7], c2, 0, 0, 0);          \
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[8 + ((I) & 7)], c3, 0, 0, 0);    \
                } else {                                                                                \
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(I) & 7], c0, 0, 0, 0);          \
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[8 + ((I) & 7)], c1, 0, 0, 0);    \
                }                                                                                       \
            } else {                                                                                    \
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[(I) & 3], c0, 0, 0, 0);              \
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[4 + ((I) & 3)], c1, 0, 0, 0);        \
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[8 + ((I) & 3)], c2, 0, 0, 0);        \
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[12 + ((I) & 3)], c3, 0, 0, 0);       \
            }                                                                                           \
        }
#define S4(I) STEP(I) STEP(I + 1) STEP(I + 2) STEP(I + 3)
        S4(0) S4(4) S4(8) S4(12) S4(16) S4(20) S4(24) S4(28)
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += A0[i] + A1[i];
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const int nblk = 256;
    std::vector<unsigned short> hw(32 * 64 * 8), hx((size_t)nblk * 16 * 64 * 8);
    srand(1);
    auto rb = []() { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.25f; unsigned u; std::memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (auto& v : hw) v = rb();
    for (auto& v : hx) v = rb();
    uint4 *w, *x; float* out;
    hipMalloc(&w, hw.size() * 2); hipMalloc(&x, hx.size() * 2); hipMalloc(&out, nblk * 512 * 4);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double per_block[8] = {32768.0, 65536.0, 32768.0, 65536.0, 32768.0, 65536.0, 32768.0, 65536.0};   // flop per A block
    for (int rep = 0; rep < 2; ++rep) {
        for (int mode = 0; mode < 8; ++mode) {
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
                else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
                else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
                else if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
                else if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
                else hipLaunchKernelGGL(k<7>, dim3(nblk), dim3(512), 0, 0, w, x, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = 5.0 * nblk * 8 * (double)iters * 32 * per_block[mode];
            printf("mode %d: %.2f ms  %.1f TFLOP/s\n", mode, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
