// Write-rate ceiling of the training forward's store pattern: 256 workgroups x 8 waves, each wave writing 1 KiB
// units (16 bytes per lane, non-temporal), tensor-major like the saved activations (a wave's 16 units of a layer
// are contiguous, layers are n_wt * 16 KiB apart).
//   mode 0: stores back to back              mode 1: two stores, then about 1000 idle cycles (a tile of MFMAs)
//   mode 2: as 1 with plain (cached) stores  mode 3: as 1, but all of a pass's units contiguous per wave (144 KiB)
//   mode 4: one store per ~500 idle cycles   mode 5: a layer's 16 stores in one burst, then ~8000 idle cycles
//   mode 6: as 1, waves start staggered by wave * 128 cycles
//   modes 7..12: as 1 with the store's cache-policy bits spelled out: sc0 / sc1 / sc0 sc1 / nt sc0 / nt sc1 / nt sc0 sc1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int LAYERS = 9, UNITS = 16;

template <int POL>
__device__ __forceinline__ void store_pol(char* a, const i32x4& v) {
    if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(a), "v"(v) : "memory");
    if (POL == 8) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(a), "v"(v) : "memory");
    if (POL == 9) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(a), "v"(v) : "memory");
    if (POL == 10) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(a), "v"(v) : "memory");
    if (POL == 11) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(a), "v"(v) : "memory");
    if (POL == 12) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(a), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(char* out, int passes, long n_wt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32x4 v = {lane, wave, (int)blockIdx.x, 7};
    if (MODE == 6) for (int q = 0; q < wave; ++q) __builtin_amdgcn_s_sleep(2);
    for (int p = 0; p < passes; ++p) {
        const long wt = ((long)p * gridDim.x + blockIdx.x) * 8 + wave;
        for (int l = 0; l < LAYERS; ++l) {
            if (MODE == 4 || MODE == 5) {
                for (int u = 0; u < UNITS; ++u) {
                    char* a = out + (((long)l * n_wt + wt) * UNITS + u) * 1024 + lane * 16;
                    __builtin_nontemporal_store(v, (i32x4*)a);
                    if (MODE == 4) __builtin_amdgcn_s_sleep(8);
                    v[3] += 1;
                }
                if (MODE == 5) for (int q = 0; q < 16; ++q) __builtin_amdgcn_s_sleep(8);
                continue;
            }
            for (int u = 0; u < UNITS; u += 2) {
                char* a = MODE == 3 ? out + ((wt * LAYERS + l) * UNITS + u) * 1024 + lane * 16
                                    : out + (((long)l * n_wt + wt) * UNITS + u) * 1024 + lane * 16;
                if (MODE >= 7) { store_pol<MODE>(a, v); store_pol<MODE>(a + 1024, v); }
                else if (MODE == 2) { *(i32x4*)a = v; *(i32x4*)(a + 1024) = v; }
                else { __builtin_nontemporal_store(v, (i32x4*)a); __builtin_nontemporal_store(v, (i32x4*)(a + 1024)); }
                if (MODE != 0) { __builtin_amdgcn_s_sleep(8); __builtin_amdgcn_s_sleep(8); }
                v[3] += 1;
            }
        }
    }
}

int main() {
    const int passes = 3, grid = 256;
    const long n_wt = (long)passes * grid * 8;
    const size_t bytes = (size_t)n_wt * LAYERS * UNITS * 1024;
    char* buf; hipMalloc(&buf, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 13; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 9) hipLaunchKernelGGL(k<9>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 10) hipLaunchKernelGGL(k<10>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 11) hipLaunchKernelGGL(k<11>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
                if (mode == 12) hipLaunchKernelGGL(k<12>, dim3(grid), dim3(512), 0, 0, buf, passes, n_wt);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.1f us per launch, %.2f GB -> %.0f GB/s\n", mode, ms * 100, bytes / 1e9, bytes / (ms / 10 * 1e-3) / 1e9);
        }
    }
    return 0;
}
