// How fast can ONE compute unit stream from HBM?  Each workgroup (8 waves) walks its own slice of a large buffer
// with non-temporal accesses, DEPTH KiB-sized requests per wave kept in flight:
//   mode 0: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave request), counted vmcnt
//   mode 1: register loads (global_load_dwordx4, 1 KiB per wave request), counted vmcnt
//   stage protocol (kernel `staged`): the weight-gradient kernel's copy schedule without its LDS reads and MFMAs -- a ring
//   of four 32 KiB stages, per step every wave waits for its own four copies of the oldest stage, all meet at a barrier,
//   then each issues its four copies of the stage three steps ahead; STREAMS = 1 reads one contiguous run per workgroup,
//   STREAMS = 2 takes the first 16 KiB of a stage from one run and the second from another (two operands)
// run with 32 and with 256 workgroups (one per compute unit): the per-unit rate times 256 against the chip's rate
// says whether a kernel on part of the chip is bound by what one unit can keep in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) char gbl_char;
typedef __attribute__((address_space(3))) char lds_char;
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 2) void k(const char* __restrict__ src, size_t bytes_per_wg, int* sink) {
    __shared__ __attribute__((aligned(16))) char lds[8 * DEPTH * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* p = src + (size_t)blockIdx.x * bytes_per_wg + (size_t)wave * 1024 + lane * 16;
    const size_t n = bytes_per_wg / (8 * 1024);          // requests per wave
    i32x4 acc = {0, 0, 0, 0};
    if (MODE == 0) {
        for (size_t i = 0; i < n; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_char*)(p + i * 8192), (lds_char*)(lds + (wave * DEPTH + (int)(i % DEPTH)) * 1024), 16, 0, 2);
            if (i >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc[0] = *(int*)(lds + threadIdx.x * 4);
    } else {
        i32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load((const i32x4*)(p + (size_t)d * 8192));
        for (size_t i = DEPTH; i + DEPTH <= n; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                acc += v[d];
                v[d] = __builtin_nontemporal_load((const i32x4*)(p + (i + d) * 8192));
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    if (acc[0] == 0x12345678) sink[0] = acc[1] + acc[2] + acc[3];
}

template <int STREAMS, int STAGES>
__global__ __launch_bounds__(512, 2) void staged(const char* __restrict__ src, size_t bytes_per_wg, int* sink) {
    __shared__ __attribute__((aligned(16))) char lds[STAGES * 32 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * bytes_per_wg;
    const int nsteps = (int)(bytes_per_wg / (32 * 1024));
    auto issue = [&](int step, int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = wave + 8 * i;                       // unit 0..31 of the stage
            const char* p;
            if (STREAMS == 1) p = base + (size_t)step * 32768 + u * 1024;
            else p = base + (u < 16 ? 0 : bytes_per_wg / 2) + (size_t)step * 16384 + (u & 15) * 1024;
            __builtin_amdgcn_global_load_lds((gbl_char*)(p + lane * 16), (lds_char*)(lds + stage * 32768 + u * 1024), 16, 0, 2);
        }
    };
    for (int s = 0; s < STAGES - 1 && s < nsteps; ++s) issue(s, s);
    int stage = 0;
    for (int step = 0; step < nsteps; ++step) {
        const int younger = nsteps - 1 - step < STAGES - 2 ? nsteps - 1 - step : STAGES - 2;
        if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (step + STAGES - 1 < nsteps) issue(step + STAGES - 1, (stage + STAGES - 1) % STAGES);
        stage = (stage + 1) % STAGES;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (*(int*)(lds + threadIdx.x * 4) == 0x12345678) sink[0] = 1;
}

template <int STREAMS, int STAGES>
static void run_staged(int grid, const char* buf, size_t per_wg, int* sink, hipEvent_t e0, hipEvent_t e1) {
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((staged<STREAMS, STAGES>), dim3(grid), dim3(512), 0, 0, buf, per_wg, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("grid %3d  stage protocol, %d stream(s), %d stages: %7.1f us  %6.0f GB/s  = %5.1f GB/s per workgroup\n", grid, STREAMS,
                        STAGES, ms * 1e3, grid * per_wg / (ms * 1e-3) / 1e9, per_wg / (ms * 1e-3) / 1e9);
    }
}

int main() {
    const size_t total = (size_t)4 << 30;
    char* buf; int* sink;
    (void)hipMalloc(&buf, total); (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 1, total);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {32, 64, 128, 256}) {
        const size_t per_wg = (total / 256) & ~(size_t)(64 * 1024 - 1);      // 16 MiB per workgroup at every grid size
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL((k<0, 4>), dim3(grid), dim3(512), 0, 0, buf, per_wg, sink);
                if (mode == 1) hipLaunchKernelGGL((k<0, 12>), dim3(grid), dim3(512), 0, 0, buf, per_wg, sink);
                if (mode == 2) hipLaunchKernelGGL((k<1, 4>), dim3(grid), dim3(512), 0, 0, buf, per_wg, sink);
                if (mode == 3) hipLaunchKernelGGL((k<1, 12>), dim3(grid), dim3(512), 0, 0, buf, per_wg, sink);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("grid %3d  %s depth %2d KiB/wave: %7.1f us  %6.0f GB/s  = %5.1f GB/s per workgroup\n", grid,
                                mode < 2 ? "LDS-DMA " : "register", (mode & 1) ? 12 : 4, ms * 1e3, grid * per_wg / (ms * 1e-3) / 1e9,
                                per_wg / (ms * 1e-3) / 1e9);
            }
        }
        run_staged<1, 4>(grid, buf, per_wg, sink, e0, e1);
        run_staged<2, 4>(grid, buf, per_wg, sink, e0, e1);
        run_staged<1, 5>(grid, buf, per_wg, sink, e0, e1);
        run_staged<2, 5>(grid, buf, per_wg, sink, e0, e1);
    }
    return 0;
}
