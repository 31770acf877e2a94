// How fast can one workgroup hand 16-bit "units" (1 KiB each) to another workgroup of the same launch through the
// cache hierarchy instead of through HBM?  The question behind it: could the backward-data chain of the training
// step hand its dZ units to weight-gradient workgroups inside one persistent launch, so that dZ is neither
// written to nor read from HBM?
//
// 256 workgroups of 512 threads, one per compute unit (100 KiB of LDS each).  Workgroups b and b + 8 land on the
// same XCD under the round-robin placement (checked with HW_REG_XCC_ID and reported); they form a pair:
// producer = the one with (b / 8) even, consumer = b + 8 (PLACE = 0), or consumer = b + 12 on another XCD
// (PLACE = 1).  Every pair owns a ring of SLOTS slots of SLOT_KIB KiB in a buffer that stays cache-resident
// (256 pairs x ring <= 4 MiB per XCD).  Per slot the producer's eight waves store SLOT_KIB units (16 bytes per
// lane), publish a sequence number, and wait for the consumer to have freed the slot SLOTS steps back; the
// consumer polls the number, copies the slot into LDS and reads it back with ds_read_b64_tr_b16 like the
// weight-gradient kernel does, then frees the slot.
//   MODE 0: write-through (sc1) payload stores, each wave drains them, barrier, one lane stores the flag sc1;
//           the consumer polls with an sc1 load, barrier, then sc1 register loads (16 bytes per lane) and
//           ds_write_b128 into LDS.
//   MODE 1: plain payload stores, drain, barrier, one lane: agent-scope release fence, drain, relaxed agent flag
//           store; the consumer: relaxed poll, agent-scope acquire fence, drain, barrier, LDS-DMA
//           (global_load_lds_dwordx4) into LDS.
// Every spin is bounded (about a second) and raises an abort word that all loops test, so the launch always ends.
// Output: GB/s per pair and over the chip, against 28 GB/s per unit (the HBM read share with all units streaming)
// and 56 GB/s (one unit streaming alone); stale or torn units counted (every dword carries its slot's number).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(1))) char gbl_char;
typedef __attribute__((address_space(3))) char lds_char;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Args {
    char* ring;            // pairs x SLOTS x SLOT_KIB KiB
    unsigned* full;        // per pair, 64-byte stride: number of slots published
    unsigned* freed;       // per pair, 64-byte stride: number of slots consumed
    unsigned* abort_word;
    unsigned* xcc;         // per workgroup: XCC id
    unsigned long long* bad;   // dwords that did not carry their slot's number
    unsigned long long* ticks; // per workgroup: s_memrealtime ticks of the loop
    int n_slots_total;     // slots each pair moves
    int place;
};

__device__ __forceinline__ unsigned load_sc1(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_sc1(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait until *p >= want (one lane polls; bounded); returns false when the launch is being aborted
__device__ __forceinline__ bool wait_for(const unsigned* p, unsigned want, unsigned* abort_word) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if ((int)(load_sc1(p) - want) >= 0) return true;
        if ((spin & 255) == 255 && load_sc1(abort_word) != 0) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    store_sc1(abort_word, 1u);
    return false;
}

template <int MODE, int SLOT_KIB, int SLOTS>
__global__ __launch_bounds__(512) void handoff(const Args a) {
    __shared__ __attribute__((aligned(16))) char lds[100 * 1024];
    __shared__ int ok_sh;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int group = b / 16, x = b % 8;
    const bool producer = ((b / 8) & 1) == 0;
    const int pair = group * 8 + x;
    // the consumer of pair (group, x): workgroup 16 group + 8 + x, or + ((x + 4) % 8) on another XCD
    const int my_pair = producer ? pair : (a.place == 0 ? pair : group * 8 + ((x + 4) % 8));
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc[b] = id & 0xf;
    }
    char* ring = a.ring + (size_t)my_pair * SLOTS * SLOT_KIB * 1024;
    unsigned* full = a.full + my_pair * 16;
    unsigned* freed = a.freed + my_pair * 16;
    constexpr int UNITS_PER_WAVE = SLOT_KIB / 8;
    static_assert(SLOT_KIB % 8 == 0 && SLOT_KIB <= 96, "slot");
    unsigned long long bad = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < a.n_slots_total; ++it) {
        char* slot = ring + (size_t)(it % SLOTS) * SLOT_KIB * 1024;
        if (producer) {
            if (it >= SLOTS) {          // the slot must have been consumed SLOTS steps ago
                if (threadIdx.x == 0) ok_sh = wait_for(freed, (unsigned)(it - SLOTS + 1), a.abort_word);
                __syncthreads();
                if (!ok_sh) break;
            }
            const i32x4 v = {it, it, it, it};
#pragma unroll
            for (int u = 0; u < UNITS_PER_WAVE; ++u) {
                i32x4* p = reinterpret_cast<i32x4*>(slot + (wave * UNITS_PER_WAVE + u) * 1024 + lane * 16);
                if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
                else *p = v;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                if (MODE == 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                store_sc1(full, (unsigned)(it + 1));
            }
        } else {
            if (threadIdx.x == 0) {
                ok_sh = wait_for(full, (unsigned)(it + 1), a.abort_word);
                if (MODE == 1) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
            if (!ok_sh) break;
            i32x4 v[UNITS_PER_WAVE];
#pragma unroll
            for (int u = 0; u < UNITS_PER_WAVE; ++u) {       // all of the wave's requests in flight at once
                const int unit = wave * UNITS_PER_WAVE + u;
                const char* g = slot + unit * 1024 + lane * 16;
                if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(g) : "memory");
                else __builtin_amdgcn_global_load_lds((gbl_char*)g, (lds_char*)(lds + unit * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 0) {
#pragma unroll
                for (int u = 0; u < UNITS_PER_WAVE; ++u) {
                    asm volatile("" : "+v"(v[u]));
                    *reinterpret_cast<i32x4*>(lds + (wave * UNITS_PER_WAVE + u) * 1024 + lane * 16) = v[u];
                }
            }
            __syncthreads();
            // read the slot back as the weight-gradient kernel would (transposing 8-byte reads), and check it
#pragma unroll
            for (int u = 0; u < UNITS_PER_WAVE; ++u) {
                const unsigned addr = (unsigned)(size_t)(lds_char*)(lds + (wave * UNITS_PER_WAVE + u) * 1024) + lane * 8;
                s16x4 lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(hi) : "v"(addr));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi));
                const short w0 = (short)(it & 0xffff), w1 = (short)((unsigned)it >> 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bad += (lo[q] != w0 && lo[q] != w1);
                    bad += (hi[q] != w0 && hi[q] != w1);
                }
            }
            __syncthreads();           // everyone is done with the slot and with LDS
            if (threadIdx.x == 0) store_sc1(freed, (unsigned)(it + 1));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) a.ticks[b] = t1 - t0;
    if (bad) atomicAdd(a.bad, bad);
}

template <int MODE, int SLOT_KIB, int SLOTS>
static void run(int place, int n_slots) {
    const int pairs = 128;
    Args a{};
    hipMalloc(&a.ring, (size_t)pairs * SLOTS * SLOT_KIB * 1024);
    hipMalloc(&a.full, pairs * 64); hipMalloc(&a.freed, pairs * 64);
    hipMalloc(&a.abort_word, 64); hipMalloc(&a.xcc, 256 * 4); hipMalloc(&a.bad, 8); hipMalloc(&a.ticks, 256 * 8);
    a.n_slots_total = n_slots; a.place = place;
    double best = 1e30;
    unsigned long long bad = 0;
    unsigned abort_word = 0;
    std::vector<unsigned> xcc(256);
    std::vector<unsigned long long> ticks(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(a.ring, 0xff, (size_t)pairs * SLOTS * SLOT_KIB * 1024);
        hipMemset(a.full, 0, pairs * 64); hipMemset(a.freed, 0, pairs * 64);
        hipMemset(a.abort_word, 0, 64); hipMemset(a.bad, 0, 8);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((handoff<MODE, SLOT_KIB, SLOTS>), dim3(256), dim3(512), 0, 0, a);
        hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        unsigned long long b2 = 0;
        hipMemcpy(&b2, a.bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&abort_word, a.abort_word, 4, hipMemcpyDeviceToHost);
        bad += b2;
        if (abort_word) break;
    }
    hipMemcpy(xcc.data(), a.xcc, 256 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ticks.data(), a.ticks, 256 * 8, hipMemcpyDeviceToHost);
    int same = 0;
    for (int p = 0; p < pairs; ++p) {
        const int g = p / 8, x = p % 8;
        const int prod = 16 * g + x, cons = 16 * g + 8 + (place == 0 ? x : (x + 4) % 8);
        same += xcc[prod] == xcc[cons];
    }
    const double bytes_pair = (double)n_slots * SLOT_KIB * 1024;
    const double gbs_pair = bytes_pair / (best * 1e-3) / 1e9;
    printf("mode %d  slot %3d KiB x %d slots  place %d: %8.3f ms  %7.1f GB/s per pair  %7.2f TB/s over 128 pairs  "
           "(pairs on one XCD: %d/128)  bad dwords %llu%s\n",
           MODE, SLOT_KIB, SLOTS, place, best, gbs_pair, gbs_pair * pairs / 1e3, same, bad, abort_word ? "  ABORTED (a spin ran out)" : "");
    hipFree(a.ring); hipFree(a.full); hipFree(a.freed); hipFree(a.abort_word); hipFree(a.xcc); hipFree(a.bad); hipFree(a.ticks);
}

int main() {
    printf("hand-off of 1 KiB units between workgroup pairs inside one launch; reference: 28 GB/s per unit = HBM read share\n"
           "with all 256 units streaming, 56 GB/s for one unit alone, 7.15 TB/s the chip\n");
    const int n = 2000;
    for (int place = 0; place < 2; ++place) {
        run<0, 32, 4>(place, n);
        run<0, 32, 8>(place, n);
        run<0, 96, 4>(place, n / 3);
        run<1, 32, 4>(place, n);
        run<1, 32, 8>(place, n);
        run<1, 96, 4>(place, n / 3);
    }
    return 0;
}
