// Building blocks of the register-resident MLP chain kernels (forward and backward-data).
// See mlp_common.hpp for the data layout.
//
// A workgroup is 8 waves (2 per SIMD, <= 256 VGPRs each); a wave owns 32 samples for a whole
// pass over the network.  Weights arrive as a stream of 1 KiB A-fragment blocks through an
// LDS ring of RING_SLOTS (3 or 4) 32 KiB chunks filled by LDS-DMA (global_load_lds_dwordx4): while chunk c
// is consumed, the following RING_SLOTS - 1 chunks are in flight or landed.  The only workgroup barrier is at
// chunk entry, paired with a COUNTED vmcnt wait that leaves the newest chunk's DMA in flight.
#pragma once
#include "mlp_common.hpp"

namespace nerf {
namespace mlp {

constexpr int RING_BYTES = RING_SLOTS * CHUNK_BYTES;            // 98304
constexpr int BIAS_BYTES = BIAS_FLOATS * 4;                     // 12672
constexpr int CHAIN_LDS_BYTES = RING_BYTES + BIAS_BYTES;        // 110976
// LDS map: [bias table | ring].  The bias table comes first so that its (compile-time) offsets fit
// the 16-bit immediate of ds_read; ring reads go through one base register plus an immediate.
constexpr int LDS_BIAS_OFF = 0;
constexpr int LDS_RING_OFF = BIAS_BYTES;                        // 12672, a multiple of 16
constexpr int CHAIN_THREADS = 512;
constexpr int DMA_PER_CHUNK = CHUNK_BYTES / (8 * 1024);        // 1 KiB pieces per wave per chunk = 4

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const char gbl_char;

struct Ring {
    const char* stream;     // start of the block stream (wave-uniform)
    uint32_t src_off;       // byte offset of this wave's share of the next chunk to prefetch (wave-uniform)
    uint32_t wave_off;      // wave * 4096: this wave's share inside a chunk / ring slot (wave-uniform)
    uint32_t lane16;        // lane * 16
    uint32_t rd_base;       // LDS address the current chunk is read from: slot*CHUNK_BYTES + lane*16
    uint32_t stream_bytes;  // n_chunks * CHUNK_BYTES
    int slot;               // ring slot of the current chunk
    int extra_enters;       // chunk entries that still have to leave a batch of side copies in flight
};

// Issue this wave's 4 KiB share of the chunk at stream offset `off` into ring slot `slot`.  One
// scalar base + one per-lane 32-bit offset serve all four 1 KiB pieces: the instruction offset
// advances the global source and the LDS destination together.
__device__ __forceinline__ void ring_issue(const Ring& r, char* lds, uint32_t off, int slot) {
    gbl_char* g = (gbl_char*)(r.stream + off) + r.lane16;
    lds_char* l = (lds_char*)(lds + LDS_RING_OFF + slot * CHUNK_BYTES + r.wave_off);
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
}

__device__ __forceinline__ void ring_init(Ring& r, char* lds, const void* stream, int n_chunks, int lane, int wave) {
    r.stream = (const char*)stream;
    r.wave_off = wave * (CHUNK_BYTES / 8);
    r.lane16 = lane * 16;
    r.slot = RING_SLOTS - 1;          // first enter() moves to slot 0
    r.extra_enters = 0;
    r.rd_base = LDS_RING_OFF;
    r.stream_bytes = n_chunks * CHUNK_BYTES;
#pragma unroll
    for (int c = 0; c < RING_SLOTS - 1; ++c) ring_issue(r, lds, r.wave_off + c * CHUNK_BYTES, c);
    r.src_off = r.wave_off + (RING_SLOTS - 1) * CHUNK_BYTES;
}

// Entry into the next chunk of the stream.  On return the chunk is readable at rd_base.
//  - vmcnt(DMA_PER_CHUNK): everything but this wave's newest chunk of DMA has landed, so its
//    share of the chunk about to be read is in LDS;
//  - the barrier extends that to every wave's share, and also says every wave has finished
//    reading the previous chunk, whose slot is the one refilled next.
//  - YOUNGER: a lower bound on the other VMEM operations (stores of saved tensors) this wave issued
//    after it requested the chunk it now waits for; they complete after it (the counter is in order),
//    so they may stay in flight as well.
//  - SIDE: a kernel that issues a batch of SIDE further LDS-DMA copies right after a chunk entry (inputs
//    of its next pass) sets r.extra_enters = 2: at the next two entries the batch is younger than the
//    chunk waited for and may stay in flight; from the third entry on it is older and has landed.
template <int YOUNGER = 0, int SIDE = 0>
__device__ __forceinline__ void ring_enter(Ring& r, char* lds) {
    // chunks requested after the one entered now: RING_SLOTS - 2 of them
    constexpr int DMA_YOUNGER = (RING_SLOTS - 2) * DMA_PER_CHUNK;
    static_assert(DMA_YOUNGER + YOUNGER + SIDE <= 63, "vmcnt is a 6-bit counter");
    if (SIDE > 0 && r.extra_enters > 0) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_YOUNGER + YOUNGER + SIDE) : "memory");
        --r.extra_enters;
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_YOUNGER + YOUNGER) : "memory");
    }
#ifndef NERF_ABL_NO_BARRIER          // timing experiment only: results are wrong without the barrier
    __builtin_amdgcn_s_barrier();
#endif
    r.slot = r.slot == RING_SLOTS - 1 ? 0 : r.slot + 1;
    int refill = r.slot + RING_SLOTS - 1;
    if (refill >= RING_SLOTS) refill -= RING_SLOTS;
    // the offset repeats with the pass; keep it opaque so that it stays one scalar register
    // instead of a hoisted table of per-chunk addresses
    asm volatile("" : "+s"(r.src_off));
    ring_issue(r, lds, r.src_off, refill);
    r.src_off += CHUNK_BYTES;
    if (r.src_off >= r.stream_bytes) r.src_off -= r.stream_bytes;
    r.rd_base = LDS_RING_OFF + r.slot * CHUNK_BYTES + r.lane16;
}

// accumulator tile initialised with the bias of rows 32m .. 32m+31 of `layer`
__device__ __forceinline__ f32x16 bias_tile(const float* bias_lds, int layer, int m, int h) {
    f32x16 acc;
    const float* b = bias_lds + layer * BIAS_STRIDE + 32 * m + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(b + 8 * q);
        acc[4 * q + 0] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
    }
    return acc;
}

// One layer: NT output tiles, each the sum over NK0 k-steps of in0 and NK1 of in1, consumed in the
// stream's pair-interleaved order (stream_tile / stream_ks): two accumulators alternate, so
// consecutive MFMAs never depend on each other.  B0 is the layer's first block in the stream;
// init(m) gives tile m's starting accumulator (the bias) and is called INIT_AHEAD blocks before the
// tile's first MFMA, so that its LDS reads are not waited for; epi(m, acc) consumes the finished tile.
// A fragments are read PF blocks ahead of their MFMA, never across a chunk entry (the data is only
// known to have landed after its barrier).
constexpr int INIT_AHEAD = 8;
// position of tile m's first block inside its layer
constexpr int first_block(int NT, int NK, int m) {
    return (m / 2) * 2 * NK + ((2 * (m / 2) + 1 < NT) ? (m & 1) : 0);
}
#ifndef NERF_PF
#define NERF_PF 4
#endif
constexpr int PF = NERF_PF;

// The A-fragment reads and their waits are written as inline asm.  Left to the compiler, the reads get
// moved next to their use and every group of MFMAs is preceded by `s_waitcnt lgkmcnt(0)`, which waits
// for the read issued just before it: the LDS latency is then exposed every few MFMAs.  Here block
// i+PF is requested, then `lgkmcnt(n)` lets the n newest requests stay in flight, then MFMA i runs.
// LDS operations return in order, so LDS reads the compiler issues in between (bias rows, masks) only
// make these waits, and the compiler's own, wait for more than they need.
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ i32x4 lds_read_b128(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is a 16-bit immediate");
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// wait until at most N LDS operations are outstanding; ties `v` so that its use stays behind the wait
template <int N>
__device__ __forceinline__ void lds_wait(i32x4& v) {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}
// last block (layer-relative) that may have been requested once step i runs: i+PF, clipped to the
// layer and to the chunk that block i lives in
constexpr int pf_hi(int B0, int NB, int i) {
    int hi = i + PF;
    if (hi > NB - 1) hi = NB - 1;
    const int chunk_last = ((B0 + i) / CHUNK_BLOCKS + 1) * CHUNK_BLOCKS - 1 - B0;
    return hi < chunk_last ? hi : chunk_last;
}
// first block to request at step i
constexpr int pf_lo(int B0, int NB, int i) {
    if (i == 0 || (B0 + i) / CHUNK_BLOCKS != (B0 + i - 1) / CHUNK_BLOCKS) return i;
    return pf_hi(B0, NB, i - 1) + 1;
}

// Tile epilogues of this layer that ran during the RING_SLOTS - 1 chunks before block i.  The chunk entered
// at block i was requested that many entries earlier, so everything issued since -- the younger chunks of
// ring DMA and the stores of these epilogues -- may stay in flight; a store then has RING_SLOTS - 1 chunk
// times to be acknowledged before it would hold up a chunk entry.
constexpr int tiles_ended_before(int NT, int NK, int i) {
    int n = 0;
    constexpr int WINDOW = (RING_SLOTS - 1) * CHUNK_BLOCKS;
    for (int j = (i > WINDOW ? i - WINDOW : 0); j < i; ++j) n += (stream_ks(NT, NK, j) == NK - 1);
    return n;
}

// Block k2's request / block i's wait with compile-time offsets and counts.  The loop that calls
// these is fully unrolled, but `i` is still a variable in the source, so the immediates are picked
// by a switch that folds to one case.
template <int OFF0>
__device__ __forceinline__ i32x4 read_block_sw(uint32_t addr, int blk) {
    switch (blk) {
#define NERF_CASE(B) case B: return lds_read_b128<OFF0 + (B) * BLOCK_BYTES>(addr);
        NERF_CASE(0) NERF_CASE(1) NERF_CASE(2) NERF_CASE(3) NERF_CASE(4) NERF_CASE(5) NERF_CASE(6) NERF_CASE(7)
        NERF_CASE(8) NERF_CASE(9) NERF_CASE(10) NERF_CASE(11) NERF_CASE(12) NERF_CASE(13) NERF_CASE(14) NERF_CASE(15)
        NERF_CASE(16) NERF_CASE(17) NERF_CASE(18) NERF_CASE(19) NERF_CASE(20) NERF_CASE(21) NERF_CASE(22) NERF_CASE(23)
        NERF_CASE(24) NERF_CASE(25) NERF_CASE(26) NERF_CASE(27) NERF_CASE(28) NERF_CASE(29) NERF_CASE(30)
#undef NERF_CASE
        default: return lds_read_b128<OFF0 + 31 * BLOCK_BYTES>(addr);
    }
}
__device__ __forceinline__ void wait_sw(i32x4& v, int n) {
    switch (n) {
        case 0: lds_wait<0>(v); break;
        case 1: lds_wait<1>(v); break;
        case 2: lds_wait<2>(v); break;
        case 3: lds_wait<3>(v); break;
        case 4: lds_wait<4>(v); break;
        case 5: lds_wait<5>(v); break;
        case 6: lds_wait<6>(v); break;
        case 7: lds_wait<7>(v); break;
        default: lds_wait<8>(v); break;
    }
}
template <int B0, int NB, class Z>
__device__ __forceinline__ void request_blocks(uint32_t rd_base, i32x4* a, Z, int i) {
    const int lo = pf_lo(B0, NB, i), hi = pf_hi(B0, NB, i);
#pragma unroll
    for (int d = 0; d <= PF; ++d) {
        const int k2 = lo + d;
        if (k2 <= hi) a[k2] = read_block_sw<0>(rd_base, (B0 + k2) % CHUNK_BLOCKS);
    }
}
template <int B0, int NB>
__device__ __forceinline__ void wait_block(i32x4* a, int i) {
    wait_sw(a[i], pf_hi(B0, NB, i) - i);      // requests younger than block i's
}

// VM_PER_TILE: VMEM operations every tile epilogue of this layer is known to issue (0 if unknown).
template <int PREC, int B0, int NT, int NK0, int NK1, int VM_PER_TILE = 0, class Init, class Epi>
__device__ __forceinline__ void run_layer(Ring& ring, char* lds, const typename Prec<PREC>::frag* in0,
                                          const typename Prec<PREC>::frag* in1, Init&& init, Epi&& epi) {
    typedef typename Prec<PREC>::frag frag;
    constexpr int NK = NK0 + NK1, NB = NT * NK;
    i32x4 a[NB];
    f32x16 acc[2], start[2];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int m = stream_tile(NT, NK, i), ks = stream_ks(NT, NK, i);
        const int b = B0 + i;
        if (b % CHUNK_BLOCKS == 0) {
            // folds to one call once the loop is unrolled
            const int younger = VM_PER_TILE * tiles_ended_before(NT, NK, i);
            if (younger >= 16) ring_enter<16>(ring, lds);
            else if (younger >= 12) ring_enter<12>(ring, lds);
            else if (younger >= 8) ring_enter<8>(ring, lds);
            else if (younger >= 4) ring_enter<4>(ring, lds);
            else if (younger >= 2) ring_enter<2>(ring, lds);
            else ring_enter<0>(ring, lds);
        }
        request_blocks<B0, NB>(ring.rd_base, a, std::integral_constant<int, 0>(), i);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // not before the previous tile of the same parity (2*NK blocks earlier) has taken its own
            const int ahead = INIT_AHEAD < 2 * NK - 1 ? INIT_AHEAD : 2 * NK - 1;
            const int f = first_block(NT, NK, t);
            if ((f - ahead > 0 ? f - ahead : 0) == i) start[t & 1] = init(t);
        }
        wait_block<B0, NB>(a, i);
        acc[m & 1] = Prec<PREC>::mfma(__builtin_bit_cast(frag, a[i]), ks < NK0 ? in0[ks] : in1[ks - NK0],
                                      ks == 0 ? start[m & 1] : acc[m & 1]);
        if (ks == NK - 1) epi(m, acc[m & 1]);
    }
}

// max(x, 0) as one integer max on the float's bits (negative floats are negative integers); avoids
// the canonicalising v_max the compiler puts in front of fmaxf on MFMA results.  A negative-signed NaN -- what
// an MFMA returns for a NaN operand -- becomes 0; non-finite parameters are caught where the weights are packed
// instead (PACK_FLAGS_OFF, load_bias_table), which costs the chain nothing (a ReLU that keeps NaNs, one compare
// and one select per value, made the training step 1.6 % slower).
__device__ __forceinline__ float relu_bits(float x) {
    const int i = __float_as_int(x);
    return __int_as_float(i > 0 ? i : 0);
}

// Bias table -> LDS at the start of a forward kernel (plain loads, before any LDS-DMA is in flight), and the
// check of the pack launch's non-finite flags: when any is set, the biases of the rows that become raw --
// rgb_layers.2 and sigma_linear, or output_linear -- are NaN, so every output of the launch is NaN, as in the
// reference.  Ends with a workgroup barrier.
template <bool VIEW>
__device__ __forceinline__ void load_bias_table(float* bias_lds, const char* packed) {
    const float* bg = reinterpret_cast<const float*>(packed + PACK_BIAS_OFF);
    for (int i = threadIdx.x; i < BIAS_FLOATS; i += CHAIN_THREADS) bias_lds[i] = bg[i];
    const uint32_t* flags = reinterpret_cast<const uint32_t*>(packed + PACK_FLAGS_OFF);
    uint32_t bad = 0;
    for (int i = threadIdx.x; i < PACK_FLAG_WORDS; i += CHAIN_THREADS) bad |= flags[i];
    if (__syncthreads_or(bad != 0) && threadIdx.x < 4) {
        const float nan = __int_as_float(0x7fc00000);
        if (VIEW) bias_lds[threadIdx.x < 3 ? 10 * BIAS_STRIDE + threadIdx.x : 8 * BIAS_STRIDE + 256] = nan;
        else bias_lds[8 * BIAS_STRIDE + threadIdx.x] = nan;
    }
    __syncthreads();
}

// accumulator tile -> the two 16-bit fragments it becomes for the next layer
template <int PREC, bool RELU>
__device__ __forceinline__ void tile_to_frags(const f32x16& acc, typename Prec<PREC>::frag& f0,
                                              typename Prec<PREC>::frag& f1) {
    typedef typename Prec<PREC>::elem T;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a = RELU ? relu_bits(acc[j]) : acc[j];
        const float b = RELU ? relu_bits(acc[8 + j]) : acc[8 + j];
        f0[j] = (T)a;
        f1[j] = (T)b;
    }
}

// Store of one lane's 16 bytes of a streamed unit (saved activations, dZ) at the wave-uniform address `unit` plus
// the lane's byte offset.  NERF_UNIT_STORE_POLICY selects the cache policy: 0 = non-temporal (the compiler's
// builtin), 1 = sc1, 2 = sc0 sc1 (inline asm in the scalar-base form; experiment switches).
#ifndef NERF_UNIT_STORE_POLICY
#define NERF_UNIT_STORE_POLICY 0
#endif
template <class F>
__device__ __forceinline__ void unit_store(char* unit, uint32_t lane_off, const F& f) {
    static_assert(sizeof(F) == 16, "one dwordx4 per lane");
#if NERF_UNIT_STORE_POLICY == 0
    __builtin_nontemporal_store(f, reinterpret_cast<F*>(unit + lane_off));
#elif NERF_UNIT_STORE_POLICY == 1
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(lane_off), "v"(f), "s"(unit));
#else
    asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(lane_off), "v"(f), "s"(unit));
#endif
}

// byte offset of (tensor starting at k-step ks0, wave tile wt, k-step ks) in a unit-organised buffer
__device__ __forceinline__ size_t unit_offset(int ks0, int nks, int64_t n_wt, int64_t wt, int ks) {
    return ((size_t)ks0 * n_wt + (size_t)wt * nks + ks) * 1024;
}

}  // namespace mlp
}  // namespace nerf
