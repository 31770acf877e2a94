// Backward of the fused MLP (the autograd of src/nerf_project/models/nerf.py:173-209 with respect
// to the 24 parameter tensors), in three launches:
//
//  1. mlp_dgrad_kernel -- the backward-data chain.  Same structure as the forward: a wave owns 32
//     samples, dZ of a layer (features in registers, samples on lanes) is the MFMA B operand of the
//     next product dH_in = W^T dZ, weights stream through the LDS ring from the transposed block
//     stream.  ReLU masks are the sign bits the forward kept, fetched by LDS-DMA.  Every layer's dZ
//     (except dF, see "factored gradients" in mlp_common.hpp) is written out (16 bit) in
//     [32 samples][16 positions] units for the weight-gradient pass.
//  2. mlp_wgrad_kernel -- dW = sum over samples dZ x In^T: 10 GEMMs whose contraction index is the
//     sample.  Units are DMA-copied into LDS as they are and read back with ds_read_b64_tr_b16,
//     which delivers the [feature][8 samples] fragments the MFMA wants.  The sample axis is
//     split across workgroups; each writes an fp32 partial block.  HBM-bound: every saved unit and
//     every dZ unit is read once (a few are read twice), 128 flop per byte.
//  3. mlp_reduce_kernel -- sums the partial blocks in fixed order (bitwise reproducible) and
//     scatters them from slot order to the flat state_dict-order gradient buffer; mlp_post_kernel then
//     forms the two factored gradients from the reduced product P = dZ9^T H7.
#include <type_traits>
#include "mlp_chain.hpp"

namespace nerf {
namespace mlp {

// ------------------------------------------------------------------------------------------------
// weight-gradient GEMM table
// ------------------------------------------------------------------------------------------------
// A GEMM's row operand is one or two runs of dZ units, its column operand one or two saved tensors:
// operands that two products share are read once (dZ5 meets both halves of the skip layer's input, H7
// meets dZ9 and d sigma, dZ9 meets H7 and the view encoding).
struct GemmDesc {
    int a_layer, a_nks;             // rows: all k-steps of dZ[a_layer] ...
    int a2_layer, a2_nks;           // ... then those of dZ[a2_layer] (a2_nks = 0: none)
    int b_tensor, b_nks;            // cols: saved tensor b_tensor ...
    int b2_tensor, b2_nks;          // ... then b2_tensor (b2_nks = 0: none)
};
constexpr int NGEMM = 10;          // most GEMMs of an architecture
struct ArchTab {
    int ngemm;
    GemmDesc g[NGEMM];
    int cost[NGEMM];      // relative time per wave tile, see WGRAD_COST below
};
// Relative time one workgroup needs per wave tile of each GEMM while every CU is streaming (measured
// with per-workgroup clock stamps).  The kernel as a whole runs at the HBM rate, but a workgroup's own
// rate depends on its GEMM's shape (the copies, LDS reads and MFMAs of a step are serial), so equal
// bytes per workgroup do not end together; the work is divided by this cost instead.
// The 256 x 256 shapes' 120 (131 by the stamps, before their waves 4..7 ran half a step late) is the best of
// 105 / 112 / 120 / 125 / 131 by the time of the whole call.
#ifndef NERF_WG_COST
#define NERF_WG_COST {67, 120, 120, 120, 120, 146, 120, 120, 127, 24}
#endif
#ifndef NERF_WG_COST_NV
#define NERF_WG_COST_NV {67, 120, 120, 120, 120, 146, 120, 120, 50, 0}
#endif
constexpr ArchTab TAB_VIEW = {10, {
    {0, 16, 0, 0, SV_PE, 4, 0, 0},                  // 0: pts_linears.0
    {1, 16, 0, 0, SV_H0 + 0, 16, 0, 0},             // 1: pts_linears.1
    {2, 16, 0, 0, SV_H0 + 1, 16, 0, 0},             // 2
    {3, 16, 0, 0, SV_H0 + 2, 16, 0, 0},             // 3
    {4, 16, 0, 0, SV_H0 + 3, 16, 0, 0},             // 4
    {5, 16, 0, 0, SV_H0 + 4, 16, SV_PE, 4},         // 5: pts_linears.5: columns [H4 | PE]
    {6, 16, 0, 0, SV_H0 + 5, 16, 0, 0},             // 6: pts_linears.6
    {7, 16, 0, 0, SV_H0 + 6, 16, 0, 0},             // 7: pts_linears.7
    // 8: rows [dZ9 | d sigma] x columns [H7 | VPE]: P = dZ9^T H7 and sum dZ9 (feature_linear and
    //    rgb_layers.0[:, :256], factored), sigma_linear (row 128), rgb_layers.0[:, 256:]
    {9, 8, 8, 1, SV_H0 + 7, 16, SV_VPE, 2},
    {10, 1, 0, 0, SV_G, 8, 0, 0},                   // 9: rgb_layers.2 (rows 0..2)
}, NERF_WG_COST};
// the network without view directions: GEMMs 0..7 as above, then output_linear (rows 0..3 of dZ_o, kept
// where the full network keeps dZ10) x H7
constexpr ArchTab TAB_NOVIEW = {9, {
    {0, 16, 0, 0, SV_PE, 4, 0, 0}, {1, 16, 0, 0, SV_H0 + 0, 16, 0, 0}, {2, 16, 0, 0, SV_H0 + 1, 16, 0, 0},
    {3, 16, 0, 0, SV_H0 + 2, 16, 0, 0}, {4, 16, 0, 0, SV_H0 + 3, 16, 0, 0}, {5, 16, 0, 0, SV_H0 + 4, 16, SV_PE, 4},
    {6, 16, 0, 0, SV_H0 + 5, 16, 0, 0}, {7, 16, 0, 0, SV_H0 + 6, 16, 0, 0},
    {10, 1, 0, 0, SV_H0 + 7, 16, 0, 0},             // 8: output_linear
    {0, 0, 0, 0, 0, 0, 0, 0},
}, NERF_WG_COST_NV};
template <bool VIEW> __host__ __device__ constexpr const ArchTab& arch_tab() { return VIEW ? TAB_VIEW : TAB_NOVIEW; }

__host__ __device__ constexpr int gemm_a_nks(const ArchTab& t, int g) { return t.g[g].a_nks + t.g[g].a2_nks; }
__host__ __device__ constexpr int gemm_b_nks(const ArchTab& t, int g) { return t.g[g].b_nks + t.g[g].b2_nks; }
// partial block of GEMM g: [rows][cols] floats, then one bias sum per row
__host__ __device__ constexpr int gemm_out_floats(const ArchTab& t, int g) {
    return gemm_a_nks(t, g) * 16 * gemm_b_nks(t, g) * 16 + gemm_a_nks(t, g) * 16;
}
__host__ __device__ constexpr int total_cost(const ArchTab& t) { int c = 0; for (int g = 0; g < t.ngemm; ++g) c += t.cost[g]; return c; }

constexpr int WGRAD_TARGET_JOBS = 256;     // one workgroup per CU

// One launch computes the weight gradients of up to MAX_NETS networks (the coarse and the fine network of a
// training step): their GEMMs form one list, "virtual GEMM" v = net * NGEMM + g.  With two networks each
// GEMM is split over half as many workgroups as in a launch of its own, so half as many partial blocks are
// written and summed.
constexpr int MAX_NETS = NERF_WGRAD_MAX_NETS;
constexpr int NVG = MAX_NETS * NGEMM;
struct NetSet {
    int n_nets;
    int view[MAX_NETS];        // 1: full network, 0: without view directions
    int64_t n_wt[MAX_NETS];    // wave tiles (32 samples) of each network
};
__host__ __device__ inline const ArchTab& tab_of(const NetSet& s, int net) { return s.view[net] ? TAB_VIEW : TAB_NOVIEW; }

// The whole job -- every GEMM of every network over all its wave tiles, laid end to end on one line measured
// in cost -- is cut into WGRAD_TARGET_JOBS pieces of equal cost; workgroup p owns piece p.  A piece that
// crosses the end of a GEMM continues with the first wave tiles of the next one, so a workgroup runs one to
// three GEMM segments in turn and all workgroups end together whatever the GEMMs' sizes.  Inside a GEMM
// a piece boundary at cost b falls on wave tile floor((b - start of the GEMM) / cost of the GEMM).
__host__ __device__ inline int64_t line_total(const NetSet& s) {
    int64_t c = 0;
    for (int n = 0; n < s.n_nets; ++n) c += s.n_wt[n] * total_cost(tab_of(s, n));
    return c;
}
__host__ __device__ inline int64_t piece_bound(const NetSet& s, int p) {
    return (int64_t)p * line_total(s) / WGRAD_TARGET_JOBS;
}
__host__ __device__ inline int64_t bound_to_wt(const NetSet& s, int net, int g, int64_t b) {
    int64_t start = 0;
    for (int n = 0; n < net; ++n) start += s.n_wt[n] * total_cost(tab_of(s, n));
    const ArchTab& t = tab_of(s, net);
    int before = 0;
    for (int i = 0; i < g; ++i) before += t.cost[i];
    start += s.n_wt[net] * before;
    if (b < start) return 0;
    const int64_t w = (b - start) / t.cost[g];
    return w > s.n_wt[net] ? s.n_wt[net] : w;
}
// wave tiles [wt0, wt1) of GEMM g of network `net` that belong to piece p (empty when wt1 <= wt0)
__host__ __device__ inline void piece_segment(const NetSet& s, int net, int g, int p, int64_t& wt0, int64_t& wt1) {
    wt0 = bound_to_wt(s, net, g, piece_bound(s, p));
    wt1 = bound_to_wt(s, net, g, piece_bound(s, p + 1));
}

struct JobTable {
    int splits[NVG];       // pieces that hold a segment of each GEMM = partial blocks of that GEMM
    int piece0[NVG];       // the first of them; a GEMM's pieces are consecutive
    int64_t part_off[NVG]; // float offset of the GEMM's first partial block in the partial buffer
};

static JobTable make_jobs(const NetSet& s, int64_t* total_floats) {
    JobTable j;
    int64_t off = 0;
    for (int v = 0; v < NVG; ++v) {
        const int net = v / NGEMM, g = v % NGEMM;
        j.splits[v] = 0; j.piece0[v] = 0; j.part_off[v] = off;
        if (net >= s.n_nets || g >= tab_of(s, net).ngemm) continue;
        // first and last piece with wave tiles of this GEMM; with very few wave tiles a piece in between
        // can be empty: it then writes a zero block, so that the GEMM's blocks stay one range
        int first = -1, last = -1;
        for (int p = 0; p < WGRAD_TARGET_JOBS; ++p) {
            int64_t a, b;
            piece_segment(s, net, g, p, a, b);
            if (b > a) {
                if (first < 0) first = p;
                last = p;
            }
        }
        j.piece0[v] = first;
        j.splits[v] = last - first + 1;
        off += (int64_t)j.splits[v] * gemm_out_floats(tab_of(s, net), g);
    }
    *total_floats = off;
    return j;
}

// ------------------------------------------------------------------------------------------------
// 1. backward-data chain
// ------------------------------------------------------------------------------------------------
struct BwdArgs {
    int64_t M, n_tiles;
    const float* d_raw;      // [M,4]
    const char* packed;
    const char* saved;
    char* dz;                // DZ_UNITS KiB per wave tile
    const float* scale;      // nullable: gradients are multiplied by *scale on entry (fp16 range)
};

template <int PREC>
__device__ __forceinline__ typename Prec<PREC>::frag load_unit_frag(const char* base, int ks0, int nks, int64_t n_wt,
                                                                    int64_t wt, int ks, int lane_off) {
    return *reinterpret_cast<const typename Prec<PREC>::frag*>(base + unit_offset(ks0, nks, n_wt, wt, ks) + lane_off);
}

template <int PREC>
__device__ __forceinline__ void store_unit_frag(char* base, int ks0, int nks, int64_t n_wt, int64_t wt, int ks,
                                                int lane_off, const typename Prec<PREC>::frag& f) {
#ifdef NERF_ABL_STORE_LOCAL      // timing experiment only: every store of a wave goes to one L2-resident KiB
    *reinterpret_cast<typename Prec<PREC>::frag*>(base + (size_t)(blockIdx.x * 8 + (threadIdx.x >> 6)) * 1024 + lane_off) = f;
    return;
#endif
#ifdef NERF_ABL_PLAIN_STORE      // timing experiment: default cache policy instead of non-temporal
    *reinterpret_cast<typename Prec<PREC>::frag*>(base + unit_offset(ks0, nks, n_wt, wt, ks) + lane_off) = f;
    return;
#endif
    unit_store(base + unit_offset(ks0, nks, n_wt, wt, ks), (uint32_t)lane_off, f);
}

// accumulator tile (gradient w.r.t. a post-ReLU activation) -> 16-bit fragments, zeroed where the
// activation was not positive.  `signs` is this lane's mask dword of the tile pair, PAR the tile's
// parity in the pair (see mask_bit).
template <int PREC, bool MASK, int PAR>
__device__ __forceinline__ void grad_to_frags(const f32x16& acc, uint32_t signs, typename Prec<PREC>::frag& f0,
                                              typename Prec<PREC>::frag& f1) {
    typedef typename Prec<PREC>::elem T;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = acc[j], b = acc[8 + j];
        if (MASK) {   // bit -> 0 / all ones, then and
            a = __int_as_float(__float_as_int(a) & __builtin_amdgcn_sbfe((int)signs, mask_bit(PAR, j), 1));
            b = __int_as_float(__float_as_int(b) & __builtin_amdgcn_sbfe((int)signs, mask_bit(PAR, 8 + j), 1));
        }
        f0[j] = (T)a;
        f1[j] = (T)b;
    }
}

// Sign masks travel global -> LDS by DMA, one KiB per wave and tensor, into three per-wave buffers
// behind the ring: two alternate between consecutive hidden layers, the third holds the masks of G
// for the next pass.  Each wave reads only what it copied itself, so no barrier is involved; the
// copy is issued a whole layer (>= 2 chunk entries) before its first read, and mask_landed() before
// that read waits until at most the 8 newest VMEM operations of the wave are outstanding - which
// are the newest chunk of ring DMA and the dZ stores of the last two tiles, never the mask copy.
constexpr int MASK_LDS_OFF = CHAIN_LDS_BYTES;
constexpr int MASK_BUF_BYTES = 8 * 1024;
constexpr int DGRAD_LDS_BYTES = CHAIN_LDS_BYTES + 3 * MASK_BUF_BYTES;
enum { MB_A = 0, MB_B = 1, MB_G = 2 };

__device__ __forceinline__ void mask_issue(const char* masks, int64_t wt, int tensor, char* lds, int buf, int wave,
                                           int lane) {
    gbl_char* g = (gbl_char*)(masks + ((size_t)wt * MASK_TENSORS + tensor) * 1024) + lane * 16;
    lds_char* l = (lds_char*)(lds + MASK_LDS_OFF + buf * MASK_BUF_BYTES + wave * 1024);
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 2);      // read once: non-temporal
}
__device__ __forceinline__ void mask_landed() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }

// VIEW = false: the network without view directions (mlp_common.hpp): dZ_o = d raw is the chain's input
template <int PREC, bool VIEW>
__global__ __launch_bounds__(CHAIN_THREADS, 2) void mlp_dgrad_kernel(const BwdArgs a) {
    typedef typename Prec<PREC>::frag frag;
    typedef typename Prec<PREC>::elem T;
    __shared__ __attribute__((aligned(16))) char lds[DGRAD_LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef NERF_PRIO_UPPER        // experiment: static priority for the later-dispatched half of the workgroup
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    const int c = lane & 31, h = lane >> 5;
#ifdef NERF_ABL_LANE_LINEAR      // timing experiment only (wrong layout): lanes store in address order
    const int lane_off = lane * 16;
#else
    const int lane_off = c * 32 + h * 16;
#endif
    const float gscale = a.scale ? *a.scale : 1.0f;
    const int64_t n_wt = a.n_tiles * 8;
    const char* const masks = a.saved + (size_t)n_wt * SAVED_UNITS * 1024;
    const char* const mask_rd = lds + MASK_LDS_OFF + wave * 1024 + lane * 4;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // masks of the first masked layer of the first pass (G; without view directions H7) go first: the
    // ring's 8 initial copies are younger
    if ((int64_t)blockIdx.x < a.n_tiles)
        mask_issue(masks, (int64_t)blockIdx.x * 8 + wave, VIEW ? MASK_G : 7, lds, VIEW ? MB_G : MB_A, wave, lane);
    Ring ring;
    ring_init(ring, lds, a.packed + PACK_BWD_OFF, VIEW ? BWD_CHUNKS : NV_BWD_CHUNKS, lane, wave);

    for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t wt = tile * 8 + wave;
        const int64_t m = wt * WAVE_SAMPLES + c;
        const bool valid = m < a.M;
        if (VIEW) mask_issue(masks, wt, 7, lds, MB_A, wave, lane);

        float4 dr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && h == 0) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.d_raw) + m);
            dr = make_float4(v[0], v[1], v[2], v[3]);
        }
        dr.x *= gscale; dr.y *= gscale; dr.z *= gscale; dr.w *= gscale;

        frag da[H_KS + 1], db[H_KS];
        uint32_t mk[4];       // masks are fetched ahead of their tile: four tiles can be pending

        // dZ10: slots 0..2 = d rgb_raw (lane half 0, elements 0..2); without view directions slot 3 = d sigma_raw
        // as well: the four outputs of output_linear
#pragma unroll
        for (int j = 0; j < 8; ++j) db[0][j] = (T)0.0f;
        db[0][0] = (T)dr.x; db[0][1] = (T)dr.y; db[0][2] = (T)dr.z;
        if (!VIEW) db[0][3] = (T)dr.w;
        store_unit_frag<PREC>(a.dz, dz_ks0(10), DZ_KS[10], n_wt, wt, 0, lane_off, db[0]);

        // generic masked step: acc tile mt of d(activation saved as tensor SV) -> out[2mt], out[2mt+1],
        // stored as dZ of layer DZL
        // tile mt of a masked layer: its mask dword is read from LDS buffer BUF one tile early (with
        // the accumulator init) and applied when the tile is finished
        auto mask_pre = [&](auto BUF) {
            return [&](int mt) {
                constexpr int buf = decltype(BUF)::value;
                mk[mt & 3] = *reinterpret_cast<const uint32_t*>(mask_rd + buf * MASK_BUF_BYTES + (mt >> 1) * 256);
                return zero;
            };
        };
        auto mask_epi = [&](auto DZL, frag* out) {
            return [&, out](int mt, const f32x16& acc) {
                constexpr int dzl = decltype(DZL)::value;
                if (mt & 1) grad_to_frags<PREC, true, 1>(acc, mk[mt & 3], out[2 * mt], out[2 * mt + 1]);
                else grad_to_frags<PREC, true, 0>(acc, mk[mt & 3], out[2 * mt], out[2 * mt + 1]);
                store_unit_frag<PREC>(a.dz, dz_ks0(dzl), DZ_KS[dzl], n_wt, wt, 2 * mt, lane_off, out[2 * mt]);
                store_unit_frag<PREC>(a.dz, dz_ks0(dzl), DZ_KS[dzl], n_wt, wt, 2 * mt + 1, lane_off, out[2 * mt + 1]);
            };
        };
#define NERF_IC(N) std::integral_constant<int, N>()
        mask_landed();
        ring_enter<0>(ring, lds);            // chunk 0 of the pass: the first layer starts behind the stream's front pad
        if (!VIEW) {
            // Before each masked layer: its masks have landed; the next layer's start their way into the other buffer.
#define NERF_MASKED_NV(E, NK0, IN, BUF, NEXT_ISSUE, DZL, OUT)                                               \
            mask_landed();                                                                                  \
            NEXT_ISSUE;                                                                                     \
            run_layer<PREC, nv_bwd_block0(E), 8, NK0, 0, 2>(ring, lds, IN, IN, mask_pre(NERF_IC(BUF)), mask_epi(NERF_IC(DZL), OUT));
            // e0: dH7 = W_o^T dZ_o, masked by H7 > 0 -> dZ7 in da
            NERF_MASKED_NV(0, 1, db, MB_A, mask_issue(masks, wt, 6, lds, MB_B, wave, lane), 7, da)
            // e1..e6: hidden layers 6, 5 (H rows), 4, 3, 2, 1
            NERF_MASKED_NV(1, H_KS, da, MB_B, mask_issue(masks, wt, 5, lds, MB_A, wave, lane), 6, db)
            NERF_MASKED_NV(2, H_KS, db, MB_A, mask_issue(masks, wt, 4, lds, MB_B, wave, lane), 5, da)
            NERF_MASKED_NV(3, H_KS, da, MB_B, mask_issue(masks, wt, 3, lds, MB_A, wave, lane), 4, db)
            NERF_MASKED_NV(4, H_KS, db, MB_A, mask_issue(masks, wt, 2, lds, MB_B, wave, lane), 3, da)
            NERF_MASKED_NV(5, H_KS, da, MB_B, mask_issue(masks, wt, 1, lds, MB_A, wave, lane), 2, db)
            NERF_MASKED_NV(6, H_KS, db, MB_A, mask_issue(masks, wt, 0, lds, MB_B, wave, lane), 1, da)
            // e7: hidden layer 0; the copy started here is the next pass's H7 mask
            NERF_MASKED_NV(7, H_KS, da, MB_B,
                           if (tile + gridDim.x < a.n_tiles) mask_issue(masks, (tile + gridDim.x) * 8 + wave, 7, lds, MB_A, wave, lane),
                           0, db)
#undef NERF_MASKED_NV
            continue;
        }
        // d0: dG = W_r2^T dZ10, masked by G > 0  -> dZ9 (8 k-steps) in da
        run_layer<PREC, bwd_block0(0), 4, 1, 0, 2>(ring, lds, db, db, mask_pre(NERF_IC(MB_G)), mask_epi(NERF_IC(9), da));
        // d1: dF = W_r0[:, :256]^T dZ9 (feature_linear has no activation) -> dZ8[0..255] in db; sigma joins as k-step 16
        // (dF stays in registers: the weight gradients that would read it are factored, mlp_common.hpp)
        run_layer<PREC, bwd_block0(1), 8, G_KS, 0, 0>(ring, lds, da, da, [&](int) { return zero; },
                                                  [&](int mt, const f32x16& acc) {
            frag f0 = {}, f1 = {};
            grad_to_frags<PREC, false, 0>(acc, 0u, f0, f1);
            db[2 * mt] = f0; db[2 * mt + 1] = f1;
        });
        frag dsig;
#pragma unroll
        for (int j = 0; j < 8; ++j) dsig[j] = (T)0.0f;
        dsig[0] = (T)dr.w;                                   // slot 256 = sigma_raw (lane half 0, element 0)
        store_unit_frag<PREC>(a.dz, dz_ks0(8), DZ_KS[8], n_wt, wt, 0, lane_off, dsig);
        // d2: dH7 = [W_f; w_sigma]^T dZ8, masked by H7 > 0 -> dZ7 in da.  Before each masked layer: its
        // masks have landed; the next layer's start their way into the other buffer.
#define NERF_MASKED(D, NK1, IN, IN1, BUF, NEXT_TENSOR, NEXT_BUF, DZL, OUT)                                   \
        mask_landed();                                                                                       \
        mask_issue(masks, wt, NEXT_TENSOR, lds, NEXT_BUF, wave, lane);                                       \
        run_layer<PREC, bwd_block0(D), 8, H_KS, NK1, 2>(ring, lds, IN, IN1, mask_pre(NERF_IC(BUF)), mask_epi(NERF_IC(DZL), OUT));
        NERF_MASKED(2, 1, db, &dsig, MB_A, 6, MB_B, 7, da)
        // d3..d8: hidden layers 6, 5 (H rows), 4, 3, 2, 1
        NERF_MASKED(3, 0, da, da, MB_B, 5, MB_A, 6, db)
        NERF_MASKED(4, 0, db, db, MB_A, 4, MB_B, 5, da)
        NERF_MASKED(5, 0, da, da, MB_B, 3, MB_A, 4, db)
        NERF_MASKED(6, 0, db, db, MB_A, 2, MB_B, 3, da)
        NERF_MASKED(7, 0, da, da, MB_B, 1, MB_A, 2, db)
        NERF_MASKED(8, 0, db, db, MB_A, 0, MB_B, 1, da)
        // d9: hidden layer 0; the copy started here is the next pass's G mask
        mask_landed();
        if (tile + gridDim.x < a.n_tiles) mask_issue(masks, (tile + gridDim.x) * 8 + wave, MASK_G, lds, MB_G, wave, lane);
        run_layer<PREC, bwd_block0(9), 8, H_KS, 0, 2>(ring, lds, da, da, mask_pre(NERF_IC(MB_B)), mask_epi(NERF_IC(0), db));
#undef NERF_MASKED
#undef NERF_IC
        // The stream is padded to whole chunks; the pad blocks of the last chunk are never read, and
        // the next pass enters a fresh chunk at its block 0.
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// 2. weight-gradient GEMMs
// ------------------------------------------------------------------------------------------------
constexpr int WG_STAGES = 4;                  // one being read, up to three in flight
constexpr int WG_STAGE_BYTES = 36 * 1024;     // up to 36 units: A units first, B units right behind them
#ifndef NERF_WG_DMA_AUX
#define NERF_WG_DMA_AUX 2
#endif
constexpr int WG_DMA_AUX = NERF_WG_DMA_AUX;    // cache policy bits of the streaming copies (2 = nt)
constexpr int WG_LDS_BYTES = WG_STAGES * WG_STAGE_BYTES;

struct WgradArgs {
    NetSet nets;
    const char* saved[MAX_NETS];
    const char* dz[MAX_NETS];
    float* partial;
    JobTable jobs;
#ifdef NERF_WG_STAMP
    unsigned long long* stamps;      // diagnostic builds: per workgroup {start, end} of s_memrealtime (100 MHz), or null
#endif
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// Transposing LDS read as inline asm.  The compiler cannot see which LDS bytes an LDS-DMA wrote, so
// in front of a builtin LDS read it waits for EVERY outstanding DMA (vmcnt(0)), which would
// serialise the staging pipeline; the asm form leaves the wait to the counted vmcnt + barrier of the
// stage protocol.  Results are valid after tr_wait() on the same registers.
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(uint32_t addr) {
    s16x4 r;
#ifdef NERF_WG_NO_LDSREAD
    r[0] = (short)(0x3000 | (addr & 0xff)); r[1] = (short)(0x3000 | ((addr >> 3) & 0xff)); r[2] = (short)(0x3000 | (OFF & 0xff)); r[3] = 0x3001;   // small finite values in either operand type
    asm volatile("" : "+v"(r));
    return r;
#endif
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
struct TrFrag { s16x4 lo, hi; };
// fragment (8 x 16 bit) for one 32-position tile and one 16-sample half: samples 4r..4r+3 of the
// lane's 8 come from read r
template <int OFF>
__device__ __forceinline__ TrFrag tr_frag(uint32_t addr) {
    TrFrag f;
    f.lo = tr_read<OFF>(addr);
    f.hi = tr_read<OFF + 128>(addr);
    return f;
}
// wait until at most N LDS operations of this wave are outstanding, and tie the wait to the
// fragments so that the compiler cannot move their uses above it
template <int N>
__device__ __forceinline__ void tr_wait2(TrFrag& a, TrFrag& b) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void tr_wait1(TrFrag& a) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.lo), "+v"(a.hi) : "n"(N) : "memory");
}
template <int TI, int TJ, int N>
__device__ __forceinline__ void tr_wait_all(TrFrag (&ta)[TI], TrFrag (&tb)[TJ]) {
    static_assert(N <= 15, "lgkmcnt is a 4-bit counter");
#pragma unroll
    for (int i = 0; i < TI; ++i) tr_wait1<N>(ta[i]);
#pragma unroll
    for (int j = 0; j < TJ; ++j) tr_wait1<N>(tb[j]);
}
__device__ __forceinline__ s16x8 tr_join(const TrFrag& f) {
    s16x8 r;
    r[0] = f.lo[0]; r[1] = f.lo[1]; r[2] = f.lo[2]; r[3] = f.lo[3];
    r[4] = f.hi[0]; r[5] = f.hi[1]; r[6] = f.hi[2]; r[7] = f.hi[3];
    return r;
}

template <int PREC>
__device__ __forceinline__ f32x16 wg_mfma(const s16x8& a, const s16x8& b, const f32x16& c) {
    typedef typename Prec<PREC>::frag frag;
#ifdef NERF_WG_NO_MFMA          // timing experiment only (wrong results): the fragments are consumed, no matrix instruction runs
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#endif
    return Prec<PREC>::mfma(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c);
}

template <int PREC>
__device__ __forceinline__ float frag_sum(const s16x8& f) {
    typedef typename Prec<PREC>::frag frag;
    const frag v = __builtin_bit_cast(frag, f);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)v[j];
    return s;
}

// One GEMM shape: A_NKS row units x B_NKS column units, computed by a WR x WC grid of waves, each
// owning TI x TJ tiles of 32 x 32.  Everything about the shape is a compile-time constant so that the
// per-step body is straight-line code.
// NSUB wave tiles are staged and contracted per step (GEMMs with few units would otherwise keep too few
// bytes in flight to cover the memory latency).
template <int PREC, int A_NKS, int B_NKS, int WR, int WC, int TI, int TJ, int NSUB = 1>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const ArchTab& tab, const int net, const int g, char* lds,
                                           const int split, const int64_t wt0, const int64_t wt1) {
    typedef typename Prec<PREC>::frag frag;
    static_assert(WR * WC <= 8 && WR * TI * 2 >= A_NKS && WC * TJ * 2 == B_NKS, "wave grid must cover the block");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const bool active = wave < WR * WC;
    const GemmDesc gd = tab.g[g];

    const int ntiles = wt1 > wt0 ? (int)(wt1 - wt0) : 0;
    const int nsteps = (ntiles + NSUB - 1) / NSUB;

    // staging: a stage holds NSUB wave tiles, each N_UNITS units (A units first, then B units); unit v of
    // the stage is copied by wave v % 8
    constexpr int N_UNITS = A_NKS + B_NKS;
    constexpr int STAGE_UNITS = N_UNITS * NSUB;
    static_assert(STAGE_UNITS * 1024 <= WG_STAGE_BYTES && STAGE_UNITS <= 40, "stage size");
    const int my_units = (STAGE_UNITS - wave + 7) >> 3;       // 0..5
    // the four operand runs: base of wave tile 0, bytes per wave tile, first unit of the run in the stage
    const int a1n = gd.a_nks, b1n = gd.b_nks;
    const int64_t n_wt = a.nets.n_wt[net];
    const char* a1_src = a.dz[net] + (size_t)dz_ks0(gd.a_layer) * n_wt * 1024;
    const char* a2_src = a.dz[net] + (size_t)dz_ks0(gd.a2_layer) * n_wt * 1024;
    const char* b1_src = a.saved[net] + (size_t)saved_ks0(gd.b_tensor) * n_wt * 1024;
    const char* b2_src = a.saved[net] + (size_t)saved_ks0(gd.b2_tensor) * n_wt * 1024;
    const size_t a1_stride = (size_t)DZ_KS[gd.a_layer] * 1024, a2_stride = (size_t)DZ_KS[gd.a2_layer] * 1024;
    const size_t b1_stride = (size_t)SAVED_KS[gd.b_tensor] * 1024, b2_stride = (size_t)SAVED_KS[gd.b2_tensor] * 1024;

    auto issue = [&](int step, int stage) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int v = wave + 8 * i;
            if (v < STAGE_UNITS) {
                const int sub = v / N_UNITS, u = v - sub * N_UNITS;
                // a wave tile past the end of the range (last step) is copied from the last valid one, so that
                // every stage is the same number of copies for the counted waits; it is not contracted
                int64_t wt = wt0 + (int64_t)step * NSUB + sub;
                if (wt >= wt1) wt = wt1 - 1;
#ifdef NERF_WG_L2_OPERANDS      // timing experiment only (wrong results): every step copies the segment's first wave tile
                wt = wt0;       // again, so the operands come from L2 and the time left is the step loop's own
#endif
                const char* src;
                if (u < a1n) src = a1_src + (size_t)wt * a1_stride + (size_t)u * 1024;
                else if (u < A_NKS) src = a2_src + (size_t)wt * a2_stride + (size_t)(u - a1n) * 1024;
                else if (u < A_NKS + b1n) src = b1_src + (size_t)wt * b1_stride + (size_t)(u - A_NKS) * 1024;
                else src = b2_src + (size_t)wt * b2_stride + (size_t)(u - A_NKS - b1n) * 1024;
                char* dst = lds + stage * WG_STAGE_BYTES + v * 1024;
                __builtin_amdgcn_global_load_lds((gbl_char*)src + lane * 16, (lds_char*)dst, 16, 0, WG_DMA_AUX);
            }
        }
    };
    // wait until at most `keep_stages` stages of this wave's copies are still in flight
    auto wait_stages = [&](int keep_stages) {
        const int n = keep_stages * my_units;
        if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    };

    const uint32_t lds_base = (uint32_t)(uintptr_t)((lds_char*)lds);   // LDS byte address of the staging area
    // per-lane part of the transposing-read address (see tr_frag): 16-lane group gg, lane t4 in it
    const int gg = lane >> 4, t4 = lane & 15;
    const uint32_t lane_rd = (8 * (gg >> 1) + (t4 >> 2)) * 32 + (t4 & 3) * 8;
    // unit of this lane's half of a 32-position tile, clamped to the operand (extra rows are dropped later)
    uint32_t a_unit[TI], b_unit[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int u = 2 * (TI * wr + i) + (gg & 1);
        a_unit[i] = lds_base + (u < A_NKS ? u : A_NKS - 1) * 1024 + lane_rd;
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int u = 2 * (TJ * wc + j) + (gg & 1);
        b_unit[j] = lds_base + (A_NKS + u) * 1024 + lane_rd;
    }

    f32x16 acc[TI][TJ];
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = zero;
    float bsum[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) bsum[i] = 0.f;
    const bool do_bias = wc == 0;

    if (nsteps > 0) issue(0, 0);
    if (nsteps > 1) issue(1, 1);
    if (nsteps > 2) issue(2, 2);
    int stage = 0;
#ifndef NERF_WG_NO_STAGGER
    // A SIMD holds waves w and w + 4 of the workgroup.  After a barrier both would read their fragments from LDS (MFMA
    // pipe idle) and then both queue MFMAs (LDS idle).  In the 256 x 256 shapes waves 4..7 therefore run half a step
    // late: they start a step with the MFMAs of the previous stage's second sample half (fragments kept in registers
    // across the barrier) and read this stage's fragments behind them, so that on every SIMD one wave's LDS reads sit
    // beside the other's MFMAs.  The ring protocol is the same (every wave has a stage's fragments in registers before
    // the next barrier) and each wave adds the halves in the same order.  The late waves run a loop of their own: both
    // orders inside one loop, or more than eight tiles per wave, make the kernel spill.
    if (NSUB == 1 && TI * TJ <= 8 && wave >= 4) {
        TrFrag ta0[TI], tb0[TJ], ta1[TI], tb1[TJ];             // the two 16-sample halves of the wave tile
        auto contract_half = [&](TrFrag (&ta)[TI], TrFrag (&tb)[TJ]) {
            s16x8 af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = tr_join(ta[i]);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = tr_join(tb[j]);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < TI; ++i) bsum[i] += frag_sum<PREC>(af[i]);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = wg_mfma<PREC>(af[i], bf[j], acc[i][j]);
        };
        constexpr int SECOND_HALF_READS = 2 * (TI + TJ) < 15 ? 2 * (TI + TJ) : 15;
        for (int step = 0; step < nsteps; ++step) {
            const int younger = nsteps - 1 - step;
            wait_stages(younger < 2 ? younger : 2);
            __builtin_amdgcn_s_barrier();
            if (step + 3 < nsteps) {
                int st3 = stage + 3; if (st3 >= WG_STAGES) st3 -= WG_STAGES;
                issue(step + 3, st3);
            }
            if (active) {
                const uint32_t sb = stage * WG_STAGE_BYTES;
                if (step > 0) contract_half(ta1, tb1);
#pragma unroll
                for (int i = 0; i < TI; ++i) ta0[i] = tr_frag<0>(sb + a_unit[i]);
#pragma unroll
                for (int j = 0; j < TJ; ++j) tb0[j] = tr_frag<0>(sb + b_unit[j]);
#pragma unroll
                for (int i = 0; i < TI; ++i) ta1[i] = tr_frag<512>(sb + a_unit[i]);
#pragma unroll
                for (int j = 0; j < TJ; ++j) tb1[j] = tr_frag<512>(sb + b_unit[j]);
                tr_wait_all<TI, TJ, SECOND_HALF_READS>(ta0, tb0);
                contract_half(ta0, tb0);
                tr_wait_all<TI, TJ, 0>(ta1, tb1);              // in registers before the barrier that frees the slot
            }
            stage = stage + 1 == WG_STAGES ? 0 : stage + 1;
        }
        if (active && nsteps > 0) contract_half(ta1, tb1);
    } else
#endif
    for (int step = 0; step < nsteps; ++step) {
        // stage `step` landed (this wave's share); up to two younger stages may stay in flight
        const int younger = nsteps - 1 - step;
        wait_stages(younger < 2 ? younger : 2);
        __builtin_amdgcn_s_barrier();
        if (step + 3 < nsteps) {
            int st3 = stage + 3; if (st3 >= WG_STAGES) st3 -= WG_STAGES;
            // Its slot was read in the previous step (all waves passed the barrier since).  The copies go out first:
            // issued behind the first half's MFMAs instead, the pass ran 30 % slower -- a stage has three steps to
            // arrive, and on the whole chip that is about the latency of a loaded HBM.
            issue(step + 3, st3);
        }
        if (active) {
            const uint32_t sb = stage * WG_STAGE_BYTES;
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                if (NSUB > 1 && step * NSUB + sub >= ntiles) break;      // workgroup-uniform
                constexpr int SUB_BYTES = N_UNITS * 1024;
                TrFrag ta0[TI], tb0[TJ], ta1[TI], tb1[TJ];     // the two 16-sample halves of the wave tile
#define NERF_TR(OFF, ADDR) (sub == 0 ? tr_frag<OFF>(ADDR) : sub == 1 ? tr_frag<OFF + (NSUB > 1 ? 1 : 0) * SUB_BYTES>(ADDR) \
                            : sub == 2 ? tr_frag<OFF + (NSUB > 2 ? 2 : 0) * SUB_BYTES>(ADDR) : tr_frag<OFF + (NSUB > 3 ? 3 : 0) * SUB_BYTES>(ADDR))
#pragma unroll
                for (int i = 0; i < TI; ++i) ta0[i] = NERF_TR(0, sb + a_unit[i]);
#pragma unroll
                for (int j = 0; j < TJ; ++j) tb0[j] = NERF_TR(0, sb + b_unit[j]);
#pragma unroll
                for (int i = 0; i < TI; ++i) ta1[i] = NERF_TR(512, sb + a_unit[i]);
#pragma unroll
                for (int j = 0; j < TJ; ++j) tb1[j] = NERF_TR(512, sb + b_unit[j]);
#undef NERF_TR
                // both halves were requested above; the first is complete once only the second half's reads (two per
                // fragment, LDS returns in order) are outstanding, and those land behind the first half's MFMAs
                constexpr int SECOND_HALF_READS = 2 * (TI + TJ) < 15 ? 2 * (TI + TJ) : 15;
                tr_wait_all<TI, TJ, SECOND_HALF_READS>(ta0, tb0);
                s16x8 af[TI], bf[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = tr_join(ta0[i]);
#pragma unroll
                for (int j = 0; j < TJ; ++j) bf[j] = tr_join(tb0[j]);
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < TI; ++i) bsum[i] += frag_sum<PREC>(af[i]);
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = wg_mfma<PREC>(af[i], bf[j], acc[i][j]);
                tr_wait_all<TI, TJ, 0>(ta1, tb1);
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = tr_join(ta1[i]);
#pragma unroll
                for (int j = 0; j < TJ; ++j) bf[j] = tr_join(tb1[j]);
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < TI; ++i) bsum[i] += frag_sum<PREC>(af[i]);
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = wg_mfma<PREC>(af[i], bf[j], acc[i][j]);
            }
        }
        stage = stage + 1 == WG_STAGES ? 0 : stage + 1;
    }

    // partial block: [R][C] row-major then R bias sums
    if (!active) return;
    constexpr int R = A_NKS * 16, C = B_NKS * 16;
    float* out = a.partial + a.jobs.part_off[net * NGEMM + g] + (int64_t)split * gemm_out_floats(tab, g);
    const int col_l = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = 32 * (TJ * wc + j) + col_l;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (TI * wr + i) + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row < R) out[(int64_t)row * C + col] = acc[i][j][r];
            }
        }
        if (do_bias) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            const int row = 32 * (TI * wr + i) + (lane & 31);
            if (hh == 0 && row < R) out[(int64_t)R * C + row] = tot;
        }
    }
}

template <int PREC>
__global__ __launch_bounds__(512, 2) void mlp_wgrad_kernel(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[WG_LDS_BYTES];
#ifdef NERF_PRIO_UPPER
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef NERF_WG_STAMP
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    const int p = blockIdx.x;
    bool ran = false;
    for (int net = 0; net < a.nets.n_nets; ++net) {
        const bool view = a.nets.view[net];
        const ArchTab& tab = tab_of(a.nets, net);
        for (int g = 0; g < tab.ngemm; ++g) {
            const int v = net * NGEMM + g;
            const int split = p - a.jobs.piece0[v];
            if (split < 0 || split >= a.jobs.splits[v]) continue;          // workgroup-uniform
            int64_t wt0, wt1;
            piece_segment(a.nets, net, g, p, wt0, wt1);
            // the previous segment's last LDS reads (other waves) before this one's first copies
            if (ran) __syncthreads();
            ran = true;
            // shape dispatch: <A units, B units, wave grid, tiles per wave, wave tiles per step>
            if (g == 0) wgrad_body<PREC, 16, 4, 8, 1, 1, 2>(a, tab, net, g, lds, split, wt0, wt1);
            else if (g == 5) wgrad_body<PREC, 16, 20, 4, 2, 2, 5>(a, tab, net, g, lds, split, wt0, wt1);
            else if (g == 8 && view) wgrad_body<PREC, 9, 18, 2, 3, 3, 3>(a, tab, net, g, lds, split, wt0, wt1);
            else if (g == 8) wgrad_body<PREC, 1, 16, 1, 8, 1, 1, 2>(a, tab, net, g, lds, split, wt0, wt1);
            else if (g == 9) wgrad_body<PREC, 1, 8, 1, 4, 1, 1, 4>(a, tab, net, g, lds, split, wt0, wt1);
            else wgrad_body<PREC, 16, 16, 4, 2, 2, 4>(a, tab, net, g, lds, split, wt0, wt1);
        }
    }
#ifdef NERF_WG_STAMP
    if (a.stamps && threadIdx.x == 0) {
        a.stamps[2 * blockIdx.x] = t_start | ((unsigned long long)(__builtin_amdgcn_s_getreg(6164) & 15) << 56);   // XCC id on top
        a.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// 3. reduce + scatter to state_dict order
// ------------------------------------------------------------------------------------------------
// position (index inside a GEMM's row or column space) of hidden feature f
__device__ __forceinline__ int hidden_pos(int f) {
    const int r = f & 15;
    return (f & ~15) + 8 * ((r >> 2) & 1) + ((r & 3) | ((r >> 3) << 2));
}
// position of reference encoding feature f (L frequencies)
__device__ __forceinline__ int pe_pos(int f, int L) {
    int q, h;
    if (f < 3) { q = 3 * L + (f == 2); h = (f == 1); }
    else { const int gq = f - 3; const int l = gq / 6, w = gq % 6; h = w / 3; q = 3 * l + (w % 3); }
    return 16 * (q >> 3) + 8 * h + (q & 7);
}

struct ReduceArgs {
    NetSet nets;
    const float* partial;
    float* d_params[MAX_NETS];
    float* prod[MAX_NETS];            // [128][256]: P = dZ9^T H7 in feature order, for mlp_post_kernel
    const float* scale[MAX_NETS];     // nullable: divide by *scale
    const float* post[MAX_NETS];      // the network's W_f, b_f, W_r0[:, :256] table (packed weights)
    JobTable jobs;
};
constexpr int PROD_FLOATS = 128 * 256;

__device__ __forceinline__ float sum_partials(const ReduceArgs& a, const ArchTab& tab, int net, int g, int idx) {
    const int v = net * NGEMM + g;
    const float* p = a.partial + a.jobs.part_off[v] + idx;
    const int n = a.jobs.splits[v];
    const int stride = gemm_out_floats(tab, g);
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) s += p[(int64_t)i * stride];      // fixed order; loads of eight terms in flight
    return s;
}

// outputs of one network: its parameters and, for the full network, the product P
__host__ __device__ inline int reduce_outputs(int view) { return view ? N_PARAMS + PROD_FLOATS : N_PARAMS_NV; }

__global__ __launch_bounds__(256) void mlp_reduce_kernel(const ReduceArgs a) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    int net = 0;
    while (net < a.nets.n_nets && e >= reduce_outputs(a.nets.view[net])) { e -= reduce_outputs(a.nets.view[net]); ++net; }
    if (net >= a.nets.n_nets) return;
    const bool VIEW = a.nets.view[net];
    const ArchTab& tab = tab_of(a.nets, net);
    const float* scale = a.scale[net];
    if (VIEW && e >= N_PARAMS) {                        // the product P, handed to mlp_post_kernel
        const int k = (e - N_PARAMS) >> 8, i = (e - N_PARAMS) & 255;
        float v = sum_partials(a, tab, net, 8, hidden_pos(k) * (gemm_b_nks(tab, 8) * 16) + hidden_pos(i));
        if (scale) v = v / *scale;
        a.prod[net][e - N_PARAMS] = v;
        return;
    }
    int g, row, col = -1;      // col < 0: bias element
    if (!VIEW && e >= OFF_WO) {                         // output_linear: rows 0..3 of GEMM 8
        g = 8;
        if (e < OFF_BO) { row = (e - OFF_WO) >> 8; col = hidden_pos((e - OFF_WO) & 255); } else row = e - OFF_BO;
    } else if (e < OFF_W1) {                                   // pts_linears.0
        g = 0;
        if (e < OFF_B0) { row = hidden_pos(e / 63); col = pe_pos(e % 63, 10); } else row = hidden_pos(e - OFF_B0);
    } else if (e < OFF_W5) {                            // pts_linears.1..4
        const int l = (e - OFF_W1) / (256 * 256 + 256), r = (e - OFF_W1) % (256 * 256 + 256);
        g = 1 + l;
        if (r < 65536) { row = hidden_pos(r >> 8); col = hidden_pos(r & 255); } else row = hidden_pos(r - 65536);
    } else if (e < OFF_W6) {                            // pts_linears.5: columns [H4 | PE]
        g = 5;
        if (e < OFF_B5) {
            const int o = (e - OFF_W5) / 319, i = (e - OFF_W5) % 319;
            row = hidden_pos(o);
            col = i < 63 ? 256 + pe_pos(i, 10) : hidden_pos(i - 63);
        } else row = hidden_pos(e - OFF_B5);
    } else if (e < OFF_WS) {                            // pts_linears.6..7
        const int l = (e - OFF_W6) / (256 * 256 + 256), r = (e - OFF_W6) % (256 * 256 + 256);
        g = 6 + l;
        if (r < 65536) { row = hidden_pos(r >> 8); col = hidden_pos(r & 255); } else row = hidden_pos(r - 65536);
    } else if (e < OFF_WF) {                            // sigma_linear: row 128 of GEMM 8
        g = 8; row = 128;
        if (e < OFF_BS) col = hidden_pos(e - OFF_WS);
    } else if (e < OFF_WR0) {                           // feature_linear: mlp_post_kernel
        return;
    } else if (e < OFF_WR2) {                           // rgb_layers.0
        g = 8;
        if (e < OFF_BR0) {
            const int o = (e - OFF_WR0) / 283, i = (e - OFF_WR0) % 283;
            if (i < 256) return;                        // mlp_post_kernel
            row = hidden_pos(o); col = 256 + pe_pos(i - 256, 4);
        } else row = hidden_pos(e - OFF_BR0);
    } else {                                            // rgb_layers.2
        g = 9;
        if (e < OFF_BR2) { row = (e - OFF_WR2) / 128; col = hidden_pos((e - OFF_WR2) % 128); } else row = e - OFF_BR2;
    }
    const int R = gemm_a_nks(tab, g) * 16, C = gemm_b_nks(tab, g) * 16;
    const int idx = col >= 0 ? row * C + col : R * C + row;
    float v = sum_partials(a, tab, net, g, idx);
    if (scale) v = v / *scale;
    a.d_params[net][e] = v;
}

// The factored gradients (mlp_common.hpp, saved-tensor layout): with P = dZ9^T H7 [128][256] and
// s = sum over samples of dZ9 (= d rgb_layers.0.bias, already in d_params),
//   d feature_linear.weight[o][i] = sum_k W_r0[k][o] P[k][i]          d feature_linear.bias[o] = sum_k W_r0[k][o] s[k]
//   d rgb_layers.0.weight[k][o]   = sum_i P[k][i] W_f[o][i] + s[k] b_f[o]
// `post` holds the forward's (rounded) W_f, b_f and W_r0[:, :256].  98 304 + 256 outputs, 25 MFLOP.
// One thread per output; consecutive threads read consecutive addresses of one operand and the same
// address of the other.  The loops are unrolled so that sixteen terms' loads are in flight together
// (a thread's chain of 128 or 256 dependent load-then-fma steps is otherwise latency-bound).
constexpr int POST_OUTPUTS = 256 * 256 + 256 + 128 * 256;
__global__ __launch_bounds__(256) void mlp_post_kernel(const ReduceArgs a) {
    const int net = blockIdx.y;
    if (!a.nets.view[net]) return;           // the network without view directions has no factored gradients
    const float* __restrict__ prod = a.prod[net];
    const float* __restrict__ post = a.post[net];
    float* __restrict__ d_params = a.d_params[net];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const float* wft = post + POST_WF;       // [in][out]
    const float* bf = post + POST_BF;
    const float* wr0 = post + POST_WR0;      // [k][o]
    const float* s = d_params + OFF_BR0;
    if (e < 256 * 256) {                                 // feature_linear.weight[o][i]
        const int o = e >> 8, i = e & 255;
        float acc = 0.f;
#pragma unroll 16
        for (int k = 0; k < 128; ++k) acc = fmaf(wr0[k * 256 + o], prod[k * 256 + i], acc);
        d_params[OFF_WF + e] = acc;
    } else if (e < 256 * 256 + 256) {                    // feature_linear.bias[o]
        const int o = e - 256 * 256;
        float acc = 0.f;
#pragma unroll 16
        for (int k = 0; k < 128; ++k) acc = fmaf(wr0[k * 256 + o], s[k], acc);
        d_params[OFF_BF + o] = acc;
    } else if (e < POST_OUTPUTS) {                       // rgb_layers.0.weight[k][o], o < 256
        const int r = e - (256 * 256 + 256);
        const int k = r >> 8, o = r & 255;
        float acc = s[k] * bf[o];
#pragma unroll 16
        for (int i = 0; i < 256; ++i) acc = fmaf(prod[k * 256 + i], wft[i * 256 + o], acc);
        d_params[OFF_WR0 + k * 283 + o] = acc;
    }
}

}  // namespace mlp
}  // namespace nerf

using namespace nerf::mlp;

static inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

static int net_set(int n_nets, const nerf_wgrad_net_t* nets, NetSet* out, int* prec) {
    NERF_REQUIRE(nets, NERF_E_NULL);
    NERF_REQUIRE(n_nets >= 1 && n_nets <= MAX_NETS, NERF_E_SHAPE);
    NetSet s{};
    s.n_nets = n_nets;
    for (int n = 0; n < n_nets; ++n) {
        NERF_REQUIRE(nets[n].M >= 1, NERF_E_SHAPE);
        NERF_REQUIRE(nerf_prec_ok(nets[n].precision), NERF_E_DTYPE);
        const int p = nets[n].precision & ~NERF_ARCH_NOVIEW;
        if (n == 0) *prec = p;
        NERF_REQUIRE(p == *prec, NERF_E_DTYPE);            // one operand type per launch
        s.view[n] = !(nets[n].precision & NERF_ARCH_NOVIEW);
        s.n_wt[n] = (nets[n].M + TILE_SAMPLES - 1) / TILE_SAMPLES * 8;
    }
    *out = s;
    return NERF_OK;
}

extern "C" int64_t nerf_mlp_bwd_weights_multi_bytes(int n_nets, const nerf_wgrad_net_t* nets) {
    NetSet s; int prec = 0;
    if (net_set(n_nets, nets, &s, &prec) != NERF_OK) return -1;
    int64_t floats = 0;
    make_jobs(s, &floats);
    return round_up(floats * 4, 256) + (int64_t)n_nets * PROD_FLOATS * 4;     // partial blocks, then the products P
}

extern "C" int64_t nerf_mlp_partial_bytes(int64_t n_samples) {
    // enough for either architecture
    nerf_wgrad_net_t a{}, b{};
    a.M = b.M = n_samples > 0 ? n_samples : 1;
    a.precision = NERF_PREC_BF16; b.precision = NERF_PREC_BF16 | NERF_ARCH_NOVIEW;
    const int64_t x = nerf_mlp_bwd_weights_multi_bytes(1, &a), y = nerf_mlp_bwd_weights_multi_bytes(1, &b);
    return x > y ? x : y;
}

extern "C" int64_t nerf_mlp_bwd_workspace_bytes(int64_t n_samples) {
    const int64_t n_tiles = (n_samples + TILE_SAMPLES - 1) / TILE_SAMPLES;
    return round_up(n_tiles * 8 * (int64_t)DZ_UNITS * 1024, 256) + nerf_mlp_partial_bytes(n_samples);
}

static int check_bwd_common(int64_t M, int precision) {
    NERF_REQUIRE(M >= 0, NERF_E_SHAPE);
    NERF_REQUIRE(nerf_prec_ok(precision), NERF_E_DTYPE);
    return NERF_OK;
}

extern "C" int64_t nerf_mlp_dz_bytes(int64_t n_samples) {
    const int64_t n_tiles = (n_samples + TILE_SAMPLES - 1) / TILE_SAMPLES;
    return round_up(n_tiles * 8 * (int64_t)DZ_UNITS * 1024, 256);
}

extern "C" int nerf_mlp_bwd_data(int64_t M, const float* d_raw, const void* packed, int precision,
                                 const void* saved, const float* grad_scale, void* dz, void* stream) {
    int rc = check_bwd_common(M, precision);
    if (rc) return rc;
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(d_raw && packed && saved && dz, NERF_E_NULL);
    NERF_REQUIRE(nerf_aligned(d_raw, 16) && nerf_aligned(packed, 256) && nerf_aligned(saved, 256) &&
                     nerf_aligned(dz, 256), NERF_E_ALIGN);
    const bool view = !(precision & NERF_ARCH_NOVIEW);
    const int prec = precision & ~NERF_ARCH_NOVIEW;
    const int64_t n_tiles = (M + TILE_SAMPLES - 1) / TILE_SAMPLES;
    BwdArgs b{};
    b.M = M; b.n_tiles = n_tiles; b.d_raw = d_raw; b.packed = (const char*)packed; b.saved = (const char*)saved;
    b.dz = (char*)dz; b.scale = grad_scale;
    const dim3 gb((unsigned)(n_tiles < 256 ? n_tiles : 256)), tb(CHAIN_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (prec == NERF_PREC_BF16) {
        if (view) hipLaunchKernelGGL((mlp_dgrad_kernel<NERF_PREC_BF16, true>), gb, tb, 0, st, b);
        else hipLaunchKernelGGL((mlp_dgrad_kernel<NERF_PREC_BF16, false>), gb, tb, 0, st, b);
    } else {
        if (view) hipLaunchKernelGGL((mlp_dgrad_kernel<NERF_PREC_FP16, true>), gb, tb, 0, st, b);
        else hipLaunchKernelGGL((mlp_dgrad_kernel<NERF_PREC_FP16, false>), gb, tb, 0, st, b);
    }
    return nerf_launch_status();
}

extern "C" int nerf_mlp_bwd_weights_multi(int n_nets, const nerf_wgrad_net_t* nets, void* partial, int64_t partial_bytes,
                                          void* stream) {
    NetSet s; int prec = 0;
    int rc = net_set(n_nets, nets, &s, &prec);
    if (rc) return rc;
    NERF_REQUIRE(partial, NERF_E_NULL);
    NERF_REQUIRE(nerf_aligned(partial, 256), NERF_E_ALIGN);
    for (int n = 0; n < n_nets; ++n) {
        NERF_REQUIRE(nets[n].packed && nets[n].saved && nets[n].dz && nets[n].d_params, NERF_E_NULL);
        NERF_REQUIRE(nerf_aligned(nets[n].packed, 256) && nerf_aligned(nets[n].saved, 256) && nerf_aligned(nets[n].dz, 256),
                     NERF_E_ALIGN);
    }
    int64_t floats = 0;
    const JobTable jobs = make_jobs(s, &floats);
    const int64_t prod_off = round_up(floats * 4, 256);
    NERF_REQUIRE(partial_bytes >= prod_off + (int64_t)n_nets * PROD_FLOATS * 4, NERF_E_WORKSPACE);
    hipStream_t st = (hipStream_t)stream;

    WgradArgs w{};
    w.nets = s; w.partial = (float*)partial; w.jobs = jobs;
    for (int n = 0; n < n_nets; ++n) { w.saved[n] = (const char*)nets[n].saved; w.dz[n] = (const char*)nets[n].dz; }
#ifdef NERF_WG_STAMP
    {   // stamps go behind everything the library uses of the partial buffer, when the caller left room
        const int64_t used = prod_off + (int64_t)n_nets * PROD_FLOATS * 4;
        w.stamps = partial_bytes >= used + 16 * 1024 ? reinterpret_cast<unsigned long long*>((char*)partial + used) : nullptr;
    }
#endif
    const dim3 gw((unsigned)WGRAD_TARGET_JOBS), tw(512);
    if (prec == NERF_PREC_BF16) hipLaunchKernelGGL(mlp_wgrad_kernel<NERF_PREC_BF16>, gw, tw, 0, st, w);
    else hipLaunchKernelGGL(mlp_wgrad_kernel<NERF_PREC_FP16>, gw, tw, 0, st, w);
    rc = nerf_launch_status();
    if (rc) return rc;

    ReduceArgs r{};
    r.nets = s; r.partial = (const float*)partial; r.jobs = jobs;
    int64_t outputs = 0;
    bool any_view = false;
    for (int n = 0; n < n_nets; ++n) {
        r.d_params[n] = nets[n].d_params; r.scale[n] = nets[n].grad_scale;
        r.prod[n] = reinterpret_cast<float*>((char*)partial + prod_off) + (int64_t)n * PROD_FLOATS;
        r.post[n] = reinterpret_cast<const float*>((const char*)nets[n].packed + PACK_POST_OFF);
        outputs += reduce_outputs(s.view[n]);
        any_view |= s.view[n] != 0;
    }
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)((outputs + 255) / 256)), dim3(256), 0, st, r);
    rc = nerf_launch_status();
    if (rc || !any_view) return rc;
    hipLaunchKernelGGL(mlp_post_kernel, dim3((POST_OUTPUTS + 255) / 256, n_nets), dim3(256), 0, st, r);
    return nerf_launch_status();
}

extern "C" int nerf_mlp_bwd_weights(int64_t M, const void* packed, const void* saved, const void* dz, int precision,
                                    const float* grad_scale, void* partial, int64_t partial_bytes, float* d_params,
                                    void* stream) {
    int rc = check_bwd_common(M, precision);
    if (rc) return rc;
    NERF_REQUIRE(d_params, NERF_E_NULL);
    if (M == 0) {
        const bool view = !(precision & NERF_ARCH_NOVIEW);
        hipError_t e = hipMemsetAsync(d_params, 0, sizeof(float) * (view ? N_PARAMS : N_PARAMS_NV), (hipStream_t)stream);
        return e == hipSuccess ? NERF_OK : (int)e;
    }
    nerf_wgrad_net_t net{};
    net.M = M; net.packed = packed; net.saved = saved; net.dz = dz; net.precision = precision;
    net.grad_scale = grad_scale; net.d_params = d_params;
    return nerf_mlp_bwd_weights_multi(1, &net, partial, partial_bytes, stream);
}

extern "C" int nerf_mlp_bwd(int64_t M, const float* d_raw, const void* packed, int precision, const void* saved,
                            const float* grad_scale, void* workspace, int64_t workspace_bytes, float* d_params,
                            void* stream) {
    int rc = check_bwd_common(M, precision);
    if (rc) return rc;
    NERF_REQUIRE(d_params, NERF_E_NULL);
    if (M > 0) {
        NERF_REQUIRE(workspace, NERF_E_NULL);
        NERF_REQUIRE(nerf_aligned(workspace, 256), NERF_E_ALIGN);
        NERF_REQUIRE(workspace_bytes >= nerf_mlp_bwd_workspace_bytes(M), NERF_E_WORKSPACE);
    }
    const int64_t dzb = nerf_mlp_dz_bytes(M);
    rc = nerf_mlp_bwd_data(M, d_raw, packed, precision, saved, grad_scale, workspace, stream);
    if (rc) return rc;
    return nerf_mlp_bwd_weights(M, packed, saved, workspace, precision, grad_scale, (char*)workspace + dzb,
                                workspace_bytes - dzb, d_params, stream);
}
