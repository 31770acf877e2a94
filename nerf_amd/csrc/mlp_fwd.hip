// Training forward of the fused positional encoding + 8x256 MLP (src/nerf_project/models/nerf.py:68-76,
// :173-209 and the point / view-direction set-up of render_rays, :409-421) on bf16/fp16
// v_mfma_f32_32x32x16: the variant that keeps the activations for the backward pass, and the forward on
// pre-encoded inputs (the literal NeRF.forward).  The no_grad ray forward is mlp_fwd16.hip.
//
// 1184 MFMAs per 32 samples (600 064 padded MAC/sample against 593 408 algorithmic).  Every 16-bit
// fragment the chain produces is also stored as a [32 samples][16 positions] unit -- the format the
// weight-gradient pass reads -- together with one sign bit per activation for the backward-data chain:
// 4.7 KiB per sample, which is what bounds this kernel.  Layout and schedule: mlp_common.hpp, mlp_chain.hpp.
#include <mutex>
#include <type_traits>
#include "mlp_chain.hpp"
#include "mlp_waits.hpp"

namespace nerf {
namespace mlp {

// chunk entries a wave passes while it consumes stream blocks [b0, b1): one per block that starts a chunk
constexpr int chunk_entries(int b0, int b1) {
    int n = 0;
    for (int b = b0; b < b1; ++b) n += (b % CHUNK_BLOCKS == 0);
    return n;
}
// vector-memory operations of a saving layer's tile epilogues: two unit stores per tile, one mask store per pair
constexpr int epilogue_stores(int tiles) { return 2 * tiles + tiles / 2; }
// What a wave issues between the next pass's input loads and their wait.  With view directions: L9 and L10
// (the final store of raw may be skipped by a wave past the end and is not counted); without: L7 and the
// output layer.
static_assert(NERF_INPUTS_WAIT_VIEW == chunk_entries(fwd_block0(9), fwd_block0(11)) * DMA_PER_CHUNK +
                                           epilogue_stores(FWD_TILES[9]), "mlp_waits.hpp: view network");
static_assert(NERF_INPUTS_WAIT_NOVIEW == chunk_entries(fwd_block0(7), NV_FWD_BLOCKS_REAL) * DMA_PER_CHUNK +
                                             epilogue_stores(FWD_TILES[7]), "mlp_waits.hpp: network without view directions");
static_assert(NERF_INPUTS_WAIT_VIEW <= 63 && NERF_INPUTS_WAIT_NOVIEW <= 63, "vmcnt is a 6-bit counter");

struct FwdArgs {
    int64_t M;            // samples
    int64_t n_tiles;      // ceil(M / 256)
    int S;                // samples per ray (fused input mode)
    const float* rays_o;  // [n_rays,3]
    const float* rays_d;  // [n_rays,3]
    const float* z;       // [M]
    const char* pe_units; // encoded input mode: the point / view encodings as 16-bit fragment units
    const char* vpe_units;//   (written by encode_units_kernel), PE_KS resp. VPE_KS KiB per wave tile
    const char* packed;
    float* raw;           // [M,4]
    char* saved;          // nullable
};

// sin(2 pi r) on the hardware unit (input in revolutions)
__device__ __forceinline__ float sin_rev(float r) { return __builtin_amdgcn_sinf(r); }

// Fragments of the positional encoding of (px,py,pz) with L frequencies 2^l (no pi), NKS k-steps.
// Lane half h = 0 evaluates sines, h = 1 cosines (a quarter-revolution phase).  The argument is
// reduced exactly: t = p/(2 pi) as a two-term product, 2^l * t_hi is exact, fract() is exact.
template <int PREC, int L, int NKS>
__device__ __forceinline__ void encode(float px, float py, float pz, int h, typename Prec<PREC>::frag* out) {
    typedef typename Prec<PREC>::elem T;
    constexpr float C_HI = 0.15915494f;                 // fl32(1 / (2 pi))
    constexpr float C_LO = 6.4206382e-09f; // 1/(2 pi) - C_HI
    const float p[3] = {px, py, pz};
    float thi[3], tlo[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        thi[c] = p[c] * C_HI;
        tlo[c] = fmaf(p[c], C_HI, -thi[c]) + p[c] * C_LO;
    }
    const float phase = h ? 0.25f : 0.0f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 8 * ks + j;
            float v;
            if (q < 3 * L) {
                const int l = q / 3, c = q % 3;
                const float sc = (float)(1 << l);
                const float r = __builtin_amdgcn_fractf(thi[c] * sc) + fmaf(tlo[c], sc, phase);
                v = sin_rev(r);
            } else if (q == 3 * L) {
                v = h ? py : px;
            } else if (q == 3 * L + 1) {
                v = h ? 0.0f : pz;
            } else {
                v = 0.0f;
            }
            out[ks][j] = (T)v;
        }
    }
}

// Encoded input mode (the literal NeRF.forward(x_pts_encoded, x_views_encoded), nerf.py:155): the fp32
// rows [M,63] and [M,27] are first rewritten as fragment units -- the layout of the saved PE / VPE
// tensors -- by this bandwidth-bound kernel; the chain kernel then starts a pass with six 16-byte loads
// per lane instead of gathering 90 strided floats per sample into registers.  One thread = one lane's
// fragment of one unit.
template <int PREC>
__global__ __launch_bounds__(256) void encode_units_kernel(int64_t M, int64_t n_wt, const float* __restrict__ x_pts,
                                                          const float* __restrict__ x_views, char* __restrict__ pe_units,
                                                          char* __restrict__ vpe_units) {
    typedef typename Prec<PREC>::elem T;
    typedef typename Prec<PREC>::frag frag;
    constexpr int UNITS = PE_KS + VPE_KS;
    const int64_t total = n_wt * UNITS * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int64_t u = i >> 6;
        const int64_t wt = u / UNITS;
        const int k = (int)(u - wt * UNITS);
        const bool view = k >= PE_KS;
        const int ks = view ? k - PE_KS : k;
        const int c = lane & 31, h = lane >> 5;
        const int64_t m = wt * WAVE_SAMPLES + c;
        frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int feat = pe_combo_feature(8 * ks + j, h, view ? 4 : 10);
            float v = 0.0f;
            if (m < M && feat >= 0) v = view ? (x_views ? x_views[m * VPE_FEATS + feat] : 0.0f) : x_pts[m * PE_FEATS + feat];
            f[j] = (T)v;
        }
        char* dst = view ? vpe_units + (wt * VPE_KS + ks) * 1024 : pe_units + (wt * PE_KS + ks) * 1024;
        *reinterpret_cast<frag*>(dst + c * 32 + h * 16) = f;
    }
}

// Stores of the kept activations.  The unit address is wave-uniform (the wave index comes from readfirstlane), so
// the store takes the scalar-base form: base in SGPRs, the lane's byte offset in one VGPR.  A per-lane 64-bit
// pointer per saved tensor would live through the whole pass, and a spilled one is reloaded with vmcnt(0), which
// waits for every store of the pass.
template <int PREC>
__device__ __forceinline__ void save_frag(char* saved, int ks0, int nks, int64_t n_wt, int64_t wt, int ks,
                                          int lane_off, const typename Prec<PREC>::frag& f) {
#ifdef NERF_ABL_STORE_LOCAL      // timing experiment only: every store of a wave goes to one L2-resident KiB
    *reinterpret_cast<typename Prec<PREC>::frag*>(saved + (size_t)(blockIdx.x * 8 + (threadIdx.x >> 6)) * 1024 + lane_off) = f;
    return;
#endif
    char* unit = saved + unit_offset(ks0, nks, n_wt, wt, ks);        // wave-uniform
#ifdef NERF_ABL_PLAIN_STORE      // timing experiment: default cache policy instead of non-temporal
    *reinterpret_cast<typename Prec<PREC>::frag*>(unit + (uint32_t)lane_off) = f;
    return;
#endif
    unit_store(unit, (uint32_t)lane_off, f);
}
// one dword per lane at byte offset lane_off4 of the wave-uniform address `base`.  A plain store on purpose: the
// sign masks of a step (75 MB at 1024 rays) then stay in the 256 MB last-level cache until the backward-data
// pass reads them (stored non-temporally, that pass ran 10 % slower).
__device__ __forceinline__ void save_dword(char* base, int lane_off4, uint32_t v) {
    *reinterpret_cast<uint32_t*>(base + (uint32_t)lane_off4) = v;
}

// signs of one finished output tile (already ReLU'd and rounded: 16-bit values with a clear top bit),
// in the bits 8-15 / 24-31 pattern described at mask_bit().  x + 0x7fff carries into bit 15 exactly
// when the 16-bit value x is non-zero.
template <int PREC>
__device__ __forceinline__ uint32_t sign_bits(const typename Prec<PREC>::frag& f0, const typename Prec<PREC>::frag& f1) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 a = __builtin_bit_cast(u32x4, f0), b = __builtin_bit_cast(u32x4, f1);
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) m = (m >> 1) | ((a[q] + 0x7fff7fffu) & 0x80008000u);
#pragma unroll
    for (int q = 0; q < 4; ++q) m = (m >> 1) | ((b[q] + 0x7fff7fffu) & 0x80008000u);
    return m;
}

// VIEW = false: the network without view directions (output_linear follows L7; mlp_common.hpp)
template <int PREC, bool SAVE, bool ENCODED, bool VIEW>
__global__ __launch_bounds__(CHAIN_THREADS, 2) void mlp_fwd_kernel(const FwdArgs a) {
    typedef typename Prec<PREC>::frag frag;
    __shared__ __attribute__((aligned(16))) char lds[CHAIN_LDS_BYTES];
    float* bias_lds = reinterpret_cast<float*>(lds + LDS_BIAS_OFF);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef NERF_PRIO_UPPER        // experiment: static priority for the later-dispatched half of the workgroup
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    const int c = lane & 31, h = lane >> 5;
#ifdef NERF_ABL_LANE_LINEAR      // timing experiment only (wrong layout): lanes store in address order
    const int lane_off = lane * 16;
#else
    const int lane_off = c * 32 + h * 16;        // byte offset of this lane's fragment in a saved unit
#endif

    load_bias_table<VIEW>(bias_lds, a.packed);       // (and the check for non-finite parameters)

    Ring ring;
    ring_init(ring, lds, a.packed + PACK_FWD_OFF, VIEW ? FWD_CHUNKS : NV_FWD_CHUNKS, lane, wave);
    const int64_t n_wt = a.n_tiles * 8;
    char* const mask_base = SAVE ? a.saved + (size_t)n_wt * SAVED_UNITS * 1024 : nullptr;

    // Ray inputs of a pass (fused mode): z of this lane's sample, origin and direction of its ray.  The loads
    // of the NEXT pass are issued two layers before the current pass ends and turned into the point encoding
    // and the unit view direction after its last layer: at a pass boundary nothing then waits for memory (a
    // load issued there would also wait for the pass's last stores, which the memory counter orders before it).
    // The loads are inline asm: the compiler would wait for a load it knows about with vmcnt(0) at the first use
    // after the chunk-entry asm, draining the weight ring and the pass's stores.  inputs_landed<N>() is the wait:
    // N must not exceed the vector-memory operations this wave issues between the loads and the wait.
    struct RayIn { float z, ox, oy, oz, dx, dy, dz; };
    auto load_inputs = [&](int64_t tile2) {
        RayIn r;
        int64_t m2 = (tile2 * 8 + wave) * WAVE_SAMPLES + c;
        if (m2 >= a.M) m2 = a.M - 1;                      // lanes past the end load a valid sample
        const int64_t ray = m2 / a.S;
        const float* zp = a.z + m2;
        const float* o = a.rays_o + 3 * ray;
        const float* d = a.rays_d + 3 * ray;
#ifdef NERF_ABL_NO_INPUTS        // (timing experiment only: no input loads, constant encodings)
        r = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#else
        asm volatile("global_load_dword %0, %1, off" : "=v"(r.z) : "v"(zp));
        asm volatile("global_load_dword %0, %1, off" : "=v"(r.ox) : "v"(o));
        asm volatile("global_load_dword %0, %1, off offset:4" : "=v"(r.oy) : "v"(o));
        asm volatile("global_load_dword %0, %1, off offset:8" : "=v"(r.oz) : "v"(o));
        asm volatile("global_load_dword %0, %1, off" : "=v"(r.dx) : "v"(d));
        asm volatile("global_load_dword %0, %1, off offset:4" : "=v"(r.dy) : "v"(d));
        asm volatile("global_load_dword %0, %1, off offset:8" : "=v"(r.dz) : "v"(d));
#endif
        return r;
    };
#define NERF_INPUTS_LANDED(R, N)                                                                               \
    asm volatile("s_waitcnt vmcnt(%7)" : "+v"(R.z), "+v"(R.ox), "+v"(R.oy), "+v"(R.oz), "+v"(R.dx), "+v"(R.dy), "+v"(R.dz) : "n"(N))
    // raw[m] of this lane's sample, from lane half 0.  The lane id passes through an empty asm here, so that the
    // address is formed at the store and no per-lane pointer lives (or is spilled) through the pass.
    auto raw_store = [&](int64_t wt_now, float4 v) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int64_t mm = wt_now * WAVE_SAMPLES + (l & 31);
        if (mm < a.M && l < 32) {
            char* row = reinterpret_cast<char*>(a.raw) + wt_now * (WAVE_SAMPLES * 16);     // wave-uniform
            *reinterpret_cast<float4*>(row + (uint32_t)(l * 16)) = v;
        }
    };
    frag pe[PE_KS];
    float vx = 0.f, vy = 0.f, vz = 0.f;   // unit view direction (fused mode)
    auto use_inputs = [&](const RayIn& r, bool ok) {
        float px = 0.f, py = 0.f, pz = 0.f;
        vx = vy = vz = 0.f;
        if (ok) {
            px = __fadd_rn(r.ox, __fmul_rn(r.dx, r.z));        // nerf.py:409
            py = __fadd_rn(r.oy, __fmul_rn(r.dy, r.z));
            pz = __fadd_rn(r.oz, __fmul_rn(r.dz, r.z));
            const float n = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(r.dx, r.dx), __fmul_rn(r.dy, r.dy)), __fmul_rn(r.dz, r.dz)));
            vx = r.dx / n; vy = r.dy / n; vz = r.dz / n;       // nerf.py:417
        }
        encode<PREC, 10, PE_KS>(px, py, pz, h, pe);
    };
    if (!ENCODED && (int64_t)blockIdx.x < a.n_tiles) {
        RayIn r = load_inputs(blockIdx.x);
        NERF_INPUTS_LANDED(r, 0);                         // first pass only: nothing to overlap with yet
        use_inputs(r, ((int64_t)blockIdx.x * 8 + wave) * WAVE_SAMPLES + c < a.M);
    }

    for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t wt = tile * 8 + wave;
        const bool has_next = tile + gridDim.x < a.n_tiles;
        RayIn next_in = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // Operations between the next pass's loads and their use: without view directions L7 (4 chunk entries of 4
        // copies, 16 unit stores, 4 mask stores), else L9 (3 entries, 8 unit stores, 2 mask stores); the final
        // store of raw may be skipped by a wave past the end and is not counted.
        constexpr int YOUNGER_THAN_INPUTS = VIEW ? NERF_INPUTS_WAIT_VIEW : NERF_INPUTS_WAIT_NOVIEW;
        static_assert(fwd_block0(9) % CHUNK_BLOCKS == 0 && fwd_block0(7) % CHUNK_BLOCKS == 16, "chunk entries inside L9 / L7");
        static_assert(ENCODED || SAVE, "the count above includes the unit and mask stores (fused inference is mlp_fwd16.hip's)");

        if (ENCODED) {
#pragma unroll
            for (int ks = 0; ks < PE_KS; ++ks)
                pe[ks] = *reinterpret_cast<const frag*>(a.pe_units + (wt * PE_KS + ks) * 1024 + lane_off);
        }
        if (SAVE && !ENCODED) {          // in encoded mode the units already are the saved tensors
#pragma unroll
            for (int ks = 0; ks < PE_KS; ++ks)
                save_frag<PREC>(a.saved, saved_ks0(SV_PE), PE_KS, n_wt, wt, ks, lane_off, pe[ks]);
        }

        constexpr int SV = SAVE ? 2 : 0;     // stores per tile epilogue
        frag ha[H_KS], hb[H_KS];
        uint32_t even_bits = 0;               // sign bits of the last even tile, until its odd partner ends
        char* const mask_wt = SAVE ? mask_base + (size_t)wt * MASK_TENSORS * 1024 : nullptr;       // wave-uniform
        auto save_signs = [&](int tensor, int mt, const frag& f0, const frag& f1) {
            const uint32_t bits = sign_bits<PREC>(f0, f1);
            if (mt & 1) save_dword(mask_wt + tensor * 1024 + (mt >> 1) * 256, lane * 4, (even_bits >> 8) | bits);
            else even_bits = bits;
        };
        auto hidden = [&](auto LAYER, frag* out) {
            return [&, out](int mt, const f32x16& acc) {
                constexpr int layer = decltype(LAYER)::value;
                tile_to_frags<PREC, true>(acc, out[2 * mt], out[2 * mt + 1]);
                if (SAVE) {
                    save_frag<PREC>(a.saved, saved_ks0(SV_H0 + layer), H_KS, n_wt, wt, 2 * mt, lane_off, out[2 * mt]);
                    save_frag<PREC>(a.saved, saved_ks0(SV_H0 + layer), H_KS, n_wt, wt, 2 * mt + 1, lane_off,
                                    out[2 * mt + 1]);
                    save_signs(layer, mt, out[2 * mt], out[2 * mt + 1]);
                }
            };
        };
#define NERF_BIAS(LAYER) [&](int mt) { return bias_tile(bias_lds, LAYER, mt, h); }
#define NERF_IC(N) std::integral_constant<int, N>()
        ring_enter<0>(ring, lds);            // chunk 0 of the pass: L0 starts behind the stream's front pad
        run_layer<PREC, fwd_block0(0), 8, PE_KS, 0, SV>(ring, lds, pe, pe, NERF_BIAS(0), hidden(NERF_IC(0), ha));
        run_layer<PREC, fwd_block0(1), 8, H_KS, 0, SV>(ring, lds, ha, ha, NERF_BIAS(1), hidden(NERF_IC(1), hb));
        run_layer<PREC, fwd_block0(2), 8, H_KS, 0, SV>(ring, lds, hb, hb, NERF_BIAS(2), hidden(NERF_IC(2), ha));
        run_layer<PREC, fwd_block0(3), 8, H_KS, 0, SV>(ring, lds, ha, ha, NERF_BIAS(3), hidden(NERF_IC(3), hb));
        run_layer<PREC, fwd_block0(4), 8, H_KS, 0, SV>(ring, lds, hb, hb, NERF_BIAS(4), hidden(NERF_IC(4), ha));
        run_layer<PREC, fwd_block0(5), 8, PE_KS, H_KS, SV>(ring, lds, pe, ha, NERF_BIAS(5), hidden(NERF_IC(5), hb));
        run_layer<PREC, fwd_block0(6), 8, H_KS, 0, SV>(ring, lds, hb, hb, NERF_BIAS(6), hidden(NERF_IC(6), ha));
        if (!VIEW && !ENCODED && has_next) next_in = load_inputs(tile + gridDim.x);
        run_layer<PREC, fwd_block0(7), 8, H_KS, 0, SV>(ring, lds, ha, ha, NERF_BIAS(7), hidden(NERF_IC(7), hb));

        if (!VIEW) {
            // L8 of the network without view directions: output_linear -> rows 0..3 = (rgb_raw, sigma_raw)
            run_layer<PREC, fwd_block0(8), 1, H_KS, 0>(ring, lds, hb, hb, NERF_BIAS(8), [&](int, const f32x16& acc) {
                raw_store(wt, make_float4(acc[0], acc[1], acc[2], acc[3]));
            });
            if (!ENCODED) {      // unconditional, so that the encoding registers are free from L6 to here
                NERF_INPUTS_LANDED(next_in, YOUNGER_THAN_INPUTS);
                use_inputs(next_in, has_next && ((tile + gridDim.x) * 8 + wave) * WAVE_SAMPLES + c < a.M);
            }
            continue;
        }
        // L8: feature_linear (no activation) -> ha, sigma_linear -> row 0 of tile 8
        float sigma = 0.f;
        run_layer<PREC, fwd_block0(8), 9, H_KS, 0, 0>(ring, lds, hb, hb, NERF_BIAS(8), [&](int mt, const f32x16& acc) {
            if (mt < 8) {
                tile_to_frags<PREC, false>(acc, ha[2 * mt], ha[2 * mt + 1]);     // F is not kept (mlp_common.hpp)
            } else {
                sigma = acc[0];
            }
        });

        // L9: rgb_layers.0 on [feature, view encoding] -> relu -> G (first 8 k-steps of hb)
        frag vpe[VPE_KS];
        if (ENCODED) {
#pragma unroll
            for (int ks = 0; ks < VPE_KS; ++ks)
                vpe[ks] = *reinterpret_cast<const frag*>(a.vpe_units + (wt * VPE_KS + ks) * 1024 + lane_off);
        } else {
            encode<PREC, 4, VPE_KS>(vx, vy, vz, h, vpe);
        }
        if (SAVE && !ENCODED) {
#pragma unroll
            for (int ks = 0; ks < VPE_KS; ++ks)
                save_frag<PREC>(a.saved, saved_ks0(SV_VPE), VPE_KS, n_wt, wt, ks, lane_off, vpe[ks]);
        }
        if (!ENCODED && has_next) next_in = load_inputs(tile + gridDim.x);
        run_layer<PREC, fwd_block0(9), 4, H_KS, VPE_KS, SV>(ring, lds, ha, vpe, NERF_BIAS(9), [&](int mt, const f32x16& acc) {
            tile_to_frags<PREC, true>(acc, hb[2 * mt], hb[2 * mt + 1]);
            if (SAVE) {
                save_frag<PREC>(a.saved, saved_ks0(SV_G), G_KS, n_wt, wt, 2 * mt, lane_off, hb[2 * mt]);
                save_frag<PREC>(a.saved, saved_ks0(SV_G), G_KS, n_wt, wt, 2 * mt + 1, lane_off, hb[2 * mt + 1]);
                save_signs(MASK_G, mt, hb[2 * mt], hb[2 * mt + 1]);
            }
        });

        // L10: rgb_layers.2 -> rows 0..2; output (rgb_raw, sigma_raw) per sample from lane half 0
        run_layer<PREC, fwd_block0(10), 1, G_KS, 0>(ring, lds, hb, hb, NERF_BIAS(10), [&](int, const f32x16& acc) {
            raw_store(wt, make_float4(acc[0], acc[1], acc[2], sigma));
        });
        if (!ENCODED) {          // unconditional, so that the encoding registers are free from L6 to here
            NERF_INPUTS_LANDED(next_in, YOUNGER_THAN_INPUTS);
            use_inputs(next_in, has_next && ((tile + gridDim.x) * 8 + wave) * WAVE_SAMPLES + c < a.M);
        }
#undef NERF_BIAS
#undef NERF_IC
    }
    // LDS-DMA prefetched for a pass that does not exist must land before the wave ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace mlp
}  // namespace nerf

using namespace nerf::mlp;

// This file is compiled three times (NERF_TU, see the Makefile): the kernels of one operand type each
// (NERF_TU = 1: bf16, 2: fp16 -- twelve fully unrolled instantiations take minutes, the two halves build in
// parallel), and the host entry points (NERF_TU = 0).
#ifndef NERF_TU
#error "compile with -DNERF_TU=0 (entry points), 1 (bf16 kernels) or 2 (fp16 kernels)"
#endif
namespace nerf {
namespace mlp {
template <int PREC>
void launch_fwd_prec(const FwdArgs& a, bool save, bool encoded, bool view, dim3 g, dim3 b, hipStream_t st)
#if NERF_TU == 0
    ;
#else
{
#define NERF_LAUNCH(SV, EN) do { if (view) hipLaunchKernelGGL((mlp_fwd_kernel<PREC, SV, EN, true>), g, b, 0, st, a); \
                                 else hipLaunchKernelGGL((mlp_fwd_kernel<PREC, SV, EN, false>), g, b, 0, st, a); } while (0)
    if (save) { if (encoded) NERF_LAUNCH(true, true); else NERF_LAUNCH(true, false); }
    else { NERF_LAUNCH(false, true); }      /* fused inference is mlp_fwd16.hip's */
#undef NERF_LAUNCH
}
template void launch_fwd_prec<NERF_TU == 1 ? NERF_PREC_BF16 : NERF_PREC_FP16>(const FwdArgs&, bool, bool, bool, dim3, dim3, hipStream_t);
#endif
}  // namespace mlp
}  // namespace nerf

#if NERF_TU == 0
extern "C" int64_t nerf_mlp_saved_bytes(int64_t n_samples) {
    const int64_t n_tiles = (n_samples + TILE_SAMPLES - 1) / TILE_SAMPLES;
    return n_tiles * 8 * saved_bytes_per_wave_tile();
}

static int launch_fwd(const FwdArgs& a, int precision, bool encoded, void* stream) {
    const bool view = !(precision & NERF_ARCH_NOVIEW);
    precision &= ~NERF_ARCH_NOVIEW;
    NERF_REQUIRE(precision == NERF_PREC_BF16 || precision == NERF_PREC_FP16, NERF_E_DTYPE);
    NERF_REQUIRE(encoded || a.saved != nullptr, NERF_E_NULL);
    if (a.M == 0) return NERF_OK;
    int64_t grid = a.n_tiles < 256 ? a.n_tiles : 256;   // one persistent workgroup per CU
    const dim3 g((unsigned)grid), b(CHAIN_THREADS);
    hipStream_t st = (hipStream_t)stream;
    const bool save = a.saved != nullptr;
    if (precision == NERF_PREC_BF16) launch_fwd_prec<NERF_PREC_BF16>(a, save, encoded, view, g, b, st);
    else launch_fwd_prec<NERF_PREC_FP16>(a, save, encoded, view, g, b, st);
    return nerf_launch_status();
}

// inference (nothing saved) runs on the 16x16x32 kernel of mlp_fwd16.hip
int nerf_mlp_fwd16_launch(int64_t n_rays, int S, const float* rays_o, const float* rays_d, const float* z,
                          const void* packed, int precision, float* raw, void* stream);

extern "C" int nerf_mlp_fwd(int64_t n_rays, int S, const float* rays_o, const float* rays_d, const float* z,
                            const void* packed, int precision, float* raw, void* saved, void* stream) {
    NERF_REQUIRE(n_rays >= 0 && S >= 1, NERF_E_SHAPE);
    NERF_REQUIRE(nerf_prec_ok(precision), NERF_E_DTYPE);
    if (n_rays == 0) return NERF_OK;          // an empty call: its tensors have no address
    NERF_REQUIRE(rays_o && rays_d && z && packed && raw, NERF_E_NULL);
    NERF_REQUIRE(nerf_aligned(raw, 16) && nerf_aligned(packed, 256) && nerf_aligned(saved, 256), NERF_E_ALIGN);
    if (saved == nullptr) return nerf_mlp_fwd16_launch(n_rays, S, rays_o, rays_d, z, packed, precision, raw, stream);
    FwdArgs a{};
    a.M = n_rays * S; a.n_tiles = (a.M + TILE_SAMPLES - 1) / TILE_SAMPLES; a.S = S;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.packed = (const char*)packed; a.raw = raw;
    a.saved = (char*)saved;
    return launch_fwd(a, precision, false, stream);
}

extern "C" int64_t nerf_mlp_encoded_workspace_bytes(int64_t M) {
    const int64_t n_tiles = (M + TILE_SAMPLES - 1) / TILE_SAMPLES;
    return n_tiles * 8 * (int64_t)(PE_KS + VPE_KS) * 1024;
}

static int fwd_encoded_impl(int64_t M, const float* x_pts, const float* x_views, const void* packed, int precision,
                            float* raw, void* saved, void* workspace, void* stream) {
    FwdArgs a{};
    a.M = M; a.n_tiles = (M + TILE_SAMPLES - 1) / TILE_SAMPLES; a.S = 1;
    a.packed = (const char*)packed; a.raw = raw; a.saved = (char*)saved;
    const int64_t n_wt = a.n_tiles * 8;
    // the encodings' units: the PE / VPE tensors of `saved` when activations are kept, else the workspace
    char* pe_units = saved ? (char*)saved + (int64_t)saved_ks0(SV_PE) * n_wt * 1024 : (char*)workspace;
    char* vpe_units = saved ? (char*)saved + (int64_t)saved_ks0(SV_VPE) * n_wt * 1024
                            : (char*)workspace + (int64_t)PE_KS * n_wt * 1024;
    a.pe_units = pe_units; a.vpe_units = vpe_units;
    const int64_t threads = n_wt * (PE_KS + VPE_KS) * 64;
    int64_t blocks = (threads + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if ((precision & ~NERF_ARCH_NOVIEW) == NERF_PREC_BF16)
        hipLaunchKernelGGL(encode_units_kernel<NERF_PREC_BF16>, dim3((unsigned)blocks), dim3(256), 0,
                           (hipStream_t)stream, M, n_wt, x_pts, x_views, pe_units, vpe_units);
    else
        hipLaunchKernelGGL(encode_units_kernel<NERF_PREC_FP16>, dim3((unsigned)blocks), dim3(256), 0,
                           (hipStream_t)stream, M, n_wt, x_pts, x_views, pe_units, vpe_units);
    const int rc = nerf_launch_status();
    if (rc) return rc;
    return launch_fwd(a, precision, true, stream);
}

extern "C" int nerf_mlp_fwd_encoded(int64_t M, const float* x_pts, const float* x_views, const void* packed,
                                    int precision, float* raw, void* saved, void* workspace, void* stream) {
    NERF_REQUIRE(M >= 0, NERF_E_SHAPE);
    NERF_REQUIRE(nerf_prec_ok(precision), NERF_E_DTYPE);
    if (M == 0) return NERF_OK;               // an empty call: its tensors have no address
    NERF_REQUIRE(x_pts && packed && raw, NERF_E_NULL);
    NERF_REQUIRE(x_views || (precision & NERF_ARCH_NOVIEW), NERF_E_NULL);
    NERF_REQUIRE(saved || workspace, NERF_E_NULL);
    NERF_REQUIRE(nerf_aligned(raw, 16) && nerf_aligned(packed, 256) && nerf_aligned(saved, 256) &&
                 nerf_aligned(workspace, 256), NERF_E_ALIGN);
    NERF_REQUIRE(nerf_prec_ok(precision), NERF_E_DTYPE);
    if (M == 0) return NERF_OK;
    return fwd_encoded_impl(M, x_pts, x_views, packed, precision, raw, saved, workspace, stream);
}

// ------------------------------------------------------------------------------------------------
// The same forward with its workspace in a static device array: an entry point that allocates nothing,
// for callers that cannot (HIP graph capture, serving loops).  The array is the fragment-unit staging
// buffer of nerf_mlp_fwd_encoded for at most NERF_MLP_FWD_ENCODED_STATIC_MAX samples (12 MiB); every
// GPU the library runs on holds its own copy until the process ends.  All of the part a call reads is
// written by that call's first kernel, so nothing has to start at zero; calls that overlap on one GPU
// (different streams or threads) share the array and race.
// ------------------------------------------------------------------------------------------------
constexpr int64_t STATIC_MAX_TILES = (NERF_MLP_FWD_ENCODED_STATIC_MAX + TILE_SAMPLES - 1) / TILE_SAMPLES;
constexpr int64_t STATIC_SCRATCH_BYTES = STATIC_MAX_TILES * 8 * (int64_t)(PE_KS + VPE_KS) * 1024;
static __device__ __align__(256) char scratch[STATIC_SCRATCH_BYTES];

// In host code `scratch` is only a handle for the symbol, not a device address: the address on each
// device comes from hipGetSymbolAddress with that device current, and is kept per device.
constexpr int MAX_DEVICES = 64;
static void* g_scratch_ptr[MAX_DEVICES];
static std::mutex g_scratch_mutex;

static int scratch_address(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    NERF_REQUIRE(dev >= 0 && dev < MAX_DEVICES, NERF_E_SHAPE);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        NERF_REQUIRE(p != nullptr, NERF_E_NULL);
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return NERF_OK;
}

extern "C" int64_t nerf_mlp_fwd_encoded_static_max_samples(void) { return NERF_MLP_FWD_ENCODED_STATIC_MAX; }

extern "C" int nerf_mlp_fwd_encoded_static(int64_t M, const float* x_pts, const float* x_views, const void* packed,
                                           int precision, float* raw, void* stream) {
    NERF_REQUIRE(M >= 0 && M <= NERF_MLP_FWD_ENCODED_STATIC_MAX, NERF_E_SHAPE);      // beyond the static buffer
    NERF_REQUIRE(nerf_prec_ok(precision), NERF_E_DTYPE);
    if (M == 0) return NERF_OK;               // an empty call: its tensors have no address
    NERF_REQUIRE(x_pts && packed && raw, NERF_E_NULL);
    NERF_REQUIRE(x_views || (precision & NERF_ARCH_NOVIEW), NERF_E_NULL);
    NERF_REQUIRE(nerf_aligned(raw, 16) && nerf_aligned(packed, 256), NERF_E_ALIGN);
    NERF_REQUIRE(nerf_prec_ok(precision), NERF_E_DTYPE);
    if (M == 0) return NERF_OK;
    void* ws = nullptr;
    const int rc = scratch_address(&ws);
    if (rc) return rc;
    return fwd_encoded_impl(M, x_pts, x_views, packed, precision, raw, nullptr, ws, stream);
}
#endif   // NERF_TU == 0
