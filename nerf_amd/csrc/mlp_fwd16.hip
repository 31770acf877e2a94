// Inference forward of the fused positional encoding + 8x256 MLP (src/nerf_project/models/nerf.py:68-76,
// :173-209, :409-421) on v_mfma_f32_16x16x32_{bf16,f16}: the no_grad render path, nothing saved.
//
// Same design as mlp_fwd.hip -- a wave keeps its 32 samples' activations in registers through all eleven
// layers, weights stream through the LDS ring -- with the narrow MFMA shape: the 32 samples are two
// 16-sample column tiles that share every 1 KiB weight block (16 output rows x 32 inputs), so one LDS
// read feeds two independent MFMAs, and two row tiles are interleaved: four accumulator chains per wave.
// Layout: "inference stream" in mlp_common.hpp.  1172 blocks = 2344 MFMAs of 16 cycles per 32 samples.
#include <type_traits>
#include "mlp_chain.hpp"

namespace nerf {
namespace mlp {

struct Fwd16Args {
    int64_t M;            // samples
    int64_t n_tiles;      // ceil(M / 256)
    int S;                // samples per ray
    const float* rays_o;  // [n_rays,3]
    const float* rays_d;  // [n_rays,3]
    const float* z;       // [M]
    const char* packed;
    float* raw;           // [M,4]
};

// Encoding fragments of one point for lane group g (see pe16_feature): sines or cosines (g odd: a
// quarter revolution of phase) of FH = L/2 frequencies starting at 2^(FH*(g>>1)), then one raw coordinate.
// Argument reduction as in mlp_fwd.hip: t = p/(2 pi) as a two-term product, 2^l * t_hi and fract() exact.
template <int PREC, int L, int NKS>
__device__ __forceinline__ void encode16(float px, float py, float pz, int g, typename Prec<PREC>::frag* out, int stride) {
    typedef typename Prec<PREC>::elem T;
    constexpr float C_HI = 0.15915494f;                 // fl32(1 / (2 pi))
    constexpr float C_LO = 6.4206382e-09f;              // 1/(2 pi) - C_HI
    constexpr int FH = L / 2, NQ = 3 * FH;
    const float p[3] = {px, py, pz};
    const float base = (g >> 1) ? (float)(1 << FH) : 1.0f;     // first frequency of this lane group
    float thi[3], tlo[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float hi = p[c] * C_HI;
        thi[c] = hi * base;                                     // exact: power of two
        tlo[c] = (fmaf(p[c], C_HI, -hi) + p[c] * C_LO) * base;
    }
    const float phase = (g & 1) ? 0.25f : 0.0f;
    const float rawc = g == 0 ? px : (g == 1 ? py : (g == 2 ? pz : 0.0f));
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 8 * ks + j;
            float v;
            if (q < NQ) {
                const float sc = (float)(1 << (q / 3));
                const float r = __builtin_amdgcn_fractf(thi[q % 3] * sc) + fmaf(tlo[q % 3], sc, phase);
                v = __builtin_amdgcn_sinf(r);
            } else {
                v = q == NQ ? rawc : 0.0f;
            }
            out[ks * stride][j] = (T)v;
        }
    }
}

// Inputs of the next pass travel global -> LDS by DMA while the current pass computes: per column tile
// z and the ray's origin and direction, one dword per lane each (IN_COPIES copies per wave and pass into
// a per-wave area behind the ring).
constexpr int IN_COPIES = 14;
constexpr int IN_LDS_OFF = CHAIN_LDS_BYTES;
constexpr int IN_WAVE_BYTES = IN_COPIES * 256;
constexpr int FWD16_LDS_BYTES = CHAIN_LDS_BYTES + 8 * IN_WAVE_BYTES;

// One layer on the narrow shape.  in0 / in1: fragments [k-step][column tile]; init(rt) gives the four
// bias values of this lane's rows of row tile rt; epi(rt, acc0, acc1) consumes the finished tile pair
// (one accumulator per column tile).  Chunk entries, A-block requests and waits as in run_layer.
template <int PREC, int B0, int NT, int NK0, int NK1, class Init, class Epi>
__device__ __forceinline__ void run_layer16(Ring& ring, char* lds, const typename Prec<PREC>::frag (*in0)[2],
                                            const typename Prec<PREC>::frag (*in1)[2], Init&& init, Epi&& epi) {
    typedef typename Prec<PREC>::frag frag;
    constexpr int NK = NK0 + NK1, NB = NT * NK;
    i32x4 a[NB];
    f32x4 acc[2][2], start[2];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int rt = stream_tile(NT, NK, i), ks = stream_ks(NT, NK, i);
        if ((B0 + i) % CHUNK_BLOCKS == 0) ring_enter<0, IN_COPIES>(ring, lds);
        request_blocks<B0, NB>(ring.rd_base, a, std::integral_constant<int, 0>(), i);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ahead = INIT_AHEAD < 2 * NK - 1 ? INIT_AHEAD : 2 * NK - 1;
            const int f = first_block(NT, NK, t);
            if ((f - ahead > 0 ? f - ahead : 0) == i) start[t & 1] = init(t);
        }
        wait_block<B0, NB>(a, i);
        const frag av = __builtin_bit_cast(frag, a[i]);
        const frag* b = ks < NK0 ? in0[ks] : in1[ks - NK0];
        acc[rt & 1][0] = Prec<PREC>::mfma16(av, b[0], ks == 0 ? start[rt & 1] : acc[rt & 1][0]);
        acc[rt & 1][1] = Prec<PREC>::mfma16(av, b[1], ks == 0 ? start[rt & 1] : acc[rt & 1][1]);
        if (ks == NK - 1) epi(rt, acc[rt & 1][0], acc[rt & 1][1]);
    }
}

// finished row tile -> elements 4*(rt&1) .. +3 of k-step rt>>1 of the next layer's input
template <int PREC, bool RELU>
__device__ __forceinline__ void tile16_to_frag(const f32x4& acc, int rt, typename Prec<PREC>::frag& f) {
    typedef typename Prec<PREC>::elem T;
#pragma unroll
    for (int r = 0; r < 4; ++r) f[4 * (rt & 1) + r] = (T)(RELU ? relu_bits(acc[r]) : acc[r]);
}

// VIEW = false: the network without view directions (output_linear follows L7; mlp_common.hpp)
template <int PREC, bool VIEW>
__global__ __launch_bounds__(CHAIN_THREADS, 2) void mlp_fwd16_kernel(const Fwd16Args a) {
    typedef typename Prec<PREC>::frag frag;
    __shared__ __attribute__((aligned(16))) char lds[FWD16_LDS_BYTES];
    float* bias_lds = reinterpret_cast<float*>(lds + LDS_BIAS_OFF);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef NERF_PRIO_UPPER        // experiment: static priority for the later-dispatched half of the workgroup
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    const int n = lane & 15, g = lane >> 4;

    load_bias_table<VIEW>(bias_lds, a.packed);       // (and the check for non-finite parameters)

    Ring ring;
    ring_init(ring, lds, a.packed + PACK_F16_OFF, VIEW ? F16_CHUNKS : NV_F16_CHUNKS, lane, wave);

    // Inputs of a pass.  issue_inputs() starts the copies (it must directly follow a chunk entry, see
    // ring_enter's SIDE); read_inputs() turns what has landed into points and unit view directions.
    const int in_base = IN_LDS_OFF + wave * IN_WAVE_BYTES;
    auto issue_inputs = [&](int64_t tile) {
        const int64_t wt = tile * 8 + wave;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            int64_t m = wt * WAVE_SAMPLES + 16 * ct + n;
            if (m >= a.M) m = a.M - 1;                        // lanes past the end copy a valid sample
            const int64_t ray = m / a.S;
            lds_char* l = (lds_char*)(lds + in_base + ct * 7 * 256);
            __builtin_amdgcn_global_load_lds((gbl_char*)(a.z + m), l, 4, 0, 0);
            const float* o = a.rays_o + 3 * ray;
            const float* d = a.rays_d + 3 * ray;
            // the instruction offset moves the LDS destination and the global source alike
            __builtin_amdgcn_global_load_lds((gbl_char*)(o + 0) - 1 * 256, l, 4, 1 * 256, 0);
            __builtin_amdgcn_global_load_lds((gbl_char*)(o + 1) - 2 * 256, l, 4, 2 * 256, 0);
            __builtin_amdgcn_global_load_lds((gbl_char*)(o + 2) - 3 * 256, l, 4, 3 * 256, 0);
            __builtin_amdgcn_global_load_lds((gbl_char*)(d + 0) - 4 * 256, l, 4, 4 * 256, 0);
            __builtin_amdgcn_global_load_lds((gbl_char*)(d + 1) - 5 * 256, l, 4, 5 * 256, 0);
            __builtin_amdgcn_global_load_lds((gbl_char*)(d + 2) - 6 * 256, l, 4, 6 * 256, 0);
        }
    };
    // what has landed -> fragments of the point encoding / of the view-direction encoding.  The
    // directions stay in LDS until the next issue_inputs(), so nothing is carried in registers.
    auto point_encoding = [&](frag (*pe)[2]) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float* f = reinterpret_cast<const float*>(lds + in_base + ct * 7 * 256) + lane;
            const float zz = f[0], ox = f[64], oy = f[128], oz = f[192], dx = f[256], dy = f[320], dz = f[384];
            encode16<PREC, 10, 2>(__fadd_rn(ox, __fmul_rn(dx, zz)), __fadd_rn(oy, __fmul_rn(dy, zz)),
                                  __fadd_rn(oz, __fmul_rn(dz, zz)), g, &pe[0][ct], 2);      // nerf.py:409
        }
    };
    auto view_encoding = [&](frag (*vpe)[2]) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float* f = reinterpret_cast<const float*>(lds + in_base + ct * 7 * 256) + lane;
            const float dx = f[256], dy = f[320], dz = f[384];
            const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
            encode16<PREC, 4, 1>(dx / nrm, dy / nrm, dz / nrm, g, &vpe[0][ct], 2);          // nerf.py:417
        }
    };

    frag pe[2][2];
    if ((int64_t)blockIdx.x < a.n_tiles) {
        issue_inputs(blockIdx.x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // first pass only: nothing to overlap with yet
        point_encoding(pe);
    }

    for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t wt = tile * 8 + wave;
        const bool has_next = tile + gridDim.x < a.n_tiles;
        frag vpe[1][2];
        if (VIEW) view_encoding(vpe);
        const bool valid0 = wt * WAVE_SAMPLES + n < a.M, valid1 = wt * WAVE_SAMPLES + 16 + n < a.M;

        frag ha[8][2], hb[8][2];
        auto bias = [&](int layer) {
            return [&, layer](int rt) {
                return *reinterpret_cast<const f32x4*>(bias_lds + layer * BIAS_STRIDE + 16 * rt + 4 * g);
            };
        };
        auto hidden = [&](frag (*out)[2]) {
            return [&, out](int rt, const f32x4& c0, const f32x4& c1) {
                tile16_to_frag<PREC, true>(c0, rt, out[rt >> 1][0]);
                tile16_to_frag<PREC, true>(c1, rt, out[rt >> 1][1]);
            };
        };
        ring_enter<0, IN_COPIES>(ring, lds); // chunk 0 of the pass: L0 starts behind the stream's front pad
        run_layer16<PREC, f16_block0(0), 16, 2, 0>(ring, lds, pe, pe, bias(0), hidden(ha));
        run_layer16<PREC, f16_block0(1), 16, 8, 0>(ring, lds, ha, ha, bias(1), hidden(hb));
        run_layer16<PREC, f16_block0(2), 16, 8, 0>(ring, lds, hb, hb, bias(2), hidden(ha));
        run_layer16<PREC, f16_block0(3), 16, 8, 0>(ring, lds, ha, ha, bias(3), hidden(hb));
        run_layer16<PREC, f16_block0(4), 16, 8, 0>(ring, lds, hb, hb, bias(4), hidden(ha));
        run_layer16<PREC, f16_block0(5), 16, 2, 8>(ring, lds, pe, ha, bias(5), hidden(hb));
        run_layer16<PREC, f16_block0(6), 16, 8, 0>(ring, lds, hb, hb, bias(6), hidden(ha));
        static_assert(f16_block0(7) % CHUNK_BLOCKS == 8, "the copies below assume at most 8 blocks since the last chunk entry");
        if (has_next) { issue_inputs(tile + gridDim.x); ring.extra_enters = 2; }   // land during L7 and L8
        run_layer16<PREC, f16_block0(7), 16, 8, 0>(ring, lds, ha, ha, bias(7), hidden(hb));

        if (!VIEW) {
            // L8 of the network without view directions: output_linear, rows 0..3 (lane group 0) = (rgb_raw, sigma_raw)
            run_layer16<PREC, f16_block0(8), 1, 8, 0>(ring, lds, hb, hb, bias(8),
                                                      [&](int, const f32x4& c0, const f32x4& c1) {
                if (g == 0) {
                    const int64_t m0 = wt * WAVE_SAMPLES + n;
                    if (valid0) reinterpret_cast<float4*>(a.raw)[m0] = make_float4(c0[0], c0[1], c0[2], c0[3]);
                    if (valid1) reinterpret_cast<float4*>(a.raw)[m0 + 16] = make_float4(c1[0], c1[1], c1[2], c1[3]);
                }
            });
            if (has_next) point_encoding(pe);
            continue;
        }
        // L8: feature_linear (no activation) -> ha, sigma_linear -> row 0 of row tile 16 (lane group 0)
        float sigma[2] = {0.f, 0.f};
        run_layer16<PREC, f16_block0(8), 17, 8, 0>(ring, lds, hb, hb, bias(8),
                                                   [&](int rt, const f32x4& c0, const f32x4& c1) {
            if (rt < 16) {
                tile16_to_frag<PREC, false>(c0, rt, ha[rt >> 1][0]);
                tile16_to_frag<PREC, false>(c1, rt, ha[rt >> 1][1]);
            } else {
                sigma[0] = c0[0]; sigma[1] = c1[0];
            }
        });
        if (has_next) point_encoding(pe);                 // its registers are free again after L5
        // L9: rgb_layers.0 on [feature, view encoding] -> relu -> G (first 4 k-steps of hb)
        run_layer16<PREC, f16_block0(9), 8, 8, 1>(ring, lds, ha, vpe, bias(9), hidden(hb));
        // L10: rgb_layers.2 -> rows 0..2 (lane group 0); (rgb_raw, sigma_raw) per sample
        run_layer16<PREC, f16_block0(10), 1, 4, 0>(ring, lds, hb, hb, bias(10),
                                                   [&](int, const f32x4& c0, const f32x4& c1) {
            if (g == 0) {
                const int64_t m0 = wt * WAVE_SAMPLES + n;
                if (valid0) reinterpret_cast<float4*>(a.raw)[m0] = make_float4(c0[0], c0[1], c0[2], sigma[0]);
                if (valid1) reinterpret_cast<float4*>(a.raw)[m0 + 16] = make_float4(c1[0], c1[1], c1[2], sigma[1]);
            }
        });
    }
    // LDS-DMA prefetched for a pass that does not exist must land before the wave ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace mlp
}  // namespace nerf

using namespace nerf::mlp;

// called by nerf_mlp_fwd when no activations are to be saved
int nerf_mlp_fwd16_launch(int64_t n_rays, int S, const float* rays_o, const float* rays_d, const float* z,
                          const void* packed, int precision, float* raw, void* stream) {
    Fwd16Args a{};
    a.M = n_rays * S; a.n_tiles = (a.M + TILE_SAMPLES - 1) / TILE_SAMPLES; a.S = S;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.packed = (const char*)packed; a.raw = raw;
    if (a.M == 0) return NERF_OK;
    const int64_t grid = a.n_tiles < 256 ? a.n_tiles : 256;   // one persistent workgroup per CU
    const dim3 g((unsigned)grid), b(CHAIN_THREADS);
    const bool view = !(precision & NERF_ARCH_NOVIEW);
    precision &= ~NERF_ARCH_NOVIEW;
    hipStream_t st = (hipStream_t)stream;
    if (precision == NERF_PREC_BF16) {
        if (view) hipLaunchKernelGGL((mlp_fwd16_kernel<NERF_PREC_BF16, true>), g, b, 0, st, a);
        else hipLaunchKernelGGL((mlp_fwd16_kernel<NERF_PREC_BF16, false>), g, b, 0, st, a);
    } else {
        if (view) hipLaunchKernelGGL((mlp_fwd16_kernel<NERF_PREC_FP16, true>), g, b, 0, st, a);
        else hipLaunchKernelGGL((mlp_fwd16_kernel<NERF_PREC_FP16, false>), g, b, 0, st, a);
    }
    return nerf_launch_status();
}
