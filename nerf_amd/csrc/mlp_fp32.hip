// fp32-operand forward of the NeRF MLP (src/nerf_project/models/nerf.py:173-209) on
// v_mfma_f32_32x32x2_f32, for every architecture the constructor accepts within the limits below
// (any D, W <= 256, any skips, with or without view directions).  It is the validation mode: operands
// are exact fp32 (the MFMA is a chain of fp32 fmas), so every ray meets the fp32 tolerance without a
// conditioning mask; it runs at the fp32 matrix rate, 1/16 of the 16-bit kernels.
//
// Structure.  The network is a list of layers (a host-side "program").  A workgroup of eight waves (two
// per SIMD, so that one wave's operand latency is covered by the other's MFMAs; each wave owns one
// 32-feature row tile of a layer at a time) owns 64 samples; their activations live in LDS for the whole network, as [k/4][sample][4] floats:
// four consecutive input features of one sample are one 16-byte read, which is the B operand of four
// MFMA k-steps (k-step j of an 8-feature chunk uses feature 4*(lane>>5) + j of the chunk on both
// operands).  An output tile's accumulator (column = sample on the lane, rows = features in the
// registers) holds, per register quad, four consecutive features of one sample: the epilogue writes it
// back with four 16-byte stores.  Weights are read from global memory (L2-resident) in a fragment order
// written by nerf_mlp_fp32_pack_segment: one coalesced 1 KiB read per tile and chunk.  A layer's input
// is one or two segments (skip connection: [encoded points, hidden]; colour head: [feature, encoded
// view direction]), its output goes to the other LDS buffer or straight to columns of raw [M,4].
#include <type_traits>
#include "common.hpp"
#include "mlp_prog.hpp"

namespace nerf {
namespace fp32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 64;                 // samples per workgroup
constexpr int THREADS = 512;
constexpr int WAVES = THREADS / 64;
constexpr int MAX_W = NERF_FP32_MAX_WIDTH;
constexpr int MAX_CP = NERF_FP32_MAX_PTS, MAX_CV = NERF_FP32_MAX_VIEWS;
constexpr int BUF_FLOATS_H = MAX_W * TILE;          // 64 KiB
constexpr int OFF_XP = 0, OFF_XV = OFF_XP + MAX_CP * TILE, OFF_HA = OFF_XV + MAX_CV * TILE, OFF_HB = OFF_HA + BUF_FLOATS_H;
constexpr int OFF_RED = OFF_HB + BUF_FLOATS_H;      // partial sums of the output layers: [wave][sample][4 rows]
constexpr int LDS_FLOATS = OFF_RED + WAVES * TILE * 4;   // 160 KiB
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS");

constexpr int KPAD = 8;                 // input features per weight chunk
typedef prog::Program<float> Program;     // layer list + the transposed side channels (mlp_prog.hpp)

// max(x, 0) that keeps a NaN a NaN, as torch.relu does (fmaxf would return 0: a diverged network must stay visible)
__device__ __forceinline__ float relu_keep_nan(float x) { return x < 0.0f ? 0.0f : x; }

__device__ __forceinline__ int buf_off(int id) {
    return id == NERF_FP32_BUF_PTS ? OFF_XP : id == NERF_FP32_BUF_VIEWS ? OFF_XV : id == NERF_FP32_BUF_A ? OFF_HA : OFF_HB;
}

// rows [M][C] fp32 -> LDS [ceil(C/4)*... ][sample][4], zero-padded to cpad features and to the tile
__device__ __forceinline__ void load_input(float* dst, const float* __restrict__ x, int64_t m0, int64_t M, int C, int cpad,
                                           float* __restrict__ xt, int64_t ldm) {
    for (int idx = threadIdx.x; idx < TILE * cpad; idx += THREADS) {
        const int n = idx / cpad, k = idx - n * cpad;
        const int64_t m = m0 + n;
        const float v = (k < C && m < M) ? x[m * C + k] : 0.0f;
        dst[((k >> 2) * TILE + n) * 4 + (k & 3)] = v;
        if (xt != nullptr) xt[(int64_t)k * ldm + m] = v;       // (samples past M: zeros, the rows are whole tiles)
    }
}

// What a wave needs at the start of a layer and that does not depend on the layer before: the bias of its row
// tile and the first two weight chunks.  They are requested one layer early, so that after the barrier the
// first MFMA waits for LDS only (eight waves leaving a barrier together otherwise all wait a full L2 round
// trip, twice: bias, then weights).  A wave owns at most one row tile per layer (width <= 256 = 8 tiles).
struct Head {
    f32x16 bias;
    f32x4 a0, a1;
};
// A layer of more than four row tiles gives each wave one tile and both 32-sample halves of the workgroup's 64
// samples; a narrower one (the 128-wide colour layer) gives waves w and w + 4 the same tile and one half each,
// so that no wave idles.
__device__ __forceinline__ bool halves_split(const nerf_fp32_layer_t& L) { return L.out_features <= 32 * (WAVES / 2); }
__device__ __forceinline__ int row_tile_of(const nerf_fp32_layer_t& L, int wave) { return halves_split(L) ? (wave & (WAVES / 2 - 1)) : wave; }

// The k loop of one row tile over a layer's input segments: HALVES = 2 accumulates both sample halves (two
// independent MFMA chains), HALVES = 1 only the half starting at sample `first` (acc0).
template <int HALVES>
__device__ __forceinline__ void tile_chunks(const nerf_fp32_layer_t& L, const float* lds, const int* seg_off, int ft, int first,
                                            int lane, const Head& head, f32x16& acc0, f32x16& acc1) {
    const int n = lane & 31, h = lane >> 5;
    for (int s = 0; s < L.n_segments; ++s) {
        const int nch = L.k_padded[s] >> 3;
        const f32x4* wf = reinterpret_cast<const f32x4*>(L.weights[s]) + ((size_t)ft * nch) * 64 + lane;
        const f32x4* src = reinterpret_cast<const f32x4*>(lds + seg_off[s]) + h * TILE + first + n;
        // weights two chunks ahead (L2 latency), activations one chunk ahead (LDS latency)
        f32x4 a0 = head.a0, a1 = head.a1;
        if (s > 0) { a0 = wf[0]; a1 = wf[(size_t)(1 < nch ? 1 : 0) * 64]; }
        f32x4 b0 = src[0], b1 = HALVES == 2 ? src[32] : b0;
        for (int c = 0; c < nch; ++c) {
            const int c1 = c + 1 < nch ? c + 1 : c, c2 = c + 2 < nch ? c + 2 : nch - 1;
            const f32x4 a2 = wf[(size_t)c2 * 64];
            const f32x4 nb0 = src[(2 * c1) * TILE];
            const f32x4 nb1 = HALVES == 2 ? src[(2 * c1) * TILE + 32] : nb0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc0, 0, 0, 0);
                if (HALVES == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc1, 0, 0, 0);
            }
            a0 = a1; a1 = a2; b0 = nb0; b1 = nb1;
        }
    }
}
__device__ __forceinline__ Head load_head(const Program& prog, int li, int wave, int lane) {
    Head hd;
#pragma unroll
    for (int r = 0; r < 16; ++r) hd.bias[r] = 0.0f;
    hd.a0 = f32x4{0.f, 0.f, 0.f, 0.f};
    hd.a1 = hd.a0;
    const nerf_fp32_layer_t& L = prog.layer[li];
    const int h = lane >> 5;
    const int ft = row_tile_of(L, wave);
    if (L.dst != NERF_FP32_DST_RAW && 32 * ft < L.out_features) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * ft + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (L.bias != nullptr && row < L.out_features) hd.bias[r] = L.bias[row];
        }
        const int nch = L.k_padded[0] >> 3;
        const f32x4* wf = reinterpret_cast<const f32x4*>(L.weights[0]) + ((size_t)ft * nch) * 64 + lane;
        hd.a0 = wf[0];
        hd.a1 = wf[(size_t)(1 < nch ? 1 : 0) * 64];
    }
    return hd;
}

__global__ __launch_bounds__(THREADS) void mlp_fp32_kernel(int64_t M, const float* __restrict__ x_pts, int cp,
                                                          const float* __restrict__ x_views, int cv, const Program prog,
                                                          float* __restrict__ raw) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int cp_pad = (cp + 7) & ~7, cv_pad = (cv + 7) & ~7;

    Head next_head = load_head(prog, 0, wave, lane);

    for (int64_t tile = blockIdx.x; tile * TILE < M; tile += gridDim.x) {
        const int64_t m0 = tile * TILE;
        __syncthreads();                      // the previous tile's readers are done with the buffers
        load_input(lds + OFF_XP, x_pts, m0, M, cp, cp_pad, prog.save_pts, prog.ldm);
        if (x_views != nullptr) load_input(lds + OFF_XV, x_views, m0, M, cv, cv_pad, prog.save_views, prog.ldm);
        __syncthreads();

        for (int li = 0; li < prog.n_layers; ++li) {
            const nerf_fp32_layer_t& L = prog.layer[li];
            const int n_ft = (L.out_features + 31) >> 5;
            const Head head = next_head;
            next_head = load_head(prog, li + 1 < prog.n_layers ? li + 1 : 0, wave, lane);      // (after the last layer: the next tile's first)
            if (L.dst == NERF_FP32_DST_RAW) {
                // An output layer (at most 4 rows) is one row tile: instead of one wave working through all of K
                // while seven wait, every wave takes an eighth of the input chunks and the partial sums of rows
                // 0..3 (registers 0..3 of lane half 0) meet in LDS.
                f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
                for (int s = 0; s < L.n_segments; ++s) {
                    const int nch = L.k_padded[s] >> 3;
                    const f32x4* wf = reinterpret_cast<const f32x4*>(L.weights[s]) + lane;
                    const f32x4* src = reinterpret_cast<const f32x4*>(lds + buf_off(L.src[s])) + h * TILE + n;
                    for (int c = wave * nch / WAVES; c < (wave + 1) * nch / WAVES; ++c) {
                        const f32x4 a = wf[(size_t)c * 64];
                        const f32x4 b0 = src[(2 * c) * TILE], b1 = src[(2 * c) * TILE + 32];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[j], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b1[j], acc1, 0, 0, 0);
                        }
                    }
                }
                f32x4* red = reinterpret_cast<f32x4*>(lds + OFF_RED) + wave * TILE;
                if (h == 0) {
                    red[n] = f32x4{acc0[0], acc0[1], acc0[2], acc0[3]};
                    red[32 + n] = f32x4{acc1[0], acc1[1], acc1[2], acc1[3]};
                }
                __syncthreads();
                for (int idx = threadIdx.x; idx < TILE * 4; idx += THREADS) {
                    const int smp = idx >> 2, row = idx & 3;
                    if (row < L.out_features && m0 + smp < M) {
                        float v = L.bias[row];
                        for (int w = 0; w < WAVES; ++w) v += lds[OFF_RED + (w * TILE + smp) * 4 + row];
                        raw[(m0 + smp) * 4 + L.dst_col + row] = L.relu ? relu_keep_nan(v) : v;
                    }
                }
                __syncthreads();              // the partial sums are read before another output layer rewrites them
                continue;
            }
            const bool split = halves_split(L);
            const int ft = row_tile_of(L, wave);
            if (ft < n_ft) {
                // accumulators of the 32-sample halves start from the bias of rows 32 ft + row(r, h)
                f32x16 acc0 = head.bias, acc1 = head.bias;
                const int seg_off[2] = {buf_off(L.src[0]), buf_off(L.src[L.n_segments > 1 ? 1 : 0])};
                const int first = split ? 32 * (wave / (WAVES / 2)) : 0;
                const float* gate = prog.gate[li];
                float* save = prog.save[li];
                // register r of lane (n, h) is feature 32 ft + 8 (r >> 2) + 4 h + (r & 3) of sample first + n (+ 32).  The
                // gate values are requested before the k loop: they arrive behind its MFMAs, not in front of the epilogue
                float g0[16], g1[16];
                if (gate != nullptr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = 32 * ft + 8 * (r >> 2) + 4 * h + (r & 3);
                        const int64_t at = (int64_t)(f < L.out_features ? f : 0) * prog.ldm + m0 + first + n;
                        g0[r] = gate[at];
                        g1[r] = split ? 1.0f : gate[at + 32];
                    }
                }
                if (split) tile_chunks<1>(L, lds, seg_off, ft, first, lane, head, acc0, acc1);
                else tile_chunks<2>(L, lds, seg_off, ft, 0, lane, head, acc0, acc1);
                if (L.relu) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc0[r] = relu_keep_nan(acc0[r]); acc1[r] = relu_keep_nan(acc1[r]); }
                }
                if (gate != nullptr || save != nullptr) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = 32 * ft + 8 * (r >> 2) + 4 * h + (r & 3);
                        if (f < L.out_features) {
                            const int64_t at = (int64_t)f * prog.ldm + m0 + first + n;
                            if (gate != nullptr) {
                                if (!(g0[r] > 0.0f)) acc0[r] = 0.0f;
                                if (!split && !(g1[r] > 0.0f)) acc1[r] = 0.0f;
                            }
                            if (save != nullptr) {
                                save[at] = acc0[r];
                                if (!split) save[at + 32] = acc1[r];
                            }
                        }
                    }
                }
                f32x4* dst = reinterpret_cast<f32x4*>(lds + buf_off(L.dst));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kq = 8 * ft + 2 * q + h;           // features 4 kq .. 4 kq + 3
                    const f32x4 v0 = {acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
                    const f32x4 v1 = {acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
                    dst[kq * TILE + first + n] = v0;
                    if (!split) dst[kq * TILE + 32 + n] = v1;
                }
            }
            __syncthreads();                  // the layer's output is complete before anyone reads it
        }
    }
}

// W[rows][ld] (columns col0 .. col0+cols) -> fragment order [row tile][chunk of 8 inputs][lane][4]:
// lane l of tile t, chunk c holds W[32 t + (l & 31)][col0 + 8 c + 4 (l >> 5) + j], j = 0..3; zero outside.
__global__ __launch_bounds__(256) void pack_segment_kernel(const float* __restrict__ w, int rows, int ld, int col0, int cols,
                                                         float* __restrict__ dst, int64_t total) {
    const int nch = (cols + 7) >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3), l = (int)((i >> 2) & 63);
        const int64_t tc = i >> 8;
        const int c = (int)(tc % nch), t = (int)(tc / nch);
        const int row = 32 * t + (l & 31), col = 8 * c + 4 * (l >> 5) + j;
        dst[i] = (row < rows && col < cols) ? w[(size_t)row * ld + col0 + col] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradients of one layer segment: dW[out][in] = sum over samples of dZ[out][m] * X[in][m], both operands in
// the transposed layout (a row per feature, samples along it), on v_mfma_f32_32x32x2_f32 with the sample as the
// contraction index.  A workgroup (four waves) owns one 32-row tile of dW over one chunk of the samples; wave w
// takes the column tiles w, w + 4, w + 8.  Per step of 32 samples a lane loads 16 consecutive samples of its row
// of each operand (64 bytes) and issues 16 MFMAs per column tile.  Every chunk writes an fp32 partial block;
// fp32_wgrad_reduce_kernel sums the chunks in index order (bitwise reproducible).  The bias gradient is the row
// sum of dZ, taken from the A operand by wave 0.
// ------------------------------------------------------------------------------------------------
constexpr int WG_MAX_COL_TILES = 3;      // per wave: 4 waves x 3 tiles x 32 = 384 >= 256 + 64 input features
struct WgradArgs {
    const float* At;      // dZ, [out rows][ldm]
    const float* Bt;      // layer input, [in rows][ldm]
    int64_t ldm;
    int64_t steps;        // ldm / 32
    int out, in;
    int out_pad, in_pad;  // rounded up to 32
    float* part;          // [chunks][out_pad][in_pad]
    float* part_bias;     // [chunks][out_pad]
};

__global__ __launch_bounds__(256) void fp32_wgrad_kernel(const WgradArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, kk = lane >> 5;
    // chunk on the fast grid axis: workgroups are dealt to the eight XCDs round robin, so with a chunk count that is a
    // multiple of eight the row tiles of one chunk -- which all read the same input rows -- share one XCD's L2
    const int chunk = blockIdx.x, ot = blockIdx.y, n_chunks = gridDim.x;
    const int n_it = a.in_pad >> 5;
    const int64_t per = (a.steps + n_chunks - 1) / n_chunks;
    const int64_t st0 = chunk * per, st1 = st0 + per < a.steps ? st0 + per : a.steps;
    f32x16 acc[WG_MAX_COL_TILES];
#pragma unroll
    for (int t = 0; t < WG_MAX_COL_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float bsum = 0.0f;
    const int rowA = 32 * ot + i;
    const bool validA = rowA < a.out;
    const f32x4* pA = reinterpret_cast<const f32x4*>(a.At + (int64_t)(validA ? rowA : 0) * a.ldm + 16 * kk);
    const f32x4* pB[WG_MAX_COL_TILES];
    bool validB[WG_MAX_COL_TILES];
#pragma unroll
    for (int t = 0; t < WG_MAX_COL_TILES; ++t) {
        const int rowB = 32 * (wave + 4 * t) + i;
        validB[t] = (wave + 4 * t) < n_it && rowB < a.in;
        pB[t] = reinterpret_cast<const f32x4*>(a.Bt + (int64_t)(validB[t] ? rowB : 0) * a.ldm + 16 * kk);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // U steps per iteration, all their loads issued before the first MFMA (a step alone keeps too few bytes in flight
    // to cover the memory latency)
    auto steps = [&](auto UC, int64_t st) {
        constexpr int U = decltype(UC)::value;
        const int64_t m4 = st * 8;            // 32 samples = 8 float4
        f32x4 av[U][4], bv[WG_MAX_COL_TILES][U][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) av[u][e] = validA ? pA[m4 + 8 * u + e] : zero4;
#pragma unroll
        for (int t = 0; t < WG_MAX_COL_TILES; ++t) {
            if (wave + 4 * t >= n_it) continue;          // wave-uniform
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[t][u][e] = validB[t] ? pB[t][m4 + 8 * u + e] : zero4;
        }
        if (wave == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum += (av[u][e][0] + av[u][e][1]) + (av[u][e][2] + av[u][e][3]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < WG_MAX_COL_TILES; ++t) {
                if (wave + 4 * t >= n_it) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][e][j], bv[t][u][e][j], acc[t], 0, 0, 0);
            }
    };
    int64_t st = st0;
    for (; st + 2 <= st1; st += 2) steps(std::integral_constant<int, 2>(), st);
    for (; st < st1; ++st) steps(std::integral_constant<int, 1>(), st);
    // accumulator register r of lane (n = lane & 31, h = lane >> 5): row 8 (r >> 2) + 4 h + (r & 3), column n
    float* part = a.part + (int64_t)chunk * a.out_pad * a.in_pad;
#pragma unroll
    for (int t = 0; t < WG_MAX_COL_TILES; ++t) {
        if (wave + 4 * t >= n_it) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * ot + 8 * (r >> 2) + 4 * kk + (r & 3);
            part[(int64_t)row * a.in_pad + 32 * (wave + 4 * t) + i] = acc[t][r];
        }
    }
    if (wave == 0) {
        const float other = __shfl_xor(bsum, 32);
        if (kk == 0) a.part_bias[(int64_t)chunk * a.out_pad + rowA] = bsum + other;
    }
}

}  // namespace fp32
}  // namespace nerf

using namespace nerf::fp32;
namespace prog = nerf::prog;

extern "C" int64_t nerf_mlp_fp32_segment_floats(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (int64_t)((rows + 31) / 32) * ((cols + 7) / 8) * 256;
}

extern "C" int nerf_mlp_fp32_pack_segment(const float* w, int rows, int ld, int col0, int cols, float* dst, void* stream) {
    NERF_REQUIRE(w && dst, NERF_E_NULL);
    NERF_REQUIRE(rows > 0 && cols > 0 && col0 >= 0 && ld >= col0 + cols, NERF_E_SHAPE);
    NERF_REQUIRE(nerf_aligned(dst, 16), NERF_E_ALIGN);
    const int64_t total = nerf_mlp_fp32_segment_floats(rows, cols);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_segment_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, rows, ld, col0, cols,
                       dst, total);
    return nerf_launch_status();
}

static inline int64_t fp32_ldm(int64_t M) { return (M + TILE - 1) / TILE * TILE; }
static inline unsigned fp32_grid(int64_t M) {
    const int64_t n_tiles = (M + TILE - 1) / TILE;
    return (unsigned)(n_tiles < 256 ? n_tiles : 256);
}

extern "C" int64_t nerf_mlp_fp32_saved_floats(int64_t M, int cp, int cv, const nerf_fp32_layer_t* layers, int n_layers) {
    if (M < 0 || !layers || n_layers < 1 || n_layers > NERF_FP32_MAX_LAYERS) return -1;
    return prog::saved_rows<KPAD>(cp, cv, layers, n_layers, nullptr) * fp32_ldm(M);
}

// workspace of nerf_mlp_fp32_bwd: the gradient rows (8 for d_raw, then every hidden layer's output gradient),
// then the partial blocks of the largest weight-gradient launch
static int64_t bwd_workspace(int64_t M, const nerf_fp32_layer_t* layers, int n_layers, int64_t* part_off) {
    const int64_t ldm = fp32_ldm(M);
    const int64_t off = (prog::grad_rows<KPAD>(layers, n_layers, nullptr) * ldm + 63) / 64 * 64;
    if (part_off) *part_off = off;
    return off + prog::partial_floats<KPAD>(ldm, layers, n_layers);
}

extern "C" int64_t nerf_mlp_fp32_bwd_workspace_floats(int64_t M, int cp, int cv, const nerf_fp32_layer_t* layers, int n_layers) {
    (void)cp; (void)cv;
    if (M < 0 || !layers || n_layers < 1 || n_layers > NERF_FP32_MAX_LAYERS) return -1;
    return bwd_workspace(M, layers, n_layers, nullptr);
}

static int fp32_forward(int64_t M, const float* x_pts, int cp, const float* x_views, int cv, const nerf_fp32_layer_t* layers,
                        int n_layers, float* raw, float* saved, void* stream) {
    NERF_REQUIRE(M >= 0, NERF_E_SHAPE);
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(x_pts && layers && raw, NERF_E_NULL);
    Program fwd;
    const int rc = prog::build_program<float, KPAD>(cp, cv, x_views != nullptr, layers, n_layers, &fwd, nullptr);
    if (rc) return rc;
    fwd.ldm = fp32_ldm(M);
    if (saved != nullptr) {
        NERF_REQUIRE(nerf_aligned(saved, 16), NERF_E_ALIGN);
        int64_t row[NERF_FP32_MAX_LAYERS];
        prog::saved_rows<KPAD>(cp, cv, layers, n_layers, row);
        fwd.save_pts = saved;
        fwd.save_views = x_views ? saved + (int64_t)prog::pad_k<KPAD>(cp) * fwd.ldm : nullptr;
        for (int i = 0; i < n_layers; ++i)
            if (layers[i].dst != NERF_FP32_DST_RAW) fwd.save[i] = saved + row[i] * fwd.ldm;
    }
    hipLaunchKernelGGL(mlp_fp32_kernel, dim3(fp32_grid(M)), dim3(THREADS), 0, (hipStream_t)stream, M, x_pts, cp, x_views, cv, fwd, raw);
    return nerf_launch_status();
}

extern "C" int nerf_mlp_fp32_fwd(int64_t M, const float* x_pts, int cp, const float* x_views, int cv,
                                 const nerf_fp32_layer_t* layers, int n_layers, float* raw, void* stream) {
    return fp32_forward(M, x_pts, cp, x_views, cv, layers, n_layers, raw, nullptr, stream);
}

extern "C" int nerf_mlp_fp32_fwd_train(int64_t M, const float* x_pts, int cp, const float* x_views, int cv,
                                       const nerf_fp32_layer_t* layers, int n_layers, float* raw, float* saved, void* stream) {
    NERF_REQUIRE(saved != nullptr || M == 0, NERF_E_NULL);
    return fp32_forward(M, x_pts, cp, x_views, cv, layers, n_layers, raw, saved, stream);
}

extern "C" int nerf_mlp_fp32_bwd(int64_t M, int cp, int cv, const nerf_fp32_layer_t* layers, const nerf_fp32_grad_t* grads,
                                 int n_layers, const float* d_raw, const float* saved, float* workspace,
                                 int64_t workspace_floats, void* stream) {
    NERF_REQUIRE(M >= 0, NERF_E_SHAPE);
    NERF_REQUIRE(layers && grads, NERF_E_NULL);
    Program fwd;
    int prod[NERF_FP32_MAX_LAYERS][2];
    int rc = prog::build_program<float, KPAD>(cp, cv, cv > 0, layers, n_layers, &fwd, prod);
    if (rc) return rc;
    rc = prog::check_grads<KPAD>(layers, grads, n_layers, prod);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {       // no samples: zero gradients
        for (int c = 0; c < n_layers; ++c) {
            hipError_t e = hipMemsetAsync(grads[c].d_weight, 0, sizeof(float) * layers[c].out_features * grads[c].ld, st);
            if (e == hipSuccess) e = hipMemsetAsync(grads[c].d_bias, 0, sizeof(float) * layers[c].out_features, st);
            if (e != hipSuccess) return (int)e;
        }
        return NERF_OK;
    }
    NERF_REQUIRE(d_raw && saved && workspace, NERF_E_NULL);
    NERF_REQUIRE(nerf_aligned(saved, 16) && nerf_aligned(workspace, 16), NERF_E_ALIGN);
    int64_t part_off = 0;
    NERF_REQUIRE(workspace_floats >= bwd_workspace(M, layers, n_layers, &part_off), NERF_E_WORKSPACE);
    const int64_t ldm = fp32_ldm(M);
    const int cp_pad = prog::pad_k<KPAD>(cp);
    int64_t yrow[NERF_FP32_MAX_LAYERS], zrow[NERF_FP32_MAX_LAYERS];
    prog::saved_rows<KPAD>(cp, cv, layers, n_layers, yrow);
    prog::grad_rows<KPAD>(layers, n_layers, zrow);
    // ---- 1. backward-data chain: the forward kernel on the transposed weights, hidden tensors in reverse order ----
    Program chain;
    int dead[NERF_FP32_MAX_LAYERS], n_dead = 0;
    rc = prog::build_chain<float, KPAD>(layers, grads, n_layers, prod, saved, workspace, yrow, zrow, ldm, &chain, dead, &n_dead);
    if (rc) return rc;
    for (int i = 0; i < n_dead; ++i) {       // nobody reads this layer: its gradient is zero
        const hipError_t e = hipMemsetAsync(workspace + zrow[dead[i]] * ldm, 0, sizeof(float) * layers[dead[i]].out_features * ldm, st);
        if (e != hipSuccess) return (int)e;
    }
    // (with no hidden layer at all the kernel still runs its input load, which writes d_raw transposed)
    hipLaunchKernelGGL(mlp_fp32_kernel, dim3(fp32_grid(M)), dim3(THREADS), 0, st, M, d_raw, 4, (const float*)nullptr, 0, chain,
                       (float*)nullptr);
    rc = nerf_launch_status();
    if (rc) return rc;
    // ---- 2. weight gradients, one launch pair per layer segment ----
    float* part = workspace + part_off;
    for (int c = 0; c < n_layers; ++c) {
        const nerf_fp32_layer_t& L = layers[c];
        const nerf_fp32_grad_t& G = grads[c];
        for (int s = 0; s < L.n_segments; ++s) {
            WgradArgs w{};
            w.At = workspace + zrow[c] * ldm;
            w.Bt = prod[c][s] == -1 ? saved : prod[c][s] == -2 ? saved + (int64_t)cp_pad * ldm : saved + yrow[prod[c][s]] * ldm;
            w.ldm = ldm; w.steps = ldm / 32;
            w.out = L.out_features; w.in = G.cols[s];
            w.out_pad = (w.out + 31) / 32 * 32; w.in_pad = (L.k_padded[s] + 31) / 32 * 32;
            NERF_REQUIRE(w.in_pad / 32 <= 4 * WG_MAX_COL_TILES, NERF_E_ARCH);
            const int chunks = prog::wgrad_chunks(w.steps, w.out_pad / 32);
            w.part = part;
            w.part_bias = part + (int64_t)chunks * w.out_pad * w.in_pad;
            hipLaunchKernelGGL(fp32_wgrad_kernel, dim3(chunks, w.out_pad / 32), dim3(256), 0, st, w);
            rc = nerf_launch_status();
            if (rc) return rc;
            prog::WgradReduceArgs r{};
            r.part = w.part; r.part_bias = w.part_bias; r.n_chunks = chunks; r.out = w.out; r.in = w.in;
            r.out_pad = w.out_pad; r.in_pad = w.in_pad;
            r.d_weight = G.d_weight; r.ld = G.ld; r.col0 = G.col0[s];
            r.d_bias = s == 0 ? G.d_bias : nullptr;
            r.scale = nullptr;
            const int64_t outputs = (int64_t)w.out * w.in + w.out;
            hipLaunchKernelGGL(prog::wgrad_reduce_kernel<0>, dim3((unsigned)((outputs + 255) / 256)), dim3(256), 0, st, r);
            rc = nerf_launch_status();
            if (rc) return rc;
        }
    }
    return NERF_OK;
}
