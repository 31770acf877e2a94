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
#include "common.hpp"

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

struct Program {
    nerf_fp32_layer_t layer[NERF_FP32_MAX_LAYERS];
    int n_layers;
};

// max(x, 0) that keeps a NaN a NaN, as torch.relu does (fmaxf would return 0: a diverged network must stay visible)
__device__ __forceinline__ float relu_keep_nan(float x) { return x < 0.0f ? 0.0f : x; }

__device__ __forceinline__ int buf_off(int id) {
    return id == NERF_FP32_BUF_PTS ? OFF_XP : id == NERF_FP32_BUF_VIEWS ? OFF_XV : id == NERF_FP32_BUF_A ? OFF_HA : OFF_HB;
}

// rows [M][C] fp32 -> LDS [ceil(C/4)*... ][sample][4], zero-padded to cpad features and to the tile
__device__ __forceinline__ void load_input(float* dst, const float* __restrict__ x, int64_t m0, int64_t M, int C, int cpad) {
    for (int idx = threadIdx.x; idx < TILE * cpad; idx += THREADS) {
        const int n = idx / cpad, k = idx - n * cpad;
        const int64_t m = m0 + n;
        const float v = (k < C && m < M) ? x[m * C + k] : 0.0f;
        dst[((k >> 2) * TILE + n) * 4 + (k & 3)] = v;
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
            if (row < L.out_features) hd.bias[r] = L.bias[row];
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
        load_input(lds + OFF_XP, x_pts, m0, M, cp, cp_pad);
        if (x_views != nullptr) load_input(lds + OFF_XV, x_views, m0, M, cv, cv_pad);
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
                if (split) tile_chunks<1>(L, lds, seg_off, ft, first, lane, head, acc0, acc1);
                else tile_chunks<2>(L, lds, seg_off, ft, 0, lane, head, acc0, acc1);
                if (L.relu) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc0[r] = relu_keep_nan(acc0[r]); acc1[r] = relu_keep_nan(acc1[r]); }
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

}  // namespace fp32
}  // namespace nerf

using namespace nerf::fp32;

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

extern "C" int nerf_mlp_fp32_fwd(int64_t M, const float* x_pts, int cp, const float* x_views, int cv,
                                 const nerf_fp32_layer_t* layers, int n_layers, float* raw, void* stream) {
    NERF_REQUIRE(M >= 0, NERF_E_SHAPE);
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(x_pts && layers && raw, NERF_E_NULL);
    NERF_REQUIRE(n_layers >= 1 && n_layers <= NERF_FP32_MAX_LAYERS, NERF_E_ARCH);
    NERF_REQUIRE(cp >= 1 && cp <= MAX_CP && cv >= 0 && cv <= MAX_CV && (x_views != nullptr || cv == 0), NERF_E_ARCH);
    Program prog;
    prog.n_layers = n_layers;
    // Every index a layer names is checked here: the kernel trusts the program.  `width[b]` is how many
    // features of buffer b hold defined values: the padded input widths, and for a hidden buffer the rows its
    // last writer produced (whole 32-row tiles: rows past out_features come out as zeros).  A layer may not
    // read beyond that, or it would multiply its zero-padded weights with stale LDS contents.
    int written = (1 << NERF_FP32_BUF_PTS) | (x_views ? 1 << NERF_FP32_BUF_VIEWS : 0);
    int width[4] = {(cp + 7) & ~7, (cv + 7) & ~7, 0, 0};
    int raw_cols = 0;
    for (int i = 0; i < n_layers; ++i) {
        const nerf_fp32_layer_t& L = layers[i];
        NERF_REQUIRE(L.n_segments >= 1 && L.n_segments <= 2 && L.bias, NERF_E_ARCH);
        NERF_REQUIRE(L.out_features >= 1, NERF_E_ARCH);
        for (int s = 0; s < L.n_segments; ++s) {
            NERF_REQUIRE(L.weights[s], NERF_E_NULL);
            NERF_REQUIRE(nerf_aligned(L.weights[s], 16), NERF_E_ALIGN);
            NERF_REQUIRE(L.src[s] >= NERF_FP32_BUF_PTS && L.src[s] <= NERF_FP32_BUF_B && ((written >> L.src[s]) & 1), NERF_E_ARCH);
            NERF_REQUIRE(L.k_padded[s] >= 8 && L.k_padded[s] % 8 == 0 && L.k_padded[s] <= width[L.src[s]], NERF_E_ARCH);
            NERF_REQUIRE(L.src[s] != L.dst, NERF_E_ARCH);
        }
        if (L.dst == NERF_FP32_DST_RAW) {
            NERF_REQUIRE(L.dst_col >= 0 && L.dst_col + L.out_features <= 4, NERF_E_ARCH);
            raw_cols |= ((1 << L.out_features) - 1) << L.dst_col;
        } else {
            NERF_REQUIRE(L.dst == NERF_FP32_BUF_A || L.dst == NERF_FP32_BUF_B, NERF_E_ARCH);
            NERF_REQUIRE(L.out_features <= MAX_W, NERF_E_ARCH);
            written |= 1 << L.dst;
            width[L.dst] = (L.out_features + 31) / 32 * 32;
        }
        prog.layer[i] = L;
    }
    NERF_REQUIRE(raw_cols == 0xf, NERF_E_ARCH);      // the program must write all four columns of raw
    const int64_t n_tiles = (M + TILE - 1) / TILE;
    const unsigned grid = (unsigned)(n_tiles < 256 ? n_tiles : 256);
    hipLaunchKernelGGL(mlp_fp32_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, M, x_pts, cp, x_views, cv, prog, raw);
    return nerf_launch_status();
}
