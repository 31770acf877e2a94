// 16-bit-operand (bf16 / fp16) forward and backward of the NeRF MLP (src/nerf_project/models/nerf.py:173-209) for
// every architecture the constructor (:100-153) accepts within the limits of the layer program (any D, W <= 256,
// any skips, with or without view directions) on v_mfma_f32_32x32x16_{bf16,f16}: the 16-bit counterpart of
// mlp_fp32.hip, for the networks the unrolled chain kernels (mlp_fwd.hip, mlp_fwd16.hip, mlp_bwd.hip) are not
// compiled for.  Same arithmetic model as those: weights and every layer input rounded to the operand type, fp32
// accumulation, fp32 bias, fp32 raw.
//
// Structure.  The network is a list of layers (the host-side "program" of mlp_prog.hpp).  A workgroup of eight
// waves owns 128 samples; their activations live in LDS for the whole network as [k/8][sample][8] 16-bit values:
// eight consecutive input features of one sample are one 16-byte read, which is the B operand of one MFMA (lane
// half h of 16-input chunk c reads feature group 2c + h).  A wave owns one 32-feature row tile of a layer and all
// four 32-sample column tiles (four independent accumulator chains share every weight fragment); narrower layers
// give several waves the same row tile and a part of the samples each.  Weights are read from global memory
// (L2-resident) in a fragment order written by nerf_mlp_p16_pack_segment: one coalesced 1 KiB read per row tile
// and 16-input chunk.  An accumulator's register quad holds four consecutive features of one sample and goes back
// to LDS as one 8-byte store.  A layer's output goes to the other LDS buffer or straight to columns of raw [M,4].
#include <type_traits>
#include "mlp_common.hpp"
#include "mlp_prog.hpp"

namespace nerf {
namespace p16 {

using mlp::Prec;
using mlp::f32x16;
using mlp::f32x4;
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int KPAD = 16;                 // input features per weight chunk (the K of one MFMA)
constexpr int TILE = 128;                // samples per workgroup
constexpr int THREADS = 512;             // eight waves, two per SIMD
constexpr int MAX_W = NERF_FP32_MAX_WIDTH;
constexpr int MAX_CP = NERF_FP32_MAX_PTS, MAX_CV = NERF_FP32_MAX_VIEWS;
// LDS map in 16-bit elements
constexpr int OFF_XP = 0, OFF_XV = OFF_XP + MAX_CP * TILE, OFF_HA = OFF_XV + MAX_CV * TILE, OFF_HB = OFF_HA + MAX_W * TILE;
constexpr int ACT_ELEMS = OFF_HB + MAX_W * TILE;
// partial sums of the output layers: [K half][sample][4 rows] floats
constexpr int RED_BYTES = 2 * TILE * 4 * 4;
constexpr int LDS_BYTES = ACT_ELEMS * 2 + RED_BYTES;      // 156 KiB
static_assert(LDS_BYTES <= 160 * 1024 && MAX_CP % KPAD == 0 && MAX_CV % KPAD == 0, "LDS");

template <int PREC> using Program = prog::Program<typename Prec<PREC>::elem>;

// max(x, 0) that keeps a NaN a NaN, as torch.relu does (a diverged network must stay visible)
__device__ __forceinline__ float relu_keep_nan(float x) { return x < 0.0f ? 0.0f : x; }

__device__ __forceinline__ int buf_off(int id) {
    return id == NERF_FP32_BUF_PTS ? OFF_XP : id == NERF_FP32_BUF_VIEWS ? OFF_XV : id == NERF_FP32_BUF_A ? OFF_HA : OFF_HB;
}

// rows [M][C] fp32 (times `scale`) -> LDS [cpad/8][sample][8] in the operand type, zero-padded to cpad features and
// to the tile.  All threads; the caller's barrier makes it visible.
template <class T>
__device__ __forceinline__ void load_input(T* dst, const float* __restrict__ x, int64_t m0, int64_t M, int C, int cpad, float scale) {
    for (int idx = threadIdx.x; idx < TILE * cpad; idx += THREADS) {
        const int n = idx / cpad, k = idx - n * cpad;
        const int64_t m = m0 + n;
        const float v = (k < C && m < M) ? x[m * C + k] * scale : 0.0f;
        dst[((k >> 3) * TILE + n) * 8 + (k & 7)] = (T)v;
    }
}
// the tile's input rows from LDS to the transposed side channel (a row per feature: consecutive lanes write
// consecutive samples); samples past M are zeros, the rows are whole tiles
template <class T>
__device__ __forceinline__ void save_input(const T* src, T* __restrict__ xt, int64_t m0, int cpad, int64_t ldm) {
    for (int idx = threadIdx.x; idx < TILE * cpad; idx += THREADS) {
        const int k = idx / TILE, n = idx - k * TILE;
        xt[(int64_t)k * ldm + m0 + n] = src[((k >> 3) * TILE + n) * 8 + (k & 7)];
    }
}

// How a layer's row tiles and the workgroup's four column tiles are dealt to the eight waves: more than four row
// tiles -> a wave owns one row tile and all four column tiles; three or four -> waves w and w + 4 share a row tile,
// two column tiles each; one or two -> four waves per row tile, one column tile each.
__device__ __forceinline__ int subs_per_wave(int n_ft) { return n_ft > 4 ? 4 : n_ft > 2 ? 2 : 1; }
__device__ __forceinline__ int row_tile_of(int n_ft, int wave) { return n_ft > 4 ? wave : n_ft > 2 ? (wave & 3) : (wave & 1); }
__device__ __forceinline__ int first_sub_of(int n_ft, int wave) { return n_ft > 4 ? 0 : n_ft > 2 ? 2 * (wave >> 2) : (wave >> 1); }

// What a wave needs at the start of a layer and that does not depend on the layer before: the bias of its row
// tile and the first two weight chunks.  They are requested one layer early, so that after the barrier the
// first MFMA waits for LDS only.
struct Head {
    f32x16 bias;
    i32x4 a0, a1;
};
template <int PREC>
__device__ __forceinline__ Head load_head(const Program<PREC>& pr, int li, int wave, int lane) {
    Head hd;
#pragma unroll
    for (int r = 0; r < 16; ++r) hd.bias[r] = 0.0f;
    hd.a0 = i32x4{0, 0, 0, 0};
    hd.a1 = hd.a0;
    const nerf_fp32_layer_t& L = pr.layer[li];
    const int h = lane >> 5;
    const int n_ft = (L.out_features + 31) >> 5;
    const int ft = row_tile_of(n_ft, wave);
    if (L.dst != NERF_FP32_DST_RAW && ft < n_ft) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * ft + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (L.bias != nullptr && row < L.out_features) hd.bias[r] = L.bias[row];
        }
        const int nch = L.k_padded[0] >> 4;
        const i32x4* wf = reinterpret_cast<const i32x4*>(L.weights[0]) + ((size_t)ft * nch) * 64 + lane;
        hd.a0 = wf[0];
        hd.a1 = wf[(size_t)(1 < nch ? 1 : 0) * 64];
    }
    return hd;
}

// The k loop of one row tile over a layer's input segments for NS column tiles starting at sample `first`.
template <int PREC, int NS>
__device__ __forceinline__ void tile_chunks(const nerf_fp32_layer_t& L, const char* lds, const int* seg_off, int ft, int first,
                                            int lane, const Head& head, f32x16 (&acc)[NS]) {
    typedef typename Prec<PREC>::frag frag;
    const int n = lane & 31, h = lane >> 5;
    for (int s = 0; s < L.n_segments; ++s) {
        const int nch = L.k_padded[s] >> 4;
        const i32x4* wf = reinterpret_cast<const i32x4*>(L.weights[s]) + ((size_t)ft * nch) * 64 + lane;
        // 16-byte groups: group g of sample m sits at index g * TILE + m
        const i32x4* src = reinterpret_cast<const i32x4*>(lds + 2 * seg_off[s]) + h * TILE + first + n;
        // weights two chunks ahead (L2 latency), activations one chunk ahead (LDS latency)
        i32x4 a0 = head.a0, a1 = head.a1;
        if (s > 0) { a0 = wf[0]; a1 = wf[(size_t)(1 < nch ? 1 : 0) * 64]; }
        i32x4 b[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) b[t] = src[32 * t];
        for (int c = 0; c < nch; ++c) {
            const int c1 = c + 1 < nch ? c + 1 : c, c2 = c + 2 < nch ? c + 2 : nch - 1;
            const i32x4 a2 = wf[(size_t)c2 * 64];
            i32x4 nb[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) nb[t] = src[(2 * c1) * TILE + 32 * t];
#pragma unroll
            for (int t = 0; t < NS; ++t)
                acc[t] = Prec<PREC>::mfma(__builtin_bit_cast(frag, a0), __builtin_bit_cast(frag, b[t]), acc[t]);
            a0 = a1; a1 = a2;
#pragma unroll
            for (int t = 0; t < NS; ++t) b[t] = nb[t];
        }
    }
}

// One hidden layer for this wave: NS column tiles of row tile ft.
template <int PREC, int NS>
__device__ __forceinline__ void hidden_layer(const Program<PREC>& pr, int li, char* lds, int ft, int first, int lane, int64_t m0,
                                             const Head& head) {
    typedef typename Prec<PREC>::elem T;
    const nerf_fp32_layer_t& L = pr.layer[li];
    const int n = lane & 31, h = lane >> 5;
    f32x16 acc[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) acc[t] = head.bias;
    const T* gate = pr.gate[li];
    T* save = pr.save[li];
    // register r of lane (n, h) is feature 32 ft + 8 (r >> 2) + 4 h + (r & 3) of sample first + 32 t + n.  The gate
    // values are requested before the k loop, so that they arrive behind its MFMAs instead of in front of the epilogue
    T gv[16][NS];
    if (gate != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * ft + 8 * (r >> 2) + 4 * h + (r & 3);
            const int64_t at = (int64_t)(f < L.out_features ? f : 0) * pr.ldm + m0 + first + n;
#pragma unroll
            for (int t = 0; t < NS; ++t) gv[r][t] = gate[at + 32 * t];
        }
    }
    const int seg_off[2] = {buf_off(L.src[0]), buf_off(L.src[L.n_segments > 1 ? 1 : 0])};
    tile_chunks<PREC, NS>(L, lds, seg_off, ft, first, lane, head, acc);
    if (L.relu) {
#pragma unroll
        for (int t = 0; t < NS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = relu_keep_nan(acc[t][r]);
    }
    if (gate != nullptr || save != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * ft + 8 * (r >> 2) + 4 * h + (r & 3);
            if (f < L.out_features) {
                const int64_t at = (int64_t)f * pr.ldm + m0 + first + n;
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    if (gate != nullptr && !((float)gv[r][t] > 0.0f)) acc[t][r] = 0.0f;
                    if (save != nullptr) save[at + 32 * t] = (T)acc[t][r];
                }
            }
        }
    }
    // features 32 ft + 8 q + 4 h .. + 3 of a sample: the h-th half of 16-byte group 4 ft + q
    char* dst = lds + 2 * buf_off(L.dst);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            typedef T T4 __attribute__((ext_vector_type(4)));
            T4 v;
            v[0] = (T)acc[t][4 * q]; v[1] = (T)acc[t][4 * q + 1]; v[2] = (T)acc[t][4 * q + 2]; v[3] = (T)acc[t][4 * q + 3];
            *reinterpret_cast<T4*>(dst + ((size_t)((4 * ft + q) * TILE + first + 32 * t + n)) * 16 + 8 * h) = v;
        }
    }
}

// `in_scale` (nullable, device): the rows of x_pts are multiplied by it on entry (the backward-data chain's d_raw,
// kept inside the fp16 range).
template <int PREC>
__global__ __launch_bounds__(THREADS) void mlp_p16_kernel(int64_t M, const float* __restrict__ x_pts, int cp,
                                                         const float* __restrict__ x_views, int cv, const Program<PREC> pr,
                                                         float* __restrict__ raw, const float* __restrict__ in_scale) {
    typedef typename Prec<PREC>::elem T;
    typedef typename Prec<PREC>::frag frag;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    T* const act = reinterpret_cast<T*>(lds);
    float* const red = reinterpret_cast<float*>(lds + ACT_ELEMS * 2);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int cp_pad = prog::pad_k<KPAD>(cp), cv_pad = prog::pad_k<KPAD>(cv);
    const float scale = in_scale != nullptr ? *in_scale : 1.0f;

    Head next_head = load_head<PREC>(pr, 0, wave, lane);

    for (int64_t tile = blockIdx.x; tile * TILE < M; tile += gridDim.x) {
        const int64_t m0 = tile * TILE;
        __syncthreads();                      // the previous tile's readers are done with the buffers
        load_input<T>(act + OFF_XP, x_pts, m0, M, cp, cp_pad, scale);
        if (x_views != nullptr) load_input<T>(act + OFF_XV, x_views, m0, M, cv, cv_pad, 1.0f);
        __syncthreads();
        if (pr.save_pts != nullptr) save_input<T>(act + OFF_XP, pr.save_pts, m0, cp_pad, pr.ldm);
        if (x_views != nullptr && pr.save_views != nullptr) save_input<T>(act + OFF_XV, pr.save_views, m0, cv_pad, pr.ldm);

        for (int li = 0; li < pr.n_layers; ++li) {
            const nerf_fp32_layer_t& L = pr.layer[li];
            const int n_ft = (L.out_features + 31) >> 5;
            const Head head = next_head;
            next_head = load_head<PREC>(pr, li + 1 < pr.n_layers ? li + 1 : 0, wave, lane);      // (after the last layer: the next tile's first)
            if (L.dst == NERF_FP32_DST_RAW) {
                // An output layer (at most 4 rows) is one row tile: wave w takes column tile w & 3 and one half of
                // the input chunks; the two partial sums of rows 0..3 (registers 0..3 of lane half 0) meet in LDS.
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                const int sub = wave & 3, kh = wave >> 2;
                for (int s = 0; s < L.n_segments; ++s) {
                    const int nch = L.k_padded[s] >> 4;
                    const i32x4* wf = reinterpret_cast<const i32x4*>(L.weights[s]) + lane;
                    const i32x4* src = reinterpret_cast<const i32x4*>(lds + 2 * buf_off(L.src[s])) + h * TILE + 32 * sub + n;
                    for (int c = kh * nch / 2; c < (kh + 1) * nch / 2; ++c) {
                        const i32x4 a = wf[(size_t)c * 64];
                        const i32x4 b = src[(2 * c) * TILE];
                        acc = Prec<PREC>::mfma(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), acc);
                    }
                }
                if (h == 0) reinterpret_cast<f32x4*>(red)[kh * TILE + 32 * sub + n] = f32x4{acc[0], acc[1], acc[2], acc[3]};
                __syncthreads();
                for (int idx = threadIdx.x; idx < TILE * 4; idx += THREADS) {
                    const int smp = idx >> 2, row = idx & 3;
                    if (row < L.out_features && m0 + smp < M) {
                        const float v = L.bias[row] + (red[smp * 4 + row] + red[(TILE + smp) * 4 + row]);
                        raw[(m0 + smp) * 4 + L.dst_col + row] = L.relu ? relu_keep_nan(v) : v;
                    }
                }
                __syncthreads();              // the partial sums are read before another output layer rewrites them
                continue;
            }
            const int ft = row_tile_of(n_ft, wave);
            if (ft < n_ft) {
                const int first = 32 * first_sub_of(n_ft, wave);
                const int ns = subs_per_wave(n_ft);
                if (ns == 4) hidden_layer<PREC, 4>(pr, li, lds, ft, first, lane, m0, head);
                else if (ns == 2) hidden_layer<PREC, 2>(pr, li, lds, ft, first, lane, m0, head);
                else hidden_layer<PREC, 1>(pr, li, lds, ft, first, lane, m0, head);
            }
            __syncthreads();                  // the layer's output is complete before anyone reads it
        }
    }
}

// W[rows][ld] (columns col0 .. col0+cols) -> fragment order [row tile][chunk of 16 inputs][lane][8] in the operand
// type: lane l of tile t, chunk c holds W[32 t + (l & 31)][col0 + 16 c + 8 (l >> 5) + j], j = 0..7; zero outside.
template <class T>
__global__ __launch_bounds__(256) void pack_segment_kernel(const float* __restrict__ w, int rows, int ld, int col0, int cols,
                                                         T* __restrict__ dst, int64_t total) {
    const int nch = (cols + KPAD - 1) / KPAD;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
        const int64_t tc = i >> 9;
        const int c = (int)(tc % nch), t = (int)(tc / nch);
        const int row = 32 * t + (l & 31), col = 16 * c + 8 * (l >> 5) + j;
        dst[i] = (T)((row < rows && col < cols) ? w[(size_t)row * ld + col0 + col] : 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradients of one layer segment: dW[out][in] = sum over samples of dZ[out][m] * X[in][m], both operands in
// the transposed 16-bit layout (a row per feature, samples along it), with the sample as the contraction index of
// v_mfma_f32_32x32x16.  A workgroup (four waves) owns one 32-row tile of dW over one chunk of the samples; wave w
// takes the column tiles w, w + 4, w + 8.  Per step of 32 samples a lane loads 16 consecutive samples of its row
// of each operand (32 bytes: two fragments; which sample a fragment element stands for is the same on both
// operands, and that is all a sum over samples needs).  Every chunk writes an fp32 partial block;
// prog::wgrad_reduce_kernel sums the chunks in index order (bitwise reproducible).  The bias gradient is the row
// sum of dZ, taken from the A operand by wave 0.
// ------------------------------------------------------------------------------------------------
constexpr int WG_MAX_COL_TILES = 3;      // per wave: 4 waves x 3 tiles x 32 = 384 >= 256 + 64 input features
template <class T>
struct WgradArgs {
    const T* At;          // dZ, [out rows][ldm]
    const T* Bt;          // layer input, [in rows][ldm]
    int64_t ldm;
    int64_t steps;        // ldm / 32
    int out, in;
    int out_pad, in_pad;  // rounded up to 32
    float* part;          // [chunks][out_pad][in_pad]
    float* part_bias;     // [chunks][out_pad]
};

constexpr int WG_ROW_TILES = 2;          // 32-row tiles of dW per workgroup: every input fragment a wave loads meets both
constexpr int WG_MAX_CHUNKS = 128;       // of the sample axis (half as many workgroup rows as row tiles: twice the chunks)
template <int PREC>
__global__ __launch_bounds__(256) void p16_wgrad_kernel(const WgradArgs<typename Prec<PREC>::elem> a) {
    typedef typename Prec<PREC>::frag frag;
    constexpr int RT = WG_ROW_TILES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, kk = lane >> 5;
    // chunk on the fast grid axis: workgroups are dealt to the eight XCDs round robin, so with a chunk count that is a
    // multiple of eight the row tiles of one chunk -- which all read the same input rows -- share one XCD's L2
    const int chunk = blockIdx.x, ot0 = blockIdx.y * RT, n_chunks = gridDim.x;
    const int n_it = a.in_pad >> 5;
    const int64_t per = (a.steps + n_chunks - 1) / n_chunks;
    const int64_t st0 = chunk * per, st1 = st0 + per < a.steps ? st0 + per : a.steps;
    f32x16 acc[RT][WG_MAX_COL_TILES];
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
        for (int t = 0; t < WG_MAX_COL_TILES; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][t][r] = 0.0f;
    float bsum[RT];
    // 16-byte fragments: fragment index = sample / 8
    const frag* pA[RT];
    bool validA[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        const int rowA = 32 * (ot0 + q) + i;
        bsum[q] = 0.0f;
        validA[q] = rowA < a.out;
        pA[q] = reinterpret_cast<const frag*>(a.At + (int64_t)(validA[q] ? rowA : 0) * a.ldm) + 2 * kk;
    }
    const bool second = 32 * (ot0 + 1) < a.out_pad;       // workgroup-uniform: the last workgroup of an odd tile count has one
    const frag* pB[WG_MAX_COL_TILES];
    bool validB[WG_MAX_COL_TILES];
#pragma unroll
    for (int t = 0; t < WG_MAX_COL_TILES; ++t) {
        const int rowB = 32 * (wave + 4 * t) + i;
        validB[t] = (wave + 4 * t) < n_it && rowB < a.in;
        pB[t] = reinterpret_cast<const frag*>(a.Bt + (int64_t)(validB[t] ? rowB : 0) * a.ldm) + 2 * kk;
    }
    frag zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (typename Prec<PREC>::elem)0.0f;
    // U steps per iteration, all their loads issued before the first MFMA: a step alone keeps too few 16-byte loads
    // per lane in flight to cover the memory latency
    auto steps = [&](auto UC, int64_t st) {
        constexpr int U = decltype(UC)::value;
        const int64_t f4 = st * 4;            // 32 samples = 4 fragments
        frag av[RT][U][2], bv[WG_MAX_COL_TILES][U][2];
#pragma unroll
        for (int q = 0; q < RT; ++q)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 2; ++e) av[q][u][e] = validA[q] ? pA[q][f4 + 4 * u + e] : zero;
#pragma unroll
        for (int t = 0; t < WG_MAX_COL_TILES; ++t) {
            if (wave + 4 * t >= n_it) continue;          // wave-uniform
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 2; ++e) bv[t][u][e] = validB[t] ? pB[t][f4 + 4 * u + e] : zero;
        }
        if (wave == 0) {
#pragma unroll
            for (int q = 0; q < RT; ++q)
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int j = 0; j < 8; ++j) bsum[q] += (float)av[q][u][e][j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < WG_MAX_COL_TILES; ++t) {
                if (wave + 4 * t >= n_it) continue;
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int q = 0; q < RT; ++q)
                        if (q == 0 || second) acc[q][t] = Prec<PREC>::mfma(av[q][u][e], bv[t][u][e], acc[q][t]);
            }
    };
    int64_t st = st0;
    for (; st + 2 <= st1; st += 2) steps(std::integral_constant<int, 2>(), st);
    for (; st < st1; ++st) steps(std::integral_constant<int, 1>(), st);
    // accumulator register r of lane (n = lane & 31, h = lane >> 5): row 8 (r >> 2) + 4 h + (r & 3), column n
    float* part = a.part + (int64_t)chunk * a.out_pad * a.in_pad;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        if (q > 0 && !second) break;
#pragma unroll
        for (int t = 0; t < WG_MAX_COL_TILES; ++t) {
            if (wave + 4 * t >= n_it) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (ot0 + q) + 8 * (r >> 2) + 4 * kk + (r & 3);
                part[(int64_t)row * a.in_pad + 32 * (wave + 4 * t) + i] = acc[q][t][r];
            }
        }
        if (wave == 0) {
            const float other = __shfl_xor(bsum[q], 32);
            if (kk == 0) a.part_bias[(int64_t)chunk * a.out_pad + 32 * (ot0 + q) + i] = bsum[q] + other;
        }
    }
}

}  // namespace p16
}  // namespace nerf

using namespace nerf::p16;
namespace prog = nerf::prog;
using nerf::mlp::Prec;

static inline int64_t p16_ldm(int64_t M) { return (M + TILE - 1) / TILE * TILE; }
static inline unsigned p16_grid(int64_t M) {
    const int64_t n_tiles = (M + TILE - 1) / TILE;
    return (unsigned)(n_tiles < 256 ? n_tiles : 256);
}
static inline bool p16_prec_ok(int precision) { return precision == NERF_PREC_BF16 || precision == NERF_PREC_FP16; }

extern "C" int64_t nerf_mlp_p16_segment_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (int64_t)((rows + 31) / 32) * ((cols + KPAD - 1) / KPAD) * 1024;
}

extern "C" int nerf_mlp_p16_pack_segment(const float* w, int rows, int ld, int col0, int cols, int precision, void* dst,
                                         void* stream) {
    NERF_REQUIRE(w && dst, NERF_E_NULL);
    NERF_REQUIRE(p16_prec_ok(precision), NERF_E_DTYPE);
    NERF_REQUIRE(rows > 0 && cols > 0 && col0 >= 0 && ld >= col0 + cols, NERF_E_SHAPE);
    NERF_REQUIRE(nerf_aligned(dst, 16), NERF_E_ALIGN);
    const int64_t total = nerf_mlp_p16_segment_bytes(rows, cols) / 2;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (precision == NERF_PREC_BF16)
        hipLaunchKernelGGL(pack_segment_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, rows, ld, col0,
                           cols, (__bf16*)dst, total);
    else
        hipLaunchKernelGGL(pack_segment_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, rows, ld, col0,
                           cols, (_Float16*)dst, total);
    return nerf_launch_status();
}

extern "C" int64_t nerf_mlp_p16_saved_bytes(int64_t M, int cp, int cv, const nerf_fp32_layer_t* layers, int n_layers) {
    if (M < 0 || !layers || n_layers < 1 || n_layers > NERF_FP32_MAX_LAYERS) return -1;
    return prog::saved_rows<KPAD>(cp, cv, layers, n_layers, nullptr) * p16_ldm(M) * 2;
}

// workspace of nerf_mlp_p16_bwd: the 16-bit gradient rows (16 for d_raw, then every hidden layer's output gradient),
// then, 256-byte aligned, the fp32 partial blocks of the largest weight-gradient launch
static int64_t bwd_workspace(int64_t M, const nerf_fp32_layer_t* layers, int n_layers, int64_t* part_off) {
    const int64_t ldm = p16_ldm(M);
    const int64_t off = (prog::grad_rows<KPAD>(layers, n_layers, nullptr) * ldm * 2 + 255) / 256 * 256;
    if (part_off) *part_off = off;
    return off + prog::partial_floats<KPAD>(ldm, layers, n_layers, WG_ROW_TILES, WG_MAX_CHUNKS) * 4;
}

extern "C" int64_t nerf_mlp_p16_bwd_workspace_bytes(int64_t M, int cp, int cv, const nerf_fp32_layer_t* layers, int n_layers) {
    (void)cp; (void)cv;
    if (M < 0 || !layers || n_layers < 1 || n_layers > NERF_FP32_MAX_LAYERS) return -1;
    return bwd_workspace(M, layers, n_layers, nullptr);
}

template <int PREC>
static int p16_forward(int64_t M, const float* x_pts, int cp, const float* x_views, int cv, const nerf_fp32_layer_t* layers,
                       int n_layers, float* raw, void* saved_v, hipStream_t st) {
    typedef typename Prec<PREC>::elem T;
    Program<PREC> fwd;
    const int rc = prog::build_program<T, KPAD>(cp, cv, x_views != nullptr, layers, n_layers, &fwd, nullptr);
    if (rc) return rc;
    fwd.ldm = p16_ldm(M);
    if (saved_v != nullptr) {
        T* saved = (T*)saved_v;
        int64_t row[NERF_FP32_MAX_LAYERS];
        prog::saved_rows<KPAD>(cp, cv, layers, n_layers, row);
        fwd.save_pts = saved;
        fwd.save_views = x_views ? saved + (int64_t)prog::pad_k<KPAD>(cp) * fwd.ldm : nullptr;
        for (int i = 0; i < n_layers; ++i)
            if (layers[i].dst != NERF_FP32_DST_RAW) fwd.save[i] = saved + row[i] * fwd.ldm;
    }
    hipLaunchKernelGGL(mlp_p16_kernel<PREC>, dim3(p16_grid(M)), dim3(THREADS), 0, st, M, x_pts, cp, x_views, cv, fwd, raw,
                       (const float*)nullptr);
    return nerf_launch_status();
}

static int p16_forward_any(int64_t M, const float* x_pts, int cp, const float* x_views, int cv, const nerf_fp32_layer_t* layers,
                           int n_layers, int precision, float* raw, void* saved, void* stream) {
    NERF_REQUIRE(M >= 0, NERF_E_SHAPE);
    NERF_REQUIRE(p16_prec_ok(precision), NERF_E_DTYPE);
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(x_pts && layers && raw, NERF_E_NULL);
    NERF_REQUIRE(saved == nullptr || nerf_aligned(saved, 16), NERF_E_ALIGN);
    if (precision == NERF_PREC_BF16)
        return p16_forward<NERF_PREC_BF16>(M, x_pts, cp, x_views, cv, layers, n_layers, raw, saved, (hipStream_t)stream);
    return p16_forward<NERF_PREC_FP16>(M, x_pts, cp, x_views, cv, layers, n_layers, raw, saved, (hipStream_t)stream);
}

extern "C" int nerf_mlp_p16_fwd(int64_t M, const float* x_pts, int cp, const float* x_views, int cv,
                                const nerf_fp32_layer_t* layers, int n_layers, int precision, float* raw, void* stream) {
    return p16_forward_any(M, x_pts, cp, x_views, cv, layers, n_layers, precision, raw, nullptr, stream);
}

extern "C" int nerf_mlp_p16_fwd_train(int64_t M, const float* x_pts, int cp, const float* x_views, int cv,
                                      const nerf_fp32_layer_t* layers, int n_layers, int precision, float* raw, void* saved,
                                      void* stream) {
    NERF_REQUIRE(saved != nullptr || M == 0, NERF_E_NULL);
    return p16_forward_any(M, x_pts, cp, x_views, cv, layers, n_layers, precision, raw, saved, stream);
}

template <int PREC>
static int p16_backward(int64_t M, int cp, int cv, const nerf_fp32_layer_t* layers, const nerf_fp32_grad_t* grads, int n_layers,
                        const int (*prod)[2], const float* d_raw, const void* saved_v, void* workspace, int64_t part_off,
                        const float* scale, hipStream_t st) {
    typedef typename Prec<PREC>::elem T;
    const T* saved = (const T*)saved_v;
    T* work = (T*)workspace;
    const int64_t ldm = p16_ldm(M);
    const int cp_pad = prog::pad_k<KPAD>(cp);
    int64_t yrow[NERF_FP32_MAX_LAYERS], zrow[NERF_FP32_MAX_LAYERS];
    prog::saved_rows<KPAD>(cp, cv, layers, n_layers, yrow);
    prog::grad_rows<KPAD>(layers, n_layers, zrow);
    // ---- 1. backward-data chain: the forward kernel on the transposed weights, hidden tensors in reverse order ----
    Program<PREC> chain;
    int dead[NERF_FP32_MAX_LAYERS], n_dead = 0;
    int rc = prog::build_chain<T, KPAD>(layers, grads, n_layers, prod, saved, work, yrow, zrow, ldm, &chain, dead, &n_dead);
    if (rc) return rc;
    for (int i = 0; i < n_dead; ++i) {       // nobody reads this layer: its gradient is zero
        const hipError_t e = hipMemsetAsync(work + zrow[dead[i]] * ldm, 0, sizeof(T) * layers[dead[i]].out_features * ldm, st);
        if (e != hipSuccess) return (int)e;
    }
    // (with no hidden layer at all the kernel still runs its input load, which writes d_raw transposed)
    hipLaunchKernelGGL(mlp_p16_kernel<PREC>, dim3(p16_grid(M)), dim3(THREADS), 0, st, M, d_raw, 4, (const float*)nullptr, 0, chain,
                       (float*)nullptr, scale);
    rc = nerf_launch_status();
    if (rc) return rc;
    // ---- 2. weight gradients, one launch pair per layer segment ----
    float* part = reinterpret_cast<float*>((char*)workspace + part_off);
    for (int c = 0; c < n_layers; ++c) {
        const nerf_fp32_layer_t& L = layers[c];
        const nerf_fp32_grad_t& G = grads[c];
        for (int s = 0; s < L.n_segments; ++s) {
            WgradArgs<T> w{};
            w.At = work + zrow[c] * ldm;
            w.Bt = prod[c][s] == -1 ? saved : prod[c][s] == -2 ? saved + (int64_t)cp_pad * ldm : saved + yrow[prod[c][s]] * ldm;
            w.ldm = ldm; w.steps = ldm / 32;
            w.out = L.out_features; w.in = G.cols[s];
            w.out_pad = (w.out + 31) / 32 * 32; w.in_pad = (L.k_padded[s] + 31) / 32 * 32;
            NERF_REQUIRE(w.in_pad / 32 <= 4 * WG_MAX_COL_TILES, NERF_E_ARCH);
            const int row_groups = (w.out_pad / 32 + WG_ROW_TILES - 1) / WG_ROW_TILES;
            const int chunks = prog::wgrad_chunks(w.steps, row_groups, WG_MAX_CHUNKS);
            w.part = part;
            w.part_bias = part + (int64_t)chunks * w.out_pad * w.in_pad;
            hipLaunchKernelGGL(p16_wgrad_kernel<PREC>, dim3(chunks, row_groups), dim3(256), 0, st, w);
            rc = nerf_launch_status();
            if (rc) return rc;
            prog::WgradReduceArgs r{};
            r.part = w.part; r.part_bias = w.part_bias; r.n_chunks = chunks; r.out = w.out; r.in = w.in;
            r.out_pad = w.out_pad; r.in_pad = w.in_pad;
            r.d_weight = G.d_weight; r.ld = G.ld; r.col0 = G.col0[s];
            r.d_bias = s == 0 ? G.d_bias : nullptr;
            r.scale = scale;
            const int64_t outputs = (int64_t)w.out * w.in + w.out;
            hipLaunchKernelGGL(prog::wgrad_reduce_kernel<1>, dim3((unsigned)((outputs + 255) / 256)), dim3(256), 0, st, r);
            rc = nerf_launch_status();
            if (rc) return rc;
        }
    }
    return NERF_OK;
}

extern "C" int nerf_mlp_p16_bwd(int64_t M, int cp, int cv, const nerf_fp32_layer_t* layers, const nerf_fp32_grad_t* grads,
                                int n_layers, int precision, const float* d_raw, const void* saved, const float* grad_scale,
                                void* workspace, int64_t workspace_bytes, void* stream) {
    NERF_REQUIRE(M >= 0, NERF_E_SHAPE);
    NERF_REQUIRE(p16_prec_ok(precision), NERF_E_DTYPE);
    NERF_REQUIRE(layers && grads, NERF_E_NULL);
    prog::Program<char> check;        // validation only: the element type does not matter
    int prod[NERF_FP32_MAX_LAYERS][2];
    int rc = prog::build_program<char, KPAD>(cp, cv, cv > 0, layers, n_layers, &check, prod);
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
    NERF_REQUIRE(nerf_aligned(saved, 16) && nerf_aligned(workspace, 256), NERF_E_ALIGN);
    int64_t part_off = 0;
    NERF_REQUIRE(workspace_bytes >= bwd_workspace(M, layers, n_layers, &part_off), NERF_E_WORKSPACE);
    if (precision == NERF_PREC_BF16)
        return p16_backward<NERF_PREC_BF16>(M, cp, cv, layers, grads, n_layers, prod, d_raw, saved, workspace, part_off, grad_scale, st);
    return p16_backward<NERF_PREC_FP16>(M, cp, cv, layers, grads, n_layers, prod, d_raw, saved, workspace, part_off, grad_scale, st);
}
