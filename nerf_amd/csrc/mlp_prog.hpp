// Host-side handling of a "layer program" -- the list of nerf_fp32_layer_t that describes one NeRF MLP
// (src/nerf_project/models/nerf.py:126-153, :173-209) to the program-driven kernels -- shared by the fp32 kernels
// (mlp_fp32.hip) and their 16-bit-operand counterparts (mlp_p16.hip): validation of the list, the row tables of the
// saved activations and of the gradient workspace, the construction of the backward-data chain (the forward kernel
// run on transposed weights), and the fixed-order sum of the weight-gradient partial blocks.
//
// E is the element type of the kernels' side channels (float, or the 16-bit operand type); KPAD the granularity the
// input features of a segment are padded to (one MFMA k-chunk: 8 inputs in fp32, 16 in the 16-bit kernels).
#pragma once
#include "common.hpp"

namespace nerf {
namespace prog {

// Optional per-layer side channels in global memory, all "transposed": one row of ldm elements per feature, samples
// along the row (ldm = M rounded up to whole workgroup tiles), so that a wave's 32 samples of one feature are one
// contiguous run.  The training forward keeps every hidden layer's output and the encoded inputs there; the
// backward-data chain gates its outputs with them (relu'(y) = [y > 0]) and keeps every layer's gradient for the
// weight-gradient pass.
template <class E>
struct Program {
    nerf_fp32_layer_t layer[NERF_FP32_MAX_LAYERS];
    E* save[NERF_FP32_MAX_LAYERS];         // nullable: rows = the layer's output features (after the epilogue)
    const E* gate[NERF_FP32_MAX_LAYERS];   // nullable: outputs are zeroed where this tensor is <= 0
    E* save_pts;                           // nullable: the padded input rows as loaded
    E* save_views;                         // nullable
    int64_t ldm;
    int n_layers;
};

template <int KPAD> constexpr int pad_k(int c) { return (c + KPAD - 1) / KPAD * KPAD; }

// Checks a layer list and copies it into a kernel program.  On success `prod[l][s]` names the layer whose output
// segment s of layer l reads (-1: encoded points, -2: encoded view directions).
template <class E, int KPAD>
static int build_program(int cp, int cv, bool have_views, const nerf_fp32_layer_t* layers, int n_layers, Program<E>* prog,
                         int (*prod)[2]) {
    NERF_REQUIRE(layers, NERF_E_NULL);
    NERF_REQUIRE(n_layers >= 1 && n_layers <= NERF_FP32_MAX_LAYERS, NERF_E_ARCH);
    NERF_REQUIRE(cp >= 1 && cp <= NERF_FP32_MAX_PTS && cv >= 0 && cv <= NERF_FP32_MAX_VIEWS && (have_views || cv == 0), NERF_E_ARCH);
    *prog = Program<E>{};
    prog->n_layers = n_layers;
    // Every index a layer names is checked here: the kernel trusts the program.  `width[b]` is how many
    // features of buffer b hold defined values: the padded input widths, and for a hidden buffer the rows its
    // last writer produced (whole 32-row tiles: rows past out_features come out as zeros).  A layer may not
    // read beyond that, or it would multiply its zero-padded weights with stale LDS contents.
    int written = (1 << NERF_FP32_BUF_PTS) | (have_views ? 1 << NERF_FP32_BUF_VIEWS : 0);
    int width[4] = {pad_k<KPAD>(cp), pad_k<KPAD>(cv), 0, 0};
    int live[4] = {-1, -2, -3, -3};
    int raw_cols = 0;
    for (int i = 0; i < n_layers; ++i) {
        const nerf_fp32_layer_t& L = layers[i];
        NERF_REQUIRE(L.n_segments >= 1 && L.n_segments <= 2 && L.bias, NERF_E_ARCH);
        NERF_REQUIRE(L.out_features >= 1, NERF_E_ARCH);
        for (int s = 0; s < L.n_segments; ++s) {
            NERF_REQUIRE(L.weights[s], NERF_E_NULL);
            NERF_REQUIRE(nerf_aligned(L.weights[s], 16), NERF_E_ALIGN);
            NERF_REQUIRE(L.src[s] >= NERF_FP32_BUF_PTS && L.src[s] <= NERF_FP32_BUF_B && ((written >> L.src[s]) & 1), NERF_E_ARCH);
            NERF_REQUIRE(L.k_padded[s] >= KPAD && L.k_padded[s] % KPAD == 0 && L.k_padded[s] <= width[L.src[s]], NERF_E_ARCH);
            NERF_REQUIRE(L.src[s] != L.dst, NERF_E_ARCH);
            if (prod) prod[i][s] = live[L.src[s]];
        }
        if (L.dst == NERF_FP32_DST_RAW) {
            NERF_REQUIRE(L.dst_col >= 0 && L.dst_col + L.out_features <= 4, NERF_E_ARCH);
            raw_cols |= ((1 << L.out_features) - 1) << L.dst_col;
        } else {
            NERF_REQUIRE(L.dst == NERF_FP32_BUF_A || L.dst == NERF_FP32_BUF_B, NERF_E_ARCH);
            NERF_REQUIRE(L.out_features <= NERF_FP32_MAX_WIDTH, NERF_E_ARCH);
            written |= 1 << L.dst;
            width[L.dst] = (L.out_features + 31) / 32 * 32;
            live[L.dst] = i;
        }
        prog->layer[i] = L;
    }
    NERF_REQUIRE(raw_cols == 0xf, NERF_E_ARCH);      // the program must write all four columns of raw
    return NERF_OK;
}

// rows of the saved buffer: encoded points (padded), encoded view directions (padded), then each hidden layer's output
template <int KPAD>
static int64_t saved_rows(int cp, int cv, const nerf_fp32_layer_t* layers, int n_layers, int64_t* row_of_layer) {
    int64_t rows = pad_k<KPAD>(cp) + pad_k<KPAD>(cv);
    for (int i = 0; i < n_layers; ++i) {
        if (row_of_layer) row_of_layer[i] = rows;
        if (layers[i].dst != NERF_FP32_DST_RAW) rows += layers[i].out_features;
    }
    return rows;
}

// rows of the gradient workspace: d_raw transposed (KPAD rows, written by the chain's input load; an output layer's dZ
// is rows dst_col .. of it), then dZ of every hidden layer
template <int KPAD>
static int64_t grad_rows(const nerf_fp32_layer_t* layers, int n_layers, int64_t* row_of_layer) {
    int64_t r = KPAD;
    for (int c = 0; c < n_layers; ++c) {
        if (layers[c].dst == NERF_FP32_DST_RAW) {
            if (row_of_layer) row_of_layer[c] = layers[c].dst_col;
        } else {
            if (row_of_layer) row_of_layer[c] = r;
            r += layers[c].out_features;
        }
    }
    return r;
}

// chunks of the sample axis of one weight-gradient launch over `out_tiles` workgroup rows: enough workgroups to fill
// the chip, at least four steps each
static inline int wgrad_chunks(int64_t steps, int out_tiles, int max_chunks = 64) {
    int64_t c = (512 + out_tiles - 1) / out_tiles;
    if (c > steps / 4) c = steps / 4;
    if (c > max_chunks) c = max_chunks;
    return c < 1 ? 1 : (int)c;
}

// floats of the partial blocks ([chunks][out_pad][in_pad] + [chunks][out_pad]) of the largest weight-gradient launch;
// a workgroup owns `row_tiles` 32-row tiles of dW
template <int KPAD>
static int64_t partial_floats(int64_t ldm, const nerf_fp32_layer_t* layers, int n_layers, int row_tiles = 1, int max_chunks = 64) {
    int64_t part = 0;
    for (int i = 0; i < n_layers; ++i) {
        const nerf_fp32_layer_t& L = layers[i];
        const int out_pad = (L.out_features + 31) / 32 * 32;
        for (int s = 0; s < L.n_segments; ++s) {
            const int in_pad = (L.k_padded[s] + 31) / 32 * 32;
            const int64_t need = (int64_t)wgrad_chunks(ldm / 32, (out_pad / 32 + row_tiles - 1) / row_tiles, max_chunks) * out_pad * (in_pad + 1);
            if (need > part) part = need;
        }
    }
    return part;
}

// Host check of the per-layer gradient descriptors against the layer list.
template <int KPAD>
static int check_grads(const nerf_fp32_layer_t* layers, const nerf_fp32_grad_t* grads, int n_layers, const int (*prod)[2]) {
    for (int c = 0; c < n_layers; ++c) {
        const nerf_fp32_layer_t& L = layers[c];
        const nerf_fp32_grad_t& G = grads[c];
        NERF_REQUIRE(G.d_weight && G.d_bias, NERF_E_NULL);
        int total = 0;
        for (int s = 0; s < L.n_segments; ++s) {
            NERF_REQUIRE(G.cols[s] >= 1 && pad_k<KPAD>(G.cols[s]) == L.k_padded[s] && G.col0[s] >= 0, NERF_E_ARCH);
            NERF_REQUIRE(G.col0[s] + G.cols[s] <= G.ld, NERF_E_ARCH);
            NERF_REQUIRE(prod[c][s] < 0 || (G.wt[s] != nullptr && nerf_aligned(G.wt[s], 16)), NERF_E_NULL);
            total += G.cols[s];
        }
        NERF_REQUIRE(total <= G.ld, NERF_E_ARCH);
    }
    return NERF_OK;
}

// The backward-data chain as a program for the forward kernel: hidden tensors in reverse order.  The gradient of a
// hidden layer p's output lives in the LDS buffer its output lived in.  It is the sum over the layers c that read that
// output of W_c[:, segment]^T dZ_c, where dZ_c is in c's own buffer -- or, for an output layer, in the input buffer
// that holds d_raw -- gated by [y_p > 0] when p has a ReLU.  `saved` / `work` are the forward's saved rows and the
// gradient rows (yrow / zrow: first row of each layer in them).  A hidden layer nobody reads gets a zero gradient:
// its index is appended to `dead` (the caller clears those rows).
template <class E, int KPAD>
static int build_chain(const nerf_fp32_layer_t* layers, const nerf_fp32_grad_t* grads, int n_layers, const int (*prod)[2],
                       const E* saved, E* work, const int64_t* yrow, const int64_t* zrow, int64_t ldm, Program<E>* chain,
                       int* dead, int* n_dead) {
    *chain = Program<E>{};
    chain->ldm = ldm;
    chain->save_pts = work;
    int n_chain = 0;
    *n_dead = 0;
    for (int p = n_layers - 1; p >= 0; --p) {
        const nerf_fp32_layer_t& P = layers[p];
        if (P.dst == NERF_FP32_DST_RAW) continue;
        nerf_fp32_layer_t C{};
        for (int c = p + 1; c < n_layers; ++c)
            for (int s = 0; s < layers[c].n_segments; ++s) {
                if (prod[c][s] != p) continue;
                NERF_REQUIRE(C.n_segments < 2, NERF_E_ARCH);       // at most two readers of one hidden tensor
                const bool out_layer = layers[c].dst == NERF_FP32_DST_RAW;
                C.weights[C.n_segments] = grads[c].wt[s];
                C.k_padded[C.n_segments] = out_layer ? KPAD : pad_k<KPAD>(layers[c].out_features);
                C.src[C.n_segments] = out_layer ? NERF_FP32_BUF_PTS : layers[c].dst;
                ++C.n_segments;
            }
        if (C.n_segments == 0) {
            dead[(*n_dead)++] = p;
            continue;
        }
        NERF_REQUIRE(n_chain < NERF_FP32_MAX_LAYERS, NERF_E_ARCH);
        C.bias = nullptr; C.out_features = P.out_features; C.relu = 0; C.dst = P.dst; C.dst_col = 0;
        chain->layer[n_chain] = C;
        chain->gate[n_chain] = P.relu ? saved + yrow[p] * ldm : nullptr;
        chain->save[n_chain] = work + zrow[p] * ldm;
        ++n_chain;
    }
    chain->n_layers = n_chain;
    return NERF_OK;
}

// Fixed-order sum of the chunks' partial blocks of one layer segment (bitwise reproducible), written to the
// segment's columns of d_weight; the first segment of a layer also sums the bias gradient.  `scale` (nullable,
// device): the gradients were computed from d_raw * *scale and are divided by it here.
struct WgradReduceArgs {
    const float* part; const float* part_bias;
    int n_chunks, out, in, out_pad, in_pad;
    float* d_weight; int ld, col0;     // d_weight[row * ld + col0 + col]
    float* d_bias;                     // nullable
    const float* scale;                // nullable
};
template <int UNUSED>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradReduceArgs a) {
    const int64_t total = (int64_t)a.out * a.in;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float inv = a.scale != nullptr ? 1.0f / *a.scale : 1.0f;       // (a power of two: the product is exact)
    if (i < total) {
        const int row = (int)(i / a.in), col = (int)(i - (int64_t)row * a.in);
        float v = 0.0f;
        for (int c = 0; c < a.n_chunks; ++c) v += a.part[((int64_t)c * a.out_pad + row) * a.in_pad + col];
        a.d_weight[(int64_t)row * a.ld + a.col0 + col] = a.scale != nullptr ? v * inv : v;
    } else if (a.d_bias != nullptr && i < total + a.out) {
        const int row = (int)(i - total);
        float v = 0.0f;
        for (int c = 0; c < a.n_chunks; ++c) v += a.part_bias[(int64_t)c * a.out_pad + row];
        a.d_bias[row] = a.scale != nullptr ? v * inv : v;
    }
}

}  // namespace prog
}  // namespace nerf
