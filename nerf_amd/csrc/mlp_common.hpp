// Layout contract shared by the MLP kernels (pack, forward, backward chain, weight gradient).
//
// Network (src/nerf_project/models/nerf.py:126-209, D=8, W=256, skips=[4], use_viewdirs):
//   L0  pts_linears.0   PE(63)            -> 256  relu
//   L1-4 pts_linears.1-4 H(256)           -> 256  relu
//   L5  pts_linears.5   [PE(63), H(256)]  -> 256  relu
//   L6-7 pts_linears.6-7 H(256)           -> 256  relu
//   L8  feature_linear + sigma_linear  H(256) -> 256 (no activation) + 1
//   L9  rgb_layers.0    [F(256), VPE(27)] -> 128  relu
//   L10 rgb_layers.2    G(128)            -> 3
//
// Orientation.  Every product is computed transposed, Y[out, sample] = W[out, in] * X[in, sample],
// with v_mfma_f32_32x32x16_{bf16,f16}: A = a 32-row tile of W, B = 16 input features of 32
// samples.  A wave owns 32 samples for the whole network.  The accumulator tile of one layer
// (column = sample on the lane, rows = features in the 16 registers) is, after conversion to
// 16 bit, directly the B operand of the next layer: register group 8s..8s+7 of output tile m is
// the fragment of k-step 2m+s.  Activations therefore never leave the register file.
//
// Slots.  An activation vector is a list of k-steps of 16 "slots".  Lane half h = lane>>5,
// fragment element j (0..7) of k-step ks is slot
//        slot(ks, h, j) = 16*ks + 8*(j>>2) + 4*h + (j&3)
// which for hidden activations equals the feature index (it is the C/D row map of the MFMA).
// The positional encodings are generated straight into fragments with a slot order chosen so
// that lane half 0 evaluates sines and lane half 1 cosines (pe_slot_feature below).
//
// Weight stream.  For layer L, output tile m (32 rows), k-step ks: one 1 KiB block holding
// lane l's A fragment at byte 16*l:  element j = W[32m + (l&31)][feature of slot(ks, l>>5, j)].
// Blocks are stored in consumption order (two output tiles at a time, k-steps interleaved, see
// stream_tile); 32 blocks form a 32 KiB chunk, the unit of the LDS ring.  The training forward
// stream has 1184 blocks behind a 16-block front pad (38 chunks); the backward (transposed) stream,
// whose A operand is a 32-row tile of W^T, has 1100 blocks behind a 4-block pad (35 chunks).  The
// inference forward uses a third stream tiled for v_mfma_f32_16x16x32 (see "inference stream").
#pragma once
#include "common.hpp"

namespace nerf {
namespace mlp {

constexpr int W_ = 256;          // hidden width
constexpr int PE_FEATS = 63;     // 3 + 3*2*10
constexpr int VPE_FEATS = 27;    // 3 + 3*2*4
constexpr int PE_KS = 4;         // 64 slots
constexpr int VPE_KS = 2;        // 32 slots
constexpr int H_KS = 16;         // 256 slots
constexpr int G_KS = 8;          // 128 slots
constexpr int NLAYERS = 11;

constexpr int TILE_SAMPLES = 256;   // samples per workgroup pass (8 waves x 32)
constexpr int WAVE_SAMPLES = 32;
constexpr int BLOCK_BYTES = 1024;
constexpr int CHUNK_BLOCKS = 32;
constexpr int CHUNK_BYTES = CHUNK_BLOCKS * BLOCK_BYTES;
// slots of the LDS weight ring: one being read, the others in flight or landed.  A translation unit may ask
// for more than three (NERF_RING_SLOTS before the include) where its LDS budget allows: a chunk is then
// requested further ahead, which gives the wave's older stores longer to complete before a chunk entry has
// to wait for them (the memory counter is in order).
#ifndef NERF_RING_SLOTS
#define NERF_RING_SLOTS 3
#endif
constexpr int RING_SLOTS = NERF_RING_SLOTS;
static_assert(RING_SLOTS >= 3, "ring");

// ---- flat parameter offsets (state_dict order) ------------------------------------------
constexpr int OFF_W0 = 0;                       constexpr int OFF_B0 = OFF_W0 + 256 * 63;
constexpr int OFF_W1 = OFF_B0 + 256;            constexpr int OFF_B1 = OFF_W1 + 256 * 256;
constexpr int OFF_W2 = OFF_B1 + 256;            constexpr int OFF_B2 = OFF_W2 + 256 * 256;
constexpr int OFF_W3 = OFF_B2 + 256;            constexpr int OFF_B3 = OFF_W3 + 256 * 256;
constexpr int OFF_W4 = OFF_B3 + 256;            constexpr int OFF_B4 = OFF_W4 + 256 * 256;
constexpr int OFF_W5 = OFF_B4 + 256;            constexpr int OFF_B5 = OFF_W5 + 256 * 319;
constexpr int OFF_W6 = OFF_B5 + 256;            constexpr int OFF_B6 = OFF_W6 + 256 * 256;
constexpr int OFF_W7 = OFF_B6 + 256;            constexpr int OFF_B7 = OFF_W7 + 256 * 256;
constexpr int OFF_WS = OFF_B7 + 256;            constexpr int OFF_BS = OFF_WS + 256;        // sigma_linear
constexpr int OFF_WF = OFF_BS + 1;              constexpr int OFF_BF = OFF_WF + 256 * 256;  // feature_linear
constexpr int OFF_WR0 = OFF_BF + 256;           constexpr int OFF_BR0 = OFF_WR0 + 128 * 283;
constexpr int OFF_WR2 = OFF_BR0 + 128;          constexpr int OFF_BR2 = OFF_WR2 + 3 * 128;
constexpr int N_PARAMS = OFF_BR2 + 3;
static_assert(N_PARAMS == NERF_MLP_NPARAMS, "parameter count");

// ---- order of a layer's blocks in a stream ---------------------------------------------------------
// A wave that feeds one accumulator tile with dependent MFMAs back to back leaves the matrix pipe idle
// between them (two waves per SIMD do not cover the dependent-issue latency: a bare single-chain loop
// reaches about 1.3 PFLOP/s, two chains 1.8).  Output tiles are therefore taken two at a time and their
// k-steps interleaved: (tile 2p, ks 0), (tile 2p+1, ks 0), (2p, 1), (2p+1, 1), ...  An odd last tile
// runs alone.  i is the block's position inside the layer, NT x NK the layer's tiles and k-steps.
__host__ __device__ constexpr int stream_tile(int NT, int NK, int i) {
    const int p = i / (2 * NK), r = i - p * 2 * NK;
    return (2 * p + 1 < NT) ? 2 * p + (r & 1) : 2 * p;
}
__host__ __device__ constexpr int stream_ks(int NT, int NK, int i) {
    const int p = i / (2 * NK), r = i - p * 2 * NK;
    return (2 * p + 1 < NT) ? (r >> 1) : r;
}

// ---- forward stream: blocks per layer = out_tiles * k_steps ------------------------------
//                                   L0  L1   L2   L3   L4   L5   L6   L7   L8   L9  L10
constexpr int FWD_TILES[NLAYERS] = {  8,  8,   8,   8,   8,   8,   8,   8,   9,   4,   1};
constexpr int FWD_KS[NLAYERS]    = {  4, 16,  16,  16,  16,  20,  16,  16,  16,  18,   8};
// The stream starts with FWD_PAD zero blocks that are never read: they shift the layers so that a tile
// pair of a 16-k-step layer (32 blocks) ends in the middle of a 32-block chunk instead of at its end.
// A pair's epilogue (bias, ReLU, conversion: vector ALU work) can then be overlapped with the next
// pair's MFMAs; at a chunk end it would sit in front of the chunk-entry barrier with nothing to overlap.
constexpr int FWD_PAD = 16;
constexpr int fwd_block0(int layer) {
    int b = FWD_PAD;
    for (int l = 0; l < layer; ++l) b += FWD_TILES[l] * FWD_KS[l];
    return b;
}
constexpr int FWD_BLOCKS_REAL = fwd_block0(NLAYERS);     // 1200
constexpr int FWD_CHUNKS = (FWD_BLOCKS_REAL + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;   // 38
constexpr int FWD_BLOCKS = FWD_CHUNKS * CHUNK_BLOCKS;    // 1216 (pad and tail blocks are zero, never read)
static_assert(FWD_BLOCKS_REAL == 1200 && FWD_PAD % 4 == 0, "forward stream");

// ---- backward (dgrad) stream, consumption order L10, L9, L8, L7, ..., L1 --------------------
// index d = 0..9 for layers 10, 9, ..., 1 (L0 needs no input gradient)
//   d0  L10: dG[128]   = W_r2^T   [128 x 16(3)]    4 tiles x 1 ks
//   d1  L9 : dF[256]   = W_r0^T   [256 x 128]      8 tiles x 8 ks   (view-direction rows not needed)
//   d2  L8 : dH7[256]  = [W_f; w_sigma]^T [256 x 272] 8 tiles x 17 ks
//   d3.. L7, L6, L5 (H rows only), L4, L3, L2, L1: 8 tiles x 16 ks
constexpr int NBWD = 10;
constexpr int BWD_TILES[NBWD] = {4, 8, 8, 8, 8, 8, 8, 8, 8, 8};
constexpr int BWD_KS[NBWD]    = {1, 8, 17, 16, 16, 16, 16, 16, 16, 16};
constexpr int BWD_PAD = 4;       // as FWD_PAD: puts the pair ends of d3..d9 at block 16 of their chunk
constexpr int bwd_block0(int d) {
    int b = BWD_PAD;
    for (int i = 0; i < d; ++i) b += BWD_TILES[i] * BWD_KS[i];
    return b;
}
constexpr int BWD_BLOCKS_REAL = bwd_block0(NBWD);        // 1104
constexpr int BWD_CHUNKS = (BWD_BLOCKS_REAL + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;   // 35
constexpr int BWD_BLOCKS = BWD_CHUNKS * CHUNK_BLOCKS;   // 1120 (pad and tail blocks are zero, never read)
static_assert(BWD_BLOCKS_REAL == 1104 && bwd_block0(3) % CHUNK_BLOCKS == 16, "backward stream");

// ---- bias table (fp32), one padded row per forward layer ------------------------------------
constexpr int BIAS_STRIDE = 288;                         // L8 has 257 real rows
constexpr int BIAS_FLOATS = NLAYERS * BIAS_STRIDE;       // 3168 floats = 12672 B

// ---- inference stream: the same network tiled for v_mfma_f32_16x16x32 ------------------------------
// The inference forward (nothing saved) uses the 16x16x32 shape: A = 16 output rows x 32 inputs, B = 32
// inputs x 16 samples, and a wave's 32 samples are two column tiles that share every A block.  On this
// chip that shape sustains a higher clock than 32x32x16 at equal work (measured with bare loops on
// random data: 1.76 vs 1.5 PFLOP/s with one LDS block per 32 cycles of MFMA).  Lane l = (n = l&15,
// g = l>>4).  C/D: column n, rows 4g..4g+3.  Fragment element j of k-step ks in lane group g is slot
//        slot16(ks, g, j) = 32*ks + 16*(j>>2) + 4*g + (j&3)
// so row tile rt of one layer lands in elements 4*(rt&1) .. +3 of k-step rt>>1 of the next.  Block
// (layer, row tile rt, k-step ks): lane l, element j = W[16 rt + (l&15)][feature of slot16(ks, l>>4, j)].
//                                      L0  L1  L2  L3  L4  L5  L6  L7  L8  L9 L10
constexpr int F16_TILES[NLAYERS] = {   16, 16, 16, 16, 16, 16, 16, 16, 17,  8,  1};
constexpr int F16_KS[NLAYERS]    = {    2,  8,  8,  8,  8, 10,  8,  8,  8,  9,  4};
constexpr int F16_PAD = 8;       // front pad: pair ends (16 blocks apart) fall at blocks 8 and 24 of a chunk
constexpr int f16_block0(int layer) {
    int b = F16_PAD;
    for (int l = 0; l < layer; ++l) b += F16_TILES[l] * F16_KS[l];
    return b;
}
constexpr int F16_BLOCKS_REAL = f16_block0(NLAYERS);     // 1180
constexpr int F16_CHUNKS = (F16_BLOCKS_REAL + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;   // 37
constexpr int F16_BLOCKS = F16_CHUNKS * CHUNK_BLOCKS;
static_assert(F16_BLOCKS_REAL == 1180, "inference stream");
__host__ __device__ constexpr int slot16(int ks, int g, int j) { return 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3); }
// Positional-encoding inputs of that layout: lane group g evaluates sines (g even) or cosines (g odd) of
// the lower (g < 2) or upper half of the L frequencies, 3*L/2 values, then one raw coordinate
// (g = 0, 1, 2 -> x, y, z).  q = 8*ks + j enumerates a lane's values; returns the reference feature or -1.
__host__ __device__ constexpr int pe16_feature(int L, int g, int q) {
    const int FH = L / 2, NQ = 3 * FH;
    if (q < NQ) return 3 + 6 * (FH * (g >> 1) + q / 3) + 3 * (g & 1) + q % 3;
    if (q == NQ) return g < 3 ? g : -1;
    return -1;
}

// ---- the network without view directions (use_viewdirs=False, nerf.py:151-153 and :202-206) -----------
// L0..L7 as above, then  L8  output_linear  H(256) -> 4 = (rgb_raw, sigma_raw).  Its parameters follow
// pts_linears.7 in the flat buffer.  The streams keep L0..L7 where they are and end with the one output
// layer: training forward 16 blocks (one 32-row tile x 16 k-steps), inference 8 blocks (one 16-row tile x
// 8 k-steps); the backward stream starts with dH7 = W_o^T dZ_o (8 tiles x 1 k-step) and goes on with the
// hidden layers 7..1.  Saved activations, masks and dZ use the same unit tables (the view-direction and
// colour-head tensors stay unused; dZ_o takes the place of dZ10).
constexpr int OFF_WO = OFF_WS;                  constexpr int OFF_BO = OFF_WO + 4 * 256;
constexpr int N_PARAMS_NV = OFF_BO + 4;
static_assert(N_PARAMS_NV == NERF_MLP_NPARAMS_NOVIEW, "parameter count");
constexpr int NV_FWD_BLOCKS_REAL = fwd_block0(8) + 16;                                    // 992
constexpr int NV_FWD_CHUNKS = (NV_FWD_BLOCKS_REAL + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;     // 31
constexpr int NV_F16_BLOCKS_REAL = f16_block0(8) + 8;                                     // 976
constexpr int NV_F16_CHUNKS = (NV_F16_BLOCKS_REAL + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;     // 31
constexpr int NV_NBWD = 8;        // e = 0: output layer, e = 1..7: hidden layers 7..1 (= d3..d9 above)
constexpr int NV_BWD_PAD = 8;     // puts the pair ends of e1..e7 at block 16 of their chunk
constexpr int nv_bwd_block0(int e) { return NV_BWD_PAD + (e > 0 ? 8 + (e - 1) * 128 : 0); }
constexpr int NV_BWD_BLOCKS_REAL = nv_bwd_block0(NV_NBWD);                                // 912
constexpr int NV_BWD_CHUNKS = (NV_BWD_BLOCKS_REAL + CHUNK_BLOCKS - 1) / CHUNK_BLOCKS;     // 29
static_assert(NV_FWD_BLOCKS_REAL == 992 && NV_F16_BLOCKS_REAL == 976 && NV_BWD_BLOCKS_REAL == 912 &&
              nv_bwd_block0(1) % CHUNK_BLOCKS == 16, "streams of the network without view directions");

// ---- packed buffer layout ----------------------------------------------------------------------
constexpr int64_t PACK_FWD_OFF = 0;
constexpr int64_t PACK_BWD_OFF = PACK_FWD_OFF + (int64_t)FWD_BLOCKS * BLOCK_BYTES;
constexpr int64_t PACK_BIAS_OFF = PACK_BWD_OFF + (int64_t)BWD_BLOCKS * BLOCK_BYTES;
constexpr int64_t PACK_F16_OFF = (PACK_BIAS_OFF + (int64_t)BIAS_FLOATS * 4 + 1023) / 1024 * 1024;
// fp32 copies, rounded to the operand type, of the two matrices the factored gradients above multiply
// with: W_f transposed [in 256][out 256], b_f [256] (exact), W_r0[:, :256] [128][256]
constexpr int64_t PACK_POST_OFF = PACK_F16_OFF + (int64_t)F16_BLOCKS * BLOCK_BYTES;
constexpr int POST_WF = 0, POST_BF = 256 * 256, POST_WR0 = POST_BF + 256, POST_FLOATS = POST_WR0 + 128 * 256;
// One word per workgroup of the pack launch, non-zero when that workgroup met a NaN or an infinity among the
// parameters it converted.  The forward kernels then replace the biases of the rows that become raw by NaN (see
// load_bias_table): the reference's torch.relu keeps a NaN (nerf.py:178), so there a non-finite parameter
// anywhere turns every output and the loss into NaN, whereas the kernels' integer-maximum ReLU would turn the
// negative-signed NaN an MFMA returns into 0 and hide a diverged network from the loss.
constexpr int PACK_FLAG_WORDS = 1024;
constexpr int64_t PACK_FLAGS_OFF = PACK_POST_OFF + (int64_t)POST_FLOATS * 4;
constexpr int64_t PACK_BYTES = PACK_FLAGS_OFF + (int64_t)PACK_FLAG_WORDS * 4;

// ---- slot maps --------------------------------------------------------------------------------
__host__ __device__ constexpr int slot_of(int ks, int h, int j) { return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3); }

// Positional-encoding fragment generation order.  "combo" q = 8*ks + j enumerates the
// (frequency l, component c) pairs: q = 3*l + c for q < 3*L; lane half 0 holds sin(2^l x_c),
// lane half 1 cos(2^l x_c).  Two more combos carry the raw coordinates: q = 3L -> (x, y),
// q = 3L+1 -> (z, pad).  Returns the reference feature index of (combo, half) or -1 for padding.
// Reference order (models/nerf.py:68-76): [x(3), sin f0 (3), cos f0 (3), sin f1 (3), ...].
__host__ __device__ constexpr int pe_combo_feature(int q, int h, int L) {
    if (q < 3 * L) {
        const int l = q / 3, c = q % 3;
        return 3 + 6 * l + 3 * h + c;
    }
    if (q == 3 * L) return h;            // x, y
    if (q == 3 * L + 1) return h == 0 ? 2 : -1;   // z, pad
    return -1;
}
// slot index -> reference feature (or -1) for the PE (L=10, 64 slots) and VPE (L=4, 32 slots) inputs
__host__ __device__ constexpr int pe_slot_feature(int slot, int L) {
    const int ks = slot >> 4, r = slot & 15;
    const int h = (r >> 2) & 1;
    const int j = (r & 3) | ((r >> 3) << 2);
    return pe_combo_feature(8 * ks + j, h, L);
}

// ---- saved-activation / gradient unit layout -----------------------------------------------------
// A "unit" is one k-step (16 slots) of one 32-sample wave tile: a 1 KiB block
// [32 samples][16 positions] of 16-bit values, position p = 8*h + j <-> slot(ks, h, j), i.e. lane
// l = sample + 32*h stores its 16-byte fragment at byte 32*sample + 16*h.
// Tensors are stored tensor-major: tensor t starts at unit  SAVED_KS0[t] * n_wave_tiles.
//   saved: PE(4) H0..H7(16 each) VPE(2) G(8)
// The feature vector F = W_f H7 + b_f (no activation) and its gradient dF are NOT kept: the two weight
// gradients that would need them factor through one product over samples, P = dZ9^T H7 [128 x 256]:
//   d rgb_layers.0.weight[:, :256] = sum_s dZ9 F^T = P W_f^T + (sum_s dZ9) b_f^T
//   d feature_linear.weight        = sum_s dF H7^T = W_r0[:, :256]^T P,     d feature_linear.bias = W_r0[:, :256]^T sum_s dZ9
// (weights-only products, done after the reduction), which saves 32 KiB per 32 samples of stores and
// as much of loads.
enum { SV_PE = 0, SV_H0 = 1, SV_VPE = 9, SV_G = 10, SV_NT = 11 };
constexpr int SAVED_KS[SV_NT] = {4, 16, 16, 16, 16, 16, 16, 16, 16, 2, 8};
constexpr int saved_ks0(int t) { int k = 0; for (int i = 0; i < t; ++i) k += SAVED_KS[i]; return k; }
constexpr int SAVED_UNITS = saved_ks0(SV_NT);   // 142 KiB per 32 samples
//   gradients dZ of each layer's pre-activation: L0..L7 (16 each), L8 (1: sigma only, see above), L9 (8), L10 (1)
constexpr int DZ_KS[NLAYERS] = {16, 16, 16, 16, 16, 16, 16, 16, 1, 8, 1};
constexpr int dz_ks0(int l) { int k = 0; for (int i = 0; i < l; ++i) k += DZ_KS[i]; return k; }
constexpr int DZ_UNITS = dz_ks0(NLAYERS);       // 138 KiB per 32 samples

// ---- ReLU sign masks -----------------------------------------------------------------------------
// The backward-data chain needs only the sign of each post-ReLU activation.  Behind the saved units
// the forward pass therefore keeps, per wave tile, MASK_TENSORS KiB of bit masks: tensor t (H0..H7 =
// 0..7, G = 8) of wave tile wt is the KiB at index wt * MASK_TENSORS + t.  Inside it the dword at
// byte 256*p + 4*lane holds, for that lane, the signs of output tiles 2p and 2p+1: accumulator
// element r of tile 2p+par is bit mask_bit(par, r).  (The bit order is what falls out of reducing
// the packed 16-bit pairs two at a time.)
constexpr int MASK_TENSORS = 9;
constexpr int MASK_G = 8;
constexpr int mask_bit(int par, int r) { return 8 * par + 16 * (r & 1) + (r >> 1); }
constexpr int64_t saved_bytes_per_wave_tile() { return (int64_t)(SAVED_UNITS + MASK_TENSORS) * 1024; }

// ---- operand types -------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PREC> struct Prec;
template <> struct Prec<NERF_PREC_BF16> {
    typedef __bf16 elem;
    typedef __bf16 frag __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Prec<NERF_PREC_FP16> {
    typedef _Float16 elem;
    typedef _Float16 frag __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

}  // namespace mlp
}  // namespace nerf
