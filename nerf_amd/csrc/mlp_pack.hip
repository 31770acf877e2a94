// Weight packing: flat fp32 parameters (state_dict order) -> the MFMA fragment streams and bias
// table described in mlp_common.hpp.  Runs once per weight update (about 3.5 MB of traffic).
#include "mlp_common.hpp"

namespace nerf {
namespace mlp {

__device__ __forceinline__ float fwd_weight(const float* __restrict__ P, int L, int row, int ks, int slot) {
    switch (L) {
        case 0: {
            const int f = pe_slot_feature(slot, 10);
            return f >= 0 ? P[OFF_W0 + row * 63 + f] : 0.0f;
        }
        case 1: return P[OFF_W1 + row * 256 + slot];
        case 2: return P[OFF_W2 + row * 256 + slot];
        case 3: return P[OFF_W3 + row * 256 + slot];
        case 4: return P[OFF_W4 + row * 256 + slot];
        case 5: {
            if (ks < PE_KS) {
                const int f = pe_slot_feature(slot, 10);
                return f >= 0 ? P[OFF_W5 + row * 319 + f] : 0.0f;
            }
            return P[OFF_W5 + row * 319 + 63 + (slot - 64)];
        }
        case 6: return P[OFF_W6 + row * 256 + slot];
        case 7: return P[OFF_W7 + row * 256 + slot];
        case 8:
            if (row < 256) return P[OFF_WF + row * 256 + slot];
            return row == 256 ? P[OFF_WS + slot] : 0.0f;
        case 9: {
            if (row >= 128) return 0.0f;
            if (ks < H_KS) return P[OFF_WR0 + row * 283 + slot];
            const int f = pe_slot_feature(slot - 256, 4);
            return f >= 0 ? P[OFF_WR0 + row * 283 + 256 + f] : 0.0f;
        }
        default: return row < 3 ? P[OFF_WR2 + row * 128 + slot] : 0.0f;
    }
}

// inference stream: weight of output `row` and the input feature that slot16 / pe16_feature assign to
// (k-step ks, lane group g, element j)
__device__ __forceinline__ float f16_weight(const float* __restrict__ P, int L, int row, int ks, int g, int j) {
    const int slot = slot16(ks, g, j);
    const int q = 8 * ks + j;
    switch (L) {
        case 0: {
            const int f = pe16_feature(10, g, q);
            return f >= 0 ? P[OFF_W0 + row * 63 + f] : 0.0f;
        }
        case 1: return P[OFF_W1 + row * 256 + slot];
        case 2: return P[OFF_W2 + row * 256 + slot];
        case 3: return P[OFF_W3 + row * 256 + slot];
        case 4: return P[OFF_W4 + row * 256 + slot];
        case 5: {
            if (ks < 2) {
                const int f = pe16_feature(10, g, q);
                return f >= 0 ? P[OFF_W5 + row * 319 + f] : 0.0f;
            }
            return P[OFF_W5 + row * 319 + 63 + (slot - 64)];
        }
        case 6: return P[OFF_W6 + row * 256 + slot];
        case 7: return P[OFF_W7 + row * 256 + slot];
        case 8:
            if (row < 256) return P[OFF_WF + row * 256 + slot];
            return row == 256 ? P[OFF_WS + slot] : 0.0f;
        case 9: {
            if (ks < 8) return P[OFF_WR0 + row * 283 + slot];
            const int f = pe16_feature(4, g, j);
            return f >= 0 ? P[OFF_WR0 + row * 283 + 256 + f] : 0.0f;
        }
        default: return row < 3 ? P[OFF_WR2 + row * 128 + slot] : 0.0f;
    }
}

// A = W^T: row = input feature, k = output slot
__device__ __forceinline__ float bwd_weight(const float* __restrict__ P, int d, int rin, int k) {
    switch (d) {
        case 0: return (rin < 128 && k < 3) ? P[OFF_WR2 + k * 128 + rin] : 0.0f;
        case 1: return k < 128 ? P[OFF_WR0 + k * 283 + rin] : 0.0f;
        case 2:
            if (k < 256) return P[OFF_WF + k * 256 + rin];
            return k == 256 ? P[OFF_WS + rin] : 0.0f;
        case 3: return P[OFF_W7 + k * 256 + rin];
        case 4: return P[OFF_W6 + k * 256 + rin];
        case 5: return P[OFF_W5 + k * 319 + 63 + rin];
        case 6: return P[OFF_W4 + k * 256 + rin];
        case 7: return P[OFF_W3 + k * 256 + rin];
        case 8: return P[OFF_W2 + k * 256 + rin];
        default: return P[OFF_W1 + k * 256 + rin];
    }
}

__device__ __forceinline__ float bias_value(const float* __restrict__ P, int L, int row) {
    if (L < 8) {
        const int off[8] = {OFF_B0, OFF_B1, OFF_B2, OFF_B3, OFF_B4, OFF_B5, OFF_B6, OFF_B7};
        return row < 256 ? P[off[L] + row] : 0.0f;
    }
    if (L == 8) return row < 256 ? P[OFF_BF + row] : (row == 256 ? P[OFF_BS] : 0.0f);
    if (L == 9) return row < 128 ? P[OFF_BR0 + row] : 0.0f;
    return row < 3 ? P[OFF_BR2 + row] : 0.0f;
}

// One thread produces one lane's 16-byte fragment of one block (or one fp32 of the bias / post tables).
// `streams`: NERF_PACK_TRAIN = forward + backward streams and the post table, NERF_PACK_INFER = the
// inference stream; the bias table is always written.
// ---- the network without view directions: L0..L7 from the functions above, then output_linear ----
__device__ __forceinline__ float nv_out_weight(const float* __restrict__ P, int row, int slot) {
    return row < 4 ? P[OFF_WO + row * 256 + slot] : 0.0f;
}

struct PackArgs {
    const float* params[NERF_WGRAD_MAX_NETS];
    char* out[NERF_WGRAD_MAX_NETS];
    int streams[NERF_WGRAD_MAX_NETS];
};

// blockIdx.y = network
template <int PREC>
__global__ __launch_bounds__(256) void pack_kernel(const PackArgs args) {
    const float* __restrict__ P = args.params[blockIdx.y];
    char* __restrict__ out = args.out[blockIdx.y];
    const int streams = args.streams[blockIdx.y];
    const bool noview = streams & NERF_PACK_NOVIEW;
    typedef typename Prec<PREC>::elem T;
    typedef typename Prec<PREC>::frag frag;
    frag* fwd = reinterpret_cast<frag*>(out + PACK_FWD_OFF);
    frag* bwd = reinterpret_cast<frag*>(out + PACK_BWD_OFF);
    frag* f16 = reinterpret_cast<frag*>(out + PACK_F16_OFF);
    float* bias = reinterpret_cast<float*>(out + PACK_BIAS_OFF);
    float* post = reinterpret_cast<float*>(out + PACK_POST_OFF);
    constexpr int64_t NF = (int64_t)FWD_BLOCKS * 64, NB = (int64_t)BWD_BLOCKS * 64, N16 = (int64_t)F16_BLOCKS * 64;
    constexpr int64_t E_BWD = NF, E_BIAS = NF + NB, E_F16 = E_BIAS + BIAS_FLOATS, E_POST = E_F16 + N16;
    constexpr int64_t total = E_POST + POST_FLOATS;
    const bool train = streams & NERF_PACK_TRAIN, infer = streams & NERF_PACK_INFER;
    // a NaN or an infinity among the values this thread converts (every parameter passes through the forward or
    // the inference stream and the bias table): reported per workgroup in the flag words behind the streams
    bool bad = false;
    auto seen = [&](float w) { bad |= !(fabsf(w) <= 3.4028234663852886e38f); return w; };
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        if (e < E_BWD) {
            if (!train) continue;
            int b = (int)(e >> 6);
            const int lane = (int)e & 63;
            frag f;
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (T)0.0f;
            if (noview && b >= fwd_block0(8)) {
                const int ks = b - fwd_block0(8);             // one 32-row tile, 16 k-steps in order
                if (ks < 16) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = (T)seen(nv_out_weight(P, lane & 31, slot_of(ks, lane >> 5, j)));
                }
            } else if (b >= FWD_PAD && b < FWD_BLOCKS_REAL) {
                b -= FWD_PAD;
                int L = 0;
                while (L + 1 < NLAYERS && b >= FWD_TILES[L] * FWD_KS[L]) { b -= FWD_TILES[L] * FWD_KS[L]; ++L; }
                const int m = stream_tile(FWD_TILES[L], FWD_KS[L], b), ks = stream_ks(FWD_TILES[L], FWD_KS[L], b);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    f[j] = (T)seen(fwd_weight(P, L, 32 * m + (lane & 31), ks, slot_of(ks, lane >> 5, j)));
            }
            fwd[e] = f;
        } else if (e < E_BIAS) {
            if (!train) continue;
            const int64_t eb = e - E_BWD;
            int b = (int)(eb >> 6);
            const int lane = (int)eb & 63;
            frag f;
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (T)0.0f;
            if (noview) {
                if (b >= NV_BWD_PAD && b < NV_BWD_BLOCKS_REAL) {
                    b -= NV_BWD_PAD;
                    if (b < 8) {          // dH7 = W_o^T dZ_o: tile b, one k-step, slots 0..3 = the four outputs
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int k = slot_of(0, lane >> 5, j);
                            f[j] = (T)(k < 4 ? P[OFF_WO + k * 256 + 32 * b + (lane & 31)] : 0.0f);
                        }
                    } else {              // hidden layers 7..1 = d3..d9 of the full network
                        b -= 8;
                        const int d = 3 + b / 128;
                        b %= 128;
                        const int m = stream_tile(8, 16, b), ks = stream_ks(8, 16, b);
#pragma unroll
                        for (int j = 0; j < 8; ++j) f[j] = (T)bwd_weight(P, d, 32 * m + (lane & 31), slot_of(ks, lane >> 5, j));
                    }
                }
            } else if (b >= BWD_PAD && b < BWD_BLOCKS_REAL) {
                b -= BWD_PAD;
                int d = 0;
                while (d + 1 < NBWD && b >= BWD_TILES[d] * BWD_KS[d]) { b -= BWD_TILES[d] * BWD_KS[d]; ++d; }
                const int m = stream_tile(BWD_TILES[d], BWD_KS[d], b), ks = stream_ks(BWD_TILES[d], BWD_KS[d], b);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (T)bwd_weight(P, d, 32 * m + (lane & 31), slot_of(ks, lane >> 5, j));
            }
            bwd[eb] = f;
        } else if (e < E_F16) {
            const int i = (int)(e - E_BIAS);
            const int L = i / BIAS_STRIDE, row = i % BIAS_STRIDE;
            bias[i] = seen((noview && L >= 8) ? ((L == 8 && row < 4) ? P[OFF_BO + row] : 0.0f) : bias_value(P, L, row));
        } else if (e < E_POST) {
            if (!infer) continue;
            const int64_t ef = e - E_F16;
            int b = (int)(ef >> 6);
            const int lane = (int)ef & 63;
            frag f;
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (T)0.0f;
            if (noview && b >= f16_block0(8)) {
                const int ks = b - f16_block0(8);             // one 16-row tile, 8 k-steps in order
                if (ks < 8 && (lane & 15) < 4) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = (T)seen(nv_out_weight(P, lane & 15, slot16(ks, lane >> 4, j)));
                }
            } else if (b >= F16_PAD && b < F16_BLOCKS_REAL) {
                b -= F16_PAD;
                int L = 0;
                while (L + 1 < NLAYERS && b >= F16_TILES[L] * F16_KS[L]) { b -= F16_TILES[L] * F16_KS[L]; ++L; }
                const int rt = stream_tile(F16_TILES[L], F16_KS[L], b), ks = stream_ks(F16_TILES[L], F16_KS[L], b);
                const int row = 16 * rt + (lane & 15);
                const int rows = L < 8 ? 256 : (L == 8 ? 257 : (L == 9 ? 128 : 3));
                if (row < rows) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = (T)seen(f16_weight(P, L, row, ks, lane >> 4, j));
                }
            }
            f16[ef] = f;
        } else {
            if (!train || noview) continue;
            const int i = (int)(e - E_POST);
            float v;
            if (i < POST_BF) { const int in = i >> 8, o = i & 255; v = (float)(T)P[OFF_WF + o * 256 + in]; }   // W_f transposed
            else if (i < POST_WR0) v = P[OFF_BF + (i - POST_BF)];
            else { const int k = (i - POST_WR0) >> 8, o = (i - POST_WR0) & 255; v = (float)(T)P[OFF_WR0 + k * 283 + o]; }
            post[i] = v;
        }
    }
    const int any = __syncthreads_or(bad);
    if (threadIdx.x == 0) reinterpret_cast<uint32_t*>(out + PACK_FLAGS_OFF)[blockIdx.x] = any ? 1u : 0u;
}

}  // namespace mlp
}  // namespace nerf

extern "C" int64_t nerf_mlp_packed_bytes(void) { return nerf::mlp::PACK_BYTES; }

extern "C" int nerf_mlp_pack_multi(int n_nets, const float* const* params, int precision, const int* streams,
                                   void* const* packed, void* stream) {
    NERF_REQUIRE(params && streams && packed, NERF_E_NULL);
    NERF_REQUIRE(n_nets >= 1 && n_nets <= NERF_WGRAD_MAX_NETS, NERF_E_SHAPE);
    NERF_REQUIRE(precision == NERF_PREC_BF16 || precision == NERF_PREC_FP16, NERF_E_DTYPE);
    nerf::mlp::PackArgs a{};
    for (int n = 0; n < n_nets; ++n) {
        NERF_REQUIRE(params[n] && packed[n], NERF_E_NULL);
        NERF_REQUIRE((streams[n] & (NERF_PACK_TRAIN | NERF_PACK_INFER)) != 0 &&
                     (streams[n] & ~(NERF_PACK_TRAIN | NERF_PACK_INFER | NERF_PACK_NOVIEW)) == 0, NERF_E_SHAPE);
        NERF_REQUIRE(nerf_aligned(packed[n], 256), NERF_E_ALIGN);
        a.params[n] = params[n]; a.out[n] = (char*)packed[n]; a.streams[n] = streams[n];
    }
    const dim3 grid(nerf::mlp::PACK_FLAG_WORDS, n_nets), block(256);      // one flag word per workgroup
    if (precision == NERF_PREC_BF16)
        hipLaunchKernelGGL(nerf::mlp::pack_kernel<NERF_PREC_BF16>, grid, block, 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(nerf::mlp::pack_kernel<NERF_PREC_FP16>, grid, block, 0, (hipStream_t)stream, a);
    return nerf_launch_status();
}

extern "C" int nerf_mlp_pack(const float* params, int precision, int streams, void* packed, void* stream) {
    if (precision & NERF_ARCH_NOVIEW) { precision &= ~NERF_ARCH_NOVIEW; streams |= NERF_PACK_NOVIEW; }
    return nerf_mlp_pack_multi(1, &params, precision, &streams, &packed, stream);
}
