// Per-ray alpha compositing, forward and backward (src/nerf_project/models/nerf.py:430-472).
//
// One wavefront owns one ray.  Sample k of a 64-sample chunk sits on lane k, so
// the 16-byte (rgb, sigma) records of a chunk are one coalesced 1 KiB load.
// The transmittance T_k = prod_{j<k} (1 - alpha_j + 1e-10) is an exclusive
// product scan across the wave, carried from chunk to chunk; it is evaluated
// in fp64 and rounded to fp32 per element, which is what the reference's CPU
// cumprod does.  The backward recomputes the forward quantities from (raw, z)
// and needs one reverse (suffix) sum scan of G_j * w_j.
// Compiled with -ffp-contract=off.
#include "common.hpp"

namespace nerf {

struct SampleFwd {
    float alpha, T, w, q, de;  // de = dist * exp(-sigma*dist) = d alpha / d sigma
    float cr, cg, cb;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// forward quantities of one chunk; `carry` is the running product of q over earlier chunks
__device__ __forceinline__ SampleFwd chunk_forward(const float4 r, float zk, float zn, bool valid, bool last,
                                                   int lane, double& carry) {
    SampleFwd s;
    const float dist = last ? 1e10f : zn - zk;
    const float sig = fmaxf(r.w, 0.0f);
    const float e = expf(-sig * dist);
    s.alpha = valid ? 1.0f - e : 0.0f;
    s.de = valid ? dist * e : 0.0f;
    s.q = (1.0f - s.alpha) + 1e-10f;
    const double incl = wave_incl_prod(valid ? (double)s.q : 1.0, lane);
    double excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0;
    s.T = (float)(carry * excl);
    carry = carry * __shfl(incl, 63, 64);
    s.w = s.alpha * s.T;
    s.cr = sigmoidf_(r.x); s.cg = sigmoidf_(r.y); s.cb = sigmoidf_(r.z);
    return s;
}

__device__ __forceinline__ float disp_of(float depth, float acc) {
    float d = 1.0f / fmaxf(1e-10f, depth / (acc + 1e-10f));
    if (isnan(d)) d = 0.0f;                       // torch.nan_to_num(nan=0.0)
    else if (isinf(d)) d = d > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    return d;
}

// NCH > 0: chunks unrolled (S <= 64*NCH); NCH == 0: run-time chunk loop.
template <int NCH>
__global__ __launch_bounds__(256) void composite_fwd_kernel(int64_t n_rays, int S, const float4* __restrict__ raw,
                                                            const float* __restrict__ z, int white,
                                                            float* __restrict__ rgb_map,
                                                            float* __restrict__ depth_map,
                                                            float* __restrict__ disp_map,
                                                            float* __restrict__ acc_map,
                                                            float* __restrict__ weights,
                                                            float* __restrict__ alpha_out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int nch = NCH > 0 ? NCH : (S + 63) >> 6;
    for (int64_t ray = wave0; ray < n_rays; ray += nwaves) {
        const int64_t base = ray * S;
        double carry = 1.0;
        float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
        float4 r[NCH > 0 ? NCH : 1];
        float zk[NCH > 0 ? NCH : 1], zn[NCH > 0 ? NCH : 1];
        if (NCH > 0) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int k = c * 64 + lane;
                r[c] = k < S ? raw[base + k] : make_float4(0.f, 0.f, 0.f, 0.f);
                zk[c] = k < S ? z[base + k] : 0.f;
                zn[c] = k + 1 < S ? z[base + k + 1] : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < nch; ++c) {
            const int k = c * 64 + lane;
            const bool valid = k < S;
            float4 rc; float zkc, znc;
            if (NCH > 0) { rc = r[c]; zkc = zk[c]; znc = zn[c]; }
            else {
                rc = valid ? raw[base + k] : make_float4(0.f, 0.f, 0.f, 0.f);
                zkc = valid ? z[base + k] : 0.f;
                znc = k + 1 < S ? z[base + k + 1] : 0.f;
            }
            const SampleFwd s = chunk_forward(rc, zkc, znc, valid, k == S - 1, lane, carry);
            if (valid) {
                weights[base + k] = s.w;
                if (alpha_out) alpha_out[base + k] = s.alpha;
            }
            sr += s.w * s.cr; sg += s.w * s.cg; sb += s.w * s.cb;
            sd += s.w * zkc; sa += s.w;
        }
        sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sa = wave_sum(sa);
        if (lane == 0) {
            const float bg = white ? 1.0f - sa : 0.0f;
            rgb_map[3 * ray + 0] = sr + bg; rgb_map[3 * ray + 1] = sg + bg; rgb_map[3 * ray + 2] = sb + bg;
            depth_map[ray] = sd; acc_map[ray] = sa; disp_map[ray] = disp_of(sd, sa);
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void composite_bwd_kernel(int64_t n_rays, int S, const float4* __restrict__ raw,
                                                            const float* __restrict__ z, int white,
                                                            const float* __restrict__ g_rgb,
                                                            const float* __restrict__ g_depth,
                                                            const float* __restrict__ g_disp,
                                                            const float* __restrict__ g_acc,
                                                            const float* __restrict__ g_w,
                                                            const float* __restrict__ g_alpha,
                                                            float4* __restrict__ d_raw) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t ray = wave0; ray < n_rays; ray += nwaves) {
        const int64_t base = ray * S;
        SampleFwd s[NCH];
        float zk[NCH], rawsig[NCH];
        double carry = 1.0;
        float sd = 0.f, sa = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = c * 64 + lane;
            const bool valid = k < S;
            const float4 rc = valid ? raw[base + k] : make_float4(0.f, 0.f, 0.f, 0.f);
            zk[c] = valid ? z[base + k] : 0.f;
            const float zn = k + 1 < S ? z[base + k + 1] : 0.f;
            rawsig[c] = rc.w;
            s[c] = chunk_forward(rc, zk[c], zn, valid, k == S - 1, lane, carry);
            sd += s[c].w * zk[c]; sa += s[c].w;
        }
        float gr = 0.f, gg = 0.f, gb = 0.f;
        if (g_rgb) { gr = g_rgb[3 * ray]; gg = g_rgb[3 * ray + 1]; gb = g_rgb[3 * ray + 2]; }
        float gD = g_depth ? g_depth[ray] : 0.f;
        float gA = g_acc ? g_acc[ray] : 0.f;
        if (white) gA -= (gr + gg) + gb;
        if (g_disp) {
            sd = wave_sum(sd); sa = wave_sum(sa);
            const float den = sa + 1e-10f;
            const float x = sd / den;
            if (x > 1e-10f && !isnan(x)) {   // max() selects x; disp = 1/x
                const float gx = -g_disp[ray] / (x * x);
                gD += gx / den;
                gA += -gx * sd / (den * den);
            }
        }
        float suffix_carry = 0.f;  // sum of G_j w_j over later chunks
#pragma unroll
        for (int c = NCH - 1; c >= 0; --c) {
            const int k = c * 64 + lane;
            const bool valid = k < S;
            const SampleFwd& v = s[c];
            float G = ((gr * v.cr + gg * v.cg) + gb * v.cb) + gD * zk[c] + gA;
            if (g_w && valid) G += g_w[base + k];
            const float gw = valid ? G * v.w : 0.f;
            const float incl = wave_incl_suffix_sum(gw, lane);
            const float after = (incl - gw) + suffix_carry;   // sum_{j>k} G_j w_j
            suffix_carry += __shfl(incl, 0, 64);
            float dalpha = G * v.T - after / v.q;
            if (g_alpha && valid) dalpha += g_alpha[base + k];
            const float dsig = rawsig[c] > 0.0f ? dalpha * v.de : 0.0f;
            if (valid) {
                float4 o;
                o.x = v.w * gr * (v.cr * (1.0f - v.cr));
                o.y = v.w * gg * (v.cg * (1.0f - v.cg));
                o.z = v.w * gb * (v.cb * (1.0f - v.cb));
                o.w = dsig;
                d_raw[base + k] = o;
            }
        }
    }
}

static inline unsigned ray_grid(int64_t n_rays) {
    int64_t blocks = (n_rays + 3) / 4;  // 4 waves (rays) per 256-thread block
    if (blocks > 16384) blocks = 16384;
    return (unsigned)blocks;
}

}  // namespace nerf

extern "C" int nerf_composite_fwd(int64_t n_rays, int S, const float* raw, const float* z, int white_bkgd,
                                  float* rgb_map, float* depth_map, float* disp_map, float* acc_map,
                                  float* weights, float* alpha, void* stream) {
    NERF_REQUIRE(n_rays >= 0 && S >= 2 && S <= 1024, NERF_E_SHAPE);
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(raw && z && rgb_map && depth_map && disp_map && acc_map && weights, NERF_E_NULL);
    NERF_REQUIRE(nerf_aligned(raw, 16), NERF_E_ALIGN);
    if (n_rays == 0) return NERF_OK;
    const dim3 grid(nerf::ray_grid(n_rays)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const float4* r4 = reinterpret_cast<const float4*>(raw);
#define NERF_FWD(N)                                                                                        \
    hipLaunchKernelGGL(nerf::composite_fwd_kernel<N>, grid, block, 0, st, n_rays, S, r4, z, white_bkgd,    \
                       rgb_map, depth_map, disp_map, acc_map, weights, alpha)
    const int nch = (S + 63) / 64;
    if (nch == 1) NERF_FWD(1);
    else if (nch == 2) NERF_FWD(2);
    else if (nch == 3) NERF_FWD(3);
    else if (nch == 4) NERF_FWD(4);
    else NERF_FWD(0);
#undef NERF_FWD
    return nerf_launch_status();
}

extern "C" int nerf_composite_bwd(int64_t n_rays, int S, const float* raw, const float* z, int white_bkgd,
                                  const float* g_rgb, const float* g_depth, const float* g_disp,
                                  const float* g_acc, const float* g_weights, const float* g_alpha,
                                  float* d_raw, void* stream) {
    NERF_REQUIRE(n_rays >= 0 && S >= 2 && S <= 1024, NERF_E_SHAPE);
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(raw && z && d_raw, NERF_E_NULL);
    NERF_REQUIRE(nerf_aligned(raw, 16) && nerf_aligned(d_raw, 16), NERF_E_ALIGN);
    if (n_rays == 0) return NERF_OK;
    const dim3 grid(nerf::ray_grid(n_rays)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const float4* r4 = reinterpret_cast<const float4*>(raw);
    float4* d4 = reinterpret_cast<float4*>(d_raw);
#define NERF_BWD(N)                                                                                        \
    hipLaunchKernelGGL(nerf::composite_bwd_kernel<N>, grid, block, 0, st, n_rays, S, r4, z, white_bkgd,    \
                       g_rgb, g_depth, g_disp, g_acc, g_weights, g_alpha, d4)
    const int nch = (S + 63) / 64;
    if (nch == 1) NERF_BWD(1);
    else if (nch == 2) NERF_BWD(2);
    else if (nch == 3) NERF_BWD(3);
    else if (nch == 4) NERF_BWD(4);
    else if (nch <= 8) NERF_BWD(8);
    else NERF_BWD(16);
#undef NERF_BWD
    return nerf_launch_status();
}
