// Per-ray alpha compositing, forward and backward (src/nerf_project/models/nerf.py:430-472).
//
// One wavefront owns one ray.  Sample k of a 64-sample chunk sits on lane k, so
// the 16-byte (rgb, sigma) records of a chunk are one coalesced 1 KiB load.
// The transmittance T_k = prod_{j<k} (1 - alpha_j + 1e-10) is an exclusive
// product scan across the wave, carried from chunk to chunk; it is evaluated
// in fp64 and rounded to fp32 per element, which is what the reference's CPU
// cumprod does.  The backward recomputes the forward quantities from (raw, z)
// and needs one reverse (suffix) sum scan of G_j * w_j.  All scans and sums move
// data between lanes with DPP modifiers on the VALU (nothing goes through LDS);
// the forward is bound by HBM once exp and sigmoid use the hardware exp2 / rcp.
// Compiled with -ffp-contract=off.
#include "common.hpp"

namespace nerf {

struct SampleFwd {
    float alpha, T, w, q, de;  // de = dist * exp(-sigma*dist) = d alpha / d sigma
    float cr, cg, cb;
};

// exp and sigmoid on the hardware exp2 / rcp units (1 ulp each).  The argument's rounding after the scaling
// by log2(e) adds a relative error of about 1e-7 * |x|, which exp(x) <= 1 and s(1-s) keep below 1e-7
// absolute in alpha and in the colours.
__device__ __forceinline__ float exp_hw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + exp_hw(-x)); }

// forward quantities of one chunk; `carry` is the running product of q over earlier chunks
__device__ __forceinline__ SampleFwd chunk_forward(const float4 r, float zk, float zn, bool valid, bool last,
                                                   double& carry) {
    SampleFwd s;
    const float dist = last ? 1e10f : zn - zk;
    const float sig = r.w < 0.0f ? 0.0f : r.w;          // relu that keeps a NaN a NaN, as torch.relu does (nerf.py:431)
    const float e = exp_hw(-sig * dist);
    s.alpha = valid ? 1.0f - e : 0.0f;
    s.de = valid ? dist * e : 0.0f;
    s.q = (1.0f - s.alpha) + 1e-10f;
    const double incl = wave_incl_prod(valid ? (double)s.q : 1.0);
    const double excl = dpp_move<0x138>(1.0, incl);        // lane k-1's product; 1 on lane 0
    s.T = (float)(carry * excl);
    carry = carry * read_lane(incl, 63);
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

// The (rgb, sigma) records and depths of one ray, sample 64c + lane in element c.
template <int NCH>
struct RayRecords {
    float4 r[NCH];
    float z[NCH];
};

// Every lane always issues its loads (lanes past the last sample re-read it), so the number of loads in
// flight is fixed and the waits on them can be counted.  Records are read once and results written once:
// both go past the caches as streaming (non-temporal) accesses.
template <int NCH>
__device__ __forceinline__ void load_ray(RayRecords<NCH>& in, const float4* __restrict__ raw, const float* __restrict__ z,
                                         int64_t base, int S, int lane) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k = min(c * 64 + lane, S - 1);
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(raw + base + k));
        in.r[c] = make_float4(v.x, v.y, v.z, v.w);
        in.z[c] = __builtin_nontemporal_load(z + base + k);
    }
}

struct RaySums {
    float r, g, b, d, a;
};

__device__ __forceinline__ void store_maps(const RaySums& t, int64_t ray, int white, int lane, float* __restrict__ rgb_map,
                                           float* __restrict__ depth_map, float* __restrict__ disp_map,
                                           float* __restrict__ acc_map) {
    const float sr = wave_sum(t.r), sg = wave_sum(t.g), sb = wave_sum(t.b), sd = wave_sum(t.d), sa = wave_sum(t.a);
    if (lane == 0) {
        const float bg = white ? 1.0f - sa : 0.0f;
        rgb_map[3 * ray + 0] = sr + bg; rgb_map[3 * ray + 1] = sg + bg; rgb_map[3 * ray + 2] = sb + bg;
        depth_map[ray] = sd; acc_map[ray] = sa; disp_map[ray] = disp_of(sd, sa);
    }
}

// S <= 64*NCH.  Each wave walks its rays with the next ray's records already in flight while the current
// one is scanned, so a resident wave always has loads outstanding; the grid is sized to one full set of
// resident waves (see ray_grid_fwd).
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
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    if (wave0 >= n_rays) return;
    RayRecords<NCH> cur, nxt;
    load_ray(cur, raw, z, wave0 * S, S, lane);
    for (int64_t ray = wave0; ray < n_rays; ray += nwaves) {
        const int64_t base = ray * S;
        load_ray(nxt, raw, z, min(ray + nwaves, n_rays - 1) * S, S, lane);   // the last round re-reads a ray
        double carry = 1.0;
        RaySums t = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = c * 64 + lane;
            const bool valid = k < S;
            // depth of sample k+1: the lane above, or lane 0 of the next chunk
            float zn = dpp_move<0x130>(0.0f, cur.z[c]);
            if (c + 1 < NCH) { const float first = read_lane(cur.z[c + 1], 0); if (lane == 63) zn = first; }
            const SampleFwd s = chunk_forward(cur.r[c], cur.z[c], zn, valid, k == S - 1, carry);
            if (valid) {
                __builtin_nontemporal_store(s.w, weights + base + k);
                if (alpha_out) __builtin_nontemporal_store(s.alpha, alpha_out + base + k);
            }
            t.r += s.w * s.cr; t.g += s.w * s.cg; t.b += s.w * s.cb;
            t.d += s.w * cur.z[c]; t.a += s.w;
        }
        store_maps(t, ray, white, lane, rgb_map, depth_map, disp_map, acc_map);
        cur = nxt;
    }
}

// Any S: run-time chunk loop, one ray per wave at a time.
__global__ __launch_bounds__(256) void composite_fwd_long_kernel(int64_t n_rays, int S, const float4* __restrict__ raw,
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
    const int nch = (S + 63) >> 6;
    for (int64_t ray = wave0; ray < n_rays; ray += nwaves) {
        const int64_t base = ray * S;
        double carry = 1.0;
        RaySums t = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nch; ++c) {
            const int k = c * 64 + lane;
            const bool valid = k < S;
            const float4 rc = valid ? raw[base + k] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float zk = valid ? z[base + k] : 0.f;
            const float zn = k + 1 < S ? z[base + k + 1] : 0.f;
            const SampleFwd s = chunk_forward(rc, zk, zn, valid, k == S - 1, carry);
            if (valid) {
                weights[base + k] = s.w;
                if (alpha_out) alpha_out[base + k] = s.alpha;
            }
            t.r += s.w * s.cr; t.g += s.w * s.cg; t.b += s.w * s.cb;
            t.d += s.w * zk; t.a += s.w;
        }
        store_maps(t, ray, white, lane, rgb_map, depth_map, disp_map, acc_map);
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
            s[c] = chunk_forward(rc, zk[c], zn, valid, k == S - 1, carry);
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
            suffix_carry += read_lane(incl, 0);
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

// Any S: the backward with a run-time chunk loop.  Pass 1 walks the ray front to back and leaves the running
// transmittance product in front of every chunk in that chunk's first d_raw record (this wave's own output
// rows, not yet written); pass 2 walks back to front, takes the product from there, recomputes the chunk and
// overwrites the record with its result.  Same arithmetic and order of summation as the unrolled kernel.
__global__ __launch_bounds__(256) void composite_bwd_long_kernel(int64_t n_rays, int S, const float4* __restrict__ raw,
                                                                 const float* __restrict__ z, int white,
                                                                 const float* __restrict__ g_rgb,
                                                                 const float* __restrict__ g_depth,
                                                                 const float* __restrict__ g_disp,
                                                                 const float* __restrict__ g_acc,
                                                                 const float* __restrict__ g_w,
                                                                 const float* __restrict__ g_alpha,
                                                                 float4* d_raw) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int nch = (S + 63) >> 6;
    for (int64_t ray = wave0; ray < n_rays; ray += nwaves) {
        const int64_t base = ray * S;
        double carry = 1.0;
        float sd = 0.f, sa = 0.f;
        for (int c = 0; c < nch; ++c) {
            const int k = c * 64 + lane;
            const bool valid = k < S;
            if (lane == 0) {
                const long long bits = __builtin_bit_cast(long long, carry);
                d_raw[base + k] = make_float4(__builtin_bit_cast(float, (int)bits), __builtin_bit_cast(float, (int)(bits >> 32)), 0.f, 0.f);
            }
            const float4 rc = valid ? raw[base + k] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float zk = valid ? z[base + k] : 0.f;
            const float zn = k + 1 < S ? z[base + k + 1] : 0.f;
            const SampleFwd s = chunk_forward(rc, zk, zn, valid, k == S - 1, carry);
            sd += s.w * zk; sa += s.w;
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
            if (x > 1e-10f && !isnan(x)) {
                const float gx = -g_disp[ray] / (x * x);
                gD += gx / den;
                gA += -gx * sd / (den * den);
            }
        }
        // the products written above are read back by the lane that wrote them, once its stores have completed
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        float suffix_carry = 0.f;
        for (int c = nch - 1; c >= 0; --c) {
            const int k = c * 64 + lane;
            const bool valid = k < S;
            float4 head = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane == 0) head = d_raw[base + k];
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, head.x));
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, head.y));
            double cprod = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
            const float4 rc = valid ? raw[base + k] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float zk = valid ? z[base + k] : 0.f;
            const float zn = k + 1 < S ? z[base + k + 1] : 0.f;
            const SampleFwd v = chunk_forward(rc, zk, zn, valid, k == S - 1, cprod);
            float G = ((gr * v.cr + gg * v.cg) + gb * v.cb) + gD * zk + gA;
            if (g_w && valid) G += g_w[base + k];
            const float gw = valid ? G * v.w : 0.f;
            const float incl = wave_incl_suffix_sum(gw, lane);
            const float after = (incl - gw) + suffix_carry;
            suffix_carry += read_lane(incl, 0);
            float dalpha = G * v.T - after / v.q;
            if (g_alpha && valid) dalpha += g_alpha[base + k];
            const float dsig = rc.w > 0.0f ? dalpha * v.de : 0.0f;
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

// forward: 8 blocks of 4 waves per CU keep all 256 CUs full (the kernels stay under 64 VGPRs); each wave
// then walks n_rays / 8192 rays
static inline unsigned ray_grid_fwd(int64_t n_rays) {
    int64_t blocks = (n_rays + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    return (unsigned)blocks;
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
    NERF_REQUIRE(n_rays >= 0 && S >= 2, NERF_E_SHAPE);
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(raw && z && rgb_map && depth_map && disp_map && acc_map && weights, NERF_E_NULL);
    NERF_REQUIRE(nerf_aligned(raw, 16), NERF_E_ALIGN);
    if (n_rays == 0) return NERF_OK;
    const dim3 grid(nerf::ray_grid_fwd(n_rays)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const float4* r4 = reinterpret_cast<const float4*>(raw);
#define NERF_FWD(K)                                                                                        \
    hipLaunchKernelGGL(nerf::K, grid, block, 0, st, n_rays, S, r4, z, white_bkgd, rgb_map, depth_map,      \
                       disp_map, acc_map, weights, alpha)
    const int nch = (S + 63) / 64;
    if (nch == 1) NERF_FWD(composite_fwd_kernel<1>);
    else if (nch == 2) NERF_FWD(composite_fwd_kernel<2>);
    else if (nch == 3) NERF_FWD(composite_fwd_kernel<3>);
    else if (nch == 4) NERF_FWD(composite_fwd_kernel<4>);
    else NERF_FWD(composite_fwd_long_kernel);
#undef NERF_FWD
    return nerf_launch_status();
}

extern "C" int nerf_composite_bwd(int64_t n_rays, int S, const float* raw, const float* z, int white_bkgd,
                                  const float* g_rgb, const float* g_depth, const float* g_disp,
                                  const float* g_acc, const float* g_weights, const float* g_alpha,
                                  float* d_raw, void* stream) {
    NERF_REQUIRE(n_rays >= 0 && S >= 2, NERF_E_SHAPE);
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
    else if (nch <= 16) NERF_BWD(16);
    else
        hipLaunchKernelGGL(nerf::composite_bwd_long_kernel, grid, block, 0, st, n_rays, S, r4, z, white_bkgd, g_rgb,
                           g_depth, g_disp, g_acc, g_weights, g_alpha, d4);
#undef NERF_BWD
    return nerf_launch_status();
}
