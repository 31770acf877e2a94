// Ray generation and sample placement: bandwidth-bound, write-mostly kernels.
//   nerf_get_rays      <- src/nerf_project/utils/ray_utils.py:38-65
//   nerf_place_samples <- src/nerf_project/models/nerf.py:373-402
//   nerf_gather_rays   <- src/nerf_project/datasets/blender.py:92-103, :146-161 (per ray instead of per image)
// Compiled with -ffp-contract=off: the reference evaluates these formulas as
// separate fp32 multiplies and adds.
#include "common.hpp"

namespace nerf {

// Each thread produces 4 consecutive pixels = 12 floats per output, so a wave owns 3 KiB of contiguous
// memory per tensor; the directions pass through a per-wave LDS tile so that the stores are whole cache
// lines (5.1 TB/s on a 980 MB launch against 3.8 TB/s with each lane storing its own 48 bytes).
__global__ __launch_bounds__(256) void get_rays_kernel(int H, int W, float focal, const float* __restrict__ c2w,
                                                       int ld, int row0, int64_t npix, float* __restrict__ ro,
                                                       float* __restrict__ rd) {
    const float r00 = c2w[0], r01 = c2w[1], r02 = c2w[2], ox = c2w[3];
    const float r10 = c2w[ld + 0], r11 = c2w[ld + 1], r12 = c2w[ld + 2], oy = c2w[ld + 3];
    const float r20 = c2w[2 * ld + 0], r21 = c2w[2 * ld + 1], r22 = c2w[2 * ld + 2], oz = c2w[2 * ld + 3];
    const float hw = (float)(W * 0.5), hh = (float)(H * 0.5);
    const int lane = threadIdx.x & 63;
    __shared__ __attribute__((aligned(16))) float tile[4][64 * 12];
    const int64_t ngroups = (npix + 3) >> 2;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
         g += (int64_t)gridDim.x * blockDim.x) {
        float d[12], o[12];
        const int64_t p0 = g << 2;
        // one division per thread (32-bit whenever the pixel count allows), then step along the row
        const int64_t line = npix <= 0x7fffffff ? (int64_t)((uint32_t)p0 / (uint32_t)W) : p0 / W;
        int y = (int)line + row0, x = (int)(p0 - line * W);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dx = ((float)x - hw) / focal;
            const float dy = ((float)y - hh) / focal;
            const float dz = -1.0f;
            float wx = (dx * r00 + dy * r01) + dz * r02;
            float wy = (dx * r10 + dy * r11) + dz * r12;
            float wz = (dx * r20 + dy * r21) + dz * r22;
            const float n = sqrtf((wx * wx + wy * wy) + wz * wz);
            d[3 * i + 0] = wx / n; d[3 * i + 1] = wy / n; d[3 * i + 2] = wz / n;
            o[3 * i + 0] = ox; o[3 * i + 1] = oy; o[3 * i + 2] = oz;
            if (++x == W) { x = 0; ++y; }
        }
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int64_t g0 = g - lane;                    // the wave's first group: 256 consecutive pixels from 4 * g0
        if (4 * (g0 + 64) <= npix) {
            // whole wave inside the image: exchange through the wave's LDS tile so that every store instruction
            // writes 1 KiB of consecutive addresses (full cache lines) instead of 16 bytes every 48
            float* t = tile[threadIdx.x >> 6];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                *reinterpret_cast<f32x4*>(t + 12 * lane + 4 * i) = f32x4{d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]};
            wave_lds_sync();
            f32x4* dd = reinterpret_cast<f32x4*>(rd + 12 * g0);
            f32x4* oo = reinterpret_cast<f32x4*>(ro + 12 * g0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = 64 * i + lane;
                const int ph = q % 3;                   // float 4q of the (x, y, z, x, y, z, ...) sequence is component q % 3
                const float e0 = ph == 0 ? ox : (ph == 1 ? oy : oz);
                const float e1 = ph == 0 ? oy : (ph == 1 ? oz : ox);
                const float e2 = ph == 0 ? oz : (ph == 1 ? ox : oy);
                dd[q] = *reinterpret_cast<const f32x4*>(t + 4 * q);
                oo[q] = f32x4{e0, e1, e2, e0};
            }
            wave_lds_sync();
        } else if (p0 + 4 <= npix) {
            f32x4* dd = reinterpret_cast<f32x4*>(rd + 3 * p0);
            f32x4* oo = reinterpret_cast<f32x4*>(ro + 3 * p0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                dd[i] = f32x4{d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]};
                oo[i] = f32x4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
            }
        } else {
            const int rem = (int)(npix - p0) * 3;
            for (int i = 0; i < rem; ++i) { rd[3 * p0 + i] = d[i]; ro[3 * p0 + i] = o[i]; }
        }
    }
}

// One training ray per thread: pixel `pix` of the image bank -> its ray (same arithmetic as
// get_rays_kernel for that pixel and that image's pose) and its target colour (8-bit channels through
// the 256-entry table `lut` = x/255 as the loader computes it, then the loader's alpha blend).
__global__ __launch_bounds__(256) void gather_rays_kernel(int64_t n, const int64_t* __restrict__ pix, int H, int W,
                                                          float focal, const float* __restrict__ poses,
                                                          int pose_stride, const uint8_t* __restrict__ images, int C,
                                                          const float* __restrict__ lut, int white,
                                                          float* __restrict__ ro, float* __restrict__ rd,
                                                          float* __restrict__ target) {
    const float hw = (float)(W * 0.5), hh = (float)(H * 0.5);
    const int64_t hwp = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = pix[i];
        const int64_t img = p / hwp;
        const int64_t q = p - img * hwp;
        const int y = (int)(q / W), x = (int)(q % W);
        const float* c2w = poses + img * pose_stride;
        const float dx = ((float)x - hw) / focal;
        const float dy = ((float)y - hh) / focal;
        const float dz = -1.0f;
        const float wx = (dx * c2w[0] + dy * c2w[1]) + dz * c2w[2];
        const float wy = (dx * c2w[4] + dy * c2w[5]) + dz * c2w[6];
        const float wz = (dx * c2w[8] + dy * c2w[9]) + dz * c2w[10];
        const float nrm = sqrtf((wx * wx + wy * wy) + wz * wz);
        rd[3 * i + 0] = wx / nrm; rd[3 * i + 1] = wy / nrm; rd[3 * i + 2] = wz / nrm;
        ro[3 * i + 0] = c2w[3]; ro[3 * i + 1] = c2w[7]; ro[3 * i + 2] = c2w[11];
        if (target) {
            const uint8_t* px = images + p * C;
            float r = lut[px[0]], g = lut[px[1]], b = lut[px[2]];
            if (C == 4) {
                const float a = lut[px[3]];
                r = r * a; g = g * a; b = b * a;                       // blender.py:103
                if (white) { const float w1 = 1.0f - a; r = r + w1; g = g + w1; b = b + w1; }   // blender.py:100
            }
            target[3 * i + 0] = r; target[3 * i + 1] = g; target[3 * i + 2] = b;
        }
    }
}

__device__ __forceinline__ float base_z(int k, int S, float nr, float fr, int lindisp) {
    const float t = linspace01(k, S);
    if (lindisp) return 1.0f / ((1.0f / nr) * (1.0f - t) + (1.0f / fr) * t);
    return nr * (1.0f - t) + fr * t;
}

__global__ __launch_bounds__(256) void place_samples_kernel(int64_t total, int S, float near_s, float far_s,
                                                            const float* __restrict__ near_ray,
                                                            const float* __restrict__ far_ray, int lindisp,
                                                            const float* __restrict__ zov,
                                                            const float* __restrict__ t_rand,
                                                            float* __restrict__ zout) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ray = i / S;
        const int k = (int)(i - ray * S);
        float zm, z0, zp;  // z[k-1], z[k], z[k+1]
        if (zov) {
            z0 = zov[i];
            zm = k > 0 ? zov[i - 1] : z0;
            zp = k + 1 < S ? zov[i + 1] : z0;
        } else {
            const float nr = near_ray ? near_ray[ray] : near_s;
            const float fr = far_ray ? far_ray[ray] : far_s;
            z0 = base_z(k, S, nr, fr, lindisp);
            zm = k > 0 ? base_z(k - 1, S, nr, fr, lindisp) : z0;
            zp = k + 1 < S ? base_z(k + 1, S, nr, fr, lindisp) : z0;
        }
        float z = z0;
        if (t_rand) {
            const float lower = k > 0 ? 0.5f * (z0 + zm) : z0;
            const float upper = k + 1 < S ? 0.5f * (zp + z0) : z0;
            z = lower + (upper - lower) * t_rand[i];
        }
        zout[i] = z;
    }
}

}  // namespace nerf

extern "C" int nerf_get_rays(int H, int W, float focal, const float* c2w, int c2w_ld, int row0, int nrows,
                             float* rays_o, float* rays_d, void* stream) {
    NERF_REQUIRE(H > 0 && W > 0 && c2w_ld >= 4 && row0 >= 0 && nrows >= 0 && row0 + nrows <= H, NERF_E_SHAPE);
    NERF_REQUIRE(focal != 0.0f, NERF_E_SHAPE);
    const int64_t npix = (int64_t)nrows * W;
    if (npix == 0) return NERF_OK;          // an empty call: its output tensors have no address
    NERF_REQUIRE(c2w && rays_o && rays_d, NERF_E_NULL);
    NERF_REQUIRE(nerf_aligned(rays_o, 16) && nerf_aligned(rays_d, 16), NERF_E_ALIGN);
    const int64_t groups = (npix + 3) / 4;
    int64_t blocks = (groups + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nerf::get_rays_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, H, W,
                       focal, c2w, c2w_ld, row0, npix, rays_o, rays_d);
    return nerf_launch_status();
}

extern "C" int nerf_gather_rays(int64_t n, const int64_t* pix, int n_images, int H, int W, float focal,
                                const float* poses, int pose_stride, const uint8_t* images, int channels,
                                const float* lut, int white_bkgd, float* rays_o, float* rays_d, float* target,
                                void* stream) {
    NERF_REQUIRE(n >= 0 && n_images >= 1 && H > 0 && W > 0 && pose_stride >= 12, NERF_E_SHAPE);
    NERF_REQUIRE(channels == 3 || channels == 4 || target == nullptr, NERF_E_SHAPE);
    NERF_REQUIRE(focal != 0.0f, NERF_E_SHAPE);
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(pix && poses && rays_o && rays_d, NERF_E_NULL);
    NERF_REQUIRE((target == nullptr) || (images && lut), NERF_E_NULL);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nerf::gather_rays_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, pix, H,
                       W, focal, poses, pose_stride, images, channels, lut, white_bkgd, rays_o, rays_d, target);
    return nerf_launch_status();
}

extern "C" int nerf_place_samples(int64_t n_rays, int S, float near_s, float far_s, const float* near_ray,
                                  const float* far_ray, int lindisp, const float* z_override,
                                  const float* t_rand, float* z_out, void* stream) {
    NERF_REQUIRE(n_rays >= 0 && S >= 1 && S <= 65536, NERF_E_SHAPE);
    const int64_t total = n_rays * S;
    if (total == 0) return NERF_OK;
    NERF_REQUIRE(z_out, NERF_E_NULL);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(nerf::place_samples_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       total, S, near_s, far_s, near_ray, far_ray, lindisp, z_override, t_rand, z_out);
    return nerf_launch_status();
}

// ---- stand-alone positional encoding (models/nerf.py:68-76) ------------------------------
namespace nerf {
__global__ __launch_bounds__(256) void posenc_kernel(int64_t total, int d, int L, const float* __restrict__ freqs,
                                                     const float* __restrict__ x, float* __restrict__ out) {
    const int od = d * (1 + 2 * L);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / od;
        const int c = (int)(i - row * od);
        float v;
        if (c < d) {
            v = x[row * d + c];
        } else {
            const int b = (c - d) / d;          // 2*l + (0: sin, 1: cos)
            const int j = (c - d) - b * d;
            const float a = x[row * d + j] * freqs[b >> 1];
            v = (b & 1) ? cosf(a) : sinf(a);
        }
        out[i] = v;
    }
}
}  // namespace nerf

extern "C" int nerf_posenc(int64_t n, int d, int L, const float* freqs, const float* x, float* out, void* stream) {
    NERF_REQUIRE(n >= 0 && d >= 1 && L >= 0 && L <= 64, NERF_E_SHAPE);
    const int64_t total = n * d * (1 + 2 * L);
    if (total == 0) return NERF_OK;
    NERF_REQUIRE(x && out && (freqs || L == 0), NERF_E_NULL);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(nerf::posenc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, total, d, L,
                       freqs, x, out);
    return nerf_launch_status();
}
