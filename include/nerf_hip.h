/*
 * nerf_hip.h -- C ABI of libnerf_hip.so, the MI355X (gfx950) implementation of
 * the NeRF render path.
 *
 * The reference (hadleyhzy34/nerf) has no FFI: its boundary is the Python call
 * surface of src/nerf_project/models/nerf.py and utils/ray_utils.py.  Each
 * entry point below names the reference lines whose arithmetic it replaces;
 * the Python package nerf_amd binds these symbols with ctypes and re-exposes
 * the reference's own signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless a comment says "host";
 *   - tensors are fp32, row-major, contiguous unless a stride is given;
 *   - the caller owns every buffer (inputs, outputs, workspaces); nothing is
 *     allocated, freed or retained by the library;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and
 *     the call returns without synchronising;
 *   - return value: 0 = ok, negative = NERF_E_* argument error (nothing was
 *     launched), positive = a hipError_t from the launch.
 *   - a call over nothing (zero rays, rows or samples) returns NERF_OK after
 *     its sizes have been checked and before any pointer is looked at: the
 *     tensors of an empty call have no address (the reference's torch code
 *     returns empty tensors in that case).
 *   - entry points are re-entrant and keep no mutable global state, with one
 *     exception that says so in its own comment: nerf_mlp_fwd_encoded_static
 *     (a per-device staging array inside the library, shared by all its calls).
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_HIP_VERSION 100 /* 0.1.0 */

enum {
    NERF_OK = 0,
    NERF_E_NULL = -1,      /* a required pointer is NULL            */
    NERF_E_SHAPE = -2,     /* a size is out of the supported range  */
    NERF_E_ALIGN = -3,     /* a pointer misses its required alignment */
    NERF_E_ARCH = -4,      /* unsupported network architecture      */
    NERF_E_WORKSPACE = -5, /* workspace too small                   */
    NERF_E_DTYPE = -6      /* unknown precision selector            */
};

/* precision selector of the MLP kernels (operand type; accumulation is fp32) */
enum { NERF_PREC_BF16 = 0, NERF_PREC_FP16 = 1 };

int nerf_version(void);
const char *nerf_error_string(int code);

/* ------------------------------------------------------------------------
 * get_rays -- src/nerf_project/utils/ray_utils.py:38-65.
 * Rows [row0, row0+nrows) of an H x W pinhole frame.  c2w: 12 floats (the
 * top 3x4 of the pose, row stride c2w_ld = 4).  rays_o, rays_d: [nrows, W, 3].
 */
int nerf_get_rays(int H, int W, float focal, const float *c2w, int c2w_ld, int row0, int nrows,
                  float *rays_o, float *rays_d, void *stream);

/* ------------------------------------------------------------------------
 * ray batch from an image bank -- datasets/blender.py:92-103 (x/255, alpha blend onto
 * black or white) and :146-161 (get_rays per image), evaluated per ray instead of per
 * image.  pix [n]: global pixel ids  image*H*W + y*W + x  (the caller guarantees
 * 0 <= pix < n_images*H*W).  poses: per image at least 12 floats (top 3x4 of the pose,
 * row-major), pose_stride floats apart (16 for [n,4,4]).  images: uint8
 * [n_images, H, W, channels], channels 3 or 4.  lut: 256 floats, lut[v] = v/255 as the
 * loader rounds it.  target may be NULL (rays only); then images and lut are not read.
 * Outputs rays_o, rays_d, target: [n, 3].
 */
int nerf_gather_rays(int64_t n, const int64_t *pix, int n_images, int H, int W, float focal,
                     const float *poses, int pose_stride, const uint8_t *images, int channels,
                     const float *lut, int white_bkgd, float *rays_o, float *rays_d, float *target,
                     void *stream);

/* ------------------------------------------------------------------------
 * stand-alone positional encoding -- models/nerf.py:68-76.
 * x [n, d] -> out [n, d*(1+2L)] = [x, sin(f0 x), cos(f0 x), ...]; freqs: L floats
 * (the module's freq_bands buffer).
 */
int nerf_posenc(int64_t n, int d, int L, const float *freqs, const float *x, float *out, void *stream);

/* ------------------------------------------------------------------------
 * sample placement -- models/nerf.py:373-402.
 * z_out [n_rays, S].  z_override (nullable) replaces the linspace in depth
 * (lindisp = 0) or disparity (lindisp = 1) between near and far; near_ray /
 * far_ray (nullable, [n_rays]) override the scalars.  t_rand (nullable,
 * [n_rays, S] uniforms in [0,1)) applies the stratified jitter of :393-402.
 */
int nerf_place_samples(int64_t n_rays, int S, float near_s, float far_s, const float *near_ray,
                       const float *far_ray, int lindisp, const float *z_override, const float *t_rand,
                       float *z_out, void *stream);

/* ------------------------------------------------------------------------
 * alpha compositing forward -- models/nerf.py:430-472.
 * raw [n_rays,S,4] (16-byte aligned), z [n_rays,S] ->
 * rgb_map [n_rays,3], depth/disp/acc [n_rays], weights [n_rays,S],
 * alpha [n_rays,S] (nullable).  S >= 2, no upper limit as in the reference (with one
 * sample the reference's own interval list is empty and its outputs degenerate);
 * rays of more than 256 samples take a kernel with a run-time chunk loop.
 */
int nerf_composite_fwd(int64_t n_rays, int S, const float *raw, const float *z, int white_bkgd,
                       float *rgb_map, float *depth_map, float *disp_map, float *acc_map, float *weights,
                       float *alpha, void *stream);

/* alpha compositing backward (the autograd of models/nerf.py:430-472).
 * Cotangents (each nullable = zero): g_rgb [n_rays,3], g_depth, g_disp, g_acc
 * [n_rays], g_weights, g_alpha [n_rays,S].  Output d_raw [n_rays,S,4].
 * The forward quantities are recomputed from raw and z.  S >= 2; rays of more than
 * 1024 samples take a two-pass kernel that keeps its per-chunk transmittance
 * products in d_raw before overwriting them (d_raw must not alias raw).
 */
int nerf_composite_bwd(int64_t n_rays, int S, const float *raw, const float *z, int white_bkgd,
                       const float *g_rgb, const float *g_depth, const float *g_disp, const float *g_acc,
                       const float *g_weights, const float *g_alpha, float *d_raw, void *stream);

/* ------------------------------------------------------------------------
 * sample_pdf -- models/nerf.py:237-313.
 * bins [n_rays, n_bins] (row stride bins_ld), weights [n_rays, n_bins-1]
 * (row stride w_ld); u (nullable) [n_rays, n_samples] uniforms, NULL =
 * deterministic linspace(0,1,n_samples).  samples [n_rays, n_samples];
 * inds (nullable) [n_rays, n_samples] = searchsorted(cdf, u, right) of :262.
 * 2 <= n_bins <= 1024.
 */
int nerf_sample_pdf(int64_t n_rays, int n_bins, const float *bins, int64_t bins_ld, const float *weights,
                    int64_t w_ld, int n_samples, const float *u, float *samples, int32_t *inds,
                    void *stream);

/* Coarse->fine glue -- scripts/train_nerf.py:309-320 (= render_nerf.py:288-294):
 * bins = midpoints of z_coarse, weights = w_coarse[1:-1] + 1e-5, sample_pdf,
 * then the ascending sort of cat(z_coarse, z_fine).
 * z_coarse, w_coarse [n_rays, Sc]; u nullable [n_rays, n_fine];
 * z_out [n_rays, Sc + n_fine]; z_fine_out nullable [n_rays, n_fine].
 * 3 <= Sc <= 1024, Sc + n_fine <= 1024.
 */
int nerf_sample_merge(int64_t n_rays, int Sc, const float *z_coarse, const float *w_coarse, int n_fine,
                      const float *u, float *z_out, float *z_fine_out, void *stream);

/* Row-wise ascending sort of cat(a[r], b[r]) -- the torch.sort(torch.cat(..))
 * of scripts/train_nerf.py:320.  b nullable (n_b = 0).  n_a + n_b <= 1024. */
int nerf_sort_rows(int64_t n_rows, int n_a, const float *a, int n_b, const float *b, float *out,
                   void *stream);

/* ------------------------------------------------------------------------
 * training loss -- scripts/train_nerf.py:299, :330-331.
 * a, b, target: n_elements floats each ([n_rays, 3] maps; b may be NULL for a coarse-only run).
 * losses[0..2] = { mean((a-t)^2) + mean((b-t)^2), mean((a-t)^2), mean((b-t)^2) }.
 * d_a, d_b (nullable): gradients of the two means, 2 (x - t) / n_elements.
 */
int nerf_mse_pair(int64_t n_elements, const float *a, const float *b, const float *target, float *losses,
                  float *d_a, float *d_b, void *stream);

/* ------------------------------------------------------------------------
 * Fused positional encoding + MLP -- models/nerf.py:68-76 (PositionalEncoder,
 * L=10 for points, L=4 for view directions, frequencies 2^l, no pi) and
 * models/nerf.py:173-209 (NeRF.forward, D=8, W=256, skips=[4], use_viewdirs).
 *
 * The network is described by a flat fp32 parameter buffer in state_dict
 * order (pts_linears.0.weight, pts_linears.0.bias, ..., rgb_layers.2.bias;
 * 595 844 floats).  nerf_mlp_pack converts it once per weight update into
 * the MFMA fragment streams the kernels read.
 */
#define NERF_MLP_NPARAMS 595844
/* The network without view directions (NeRF(use_viewdirs=False), models/nerf.py:151-153, :202-206, the
 * default of scripts/train_nerf.py:79): pts_linears.0 .. 7 then output_linear [4,256]; 494 084 floats.
 * Every MLP entry point takes it when NERF_ARCH_NOVIEW is or-ed into its `precision` argument (and
 * NERF_PACK_NOVIEW into nerf_mlp_pack's `streams`); buffer sizes are the same as for the full network,
 * d_params has NERF_MLP_NPARAMS_NOVIEW entries, and x_views may be NULL. */
#define NERF_MLP_NPARAMS_NOVIEW 494084
#define NERF_ARCH_NOVIEW 0x100
/* networks one multi-network call (nerf_mlp_pack_multi, nerf_mlp_bwd_weights_multi) can take */
#define NERF_WGRAD_MAX_NETS 2

/* bytes of the packed-weight buffer (forward + transposed streams + biases) */
int64_t nerf_mlp_packed_bytes(void);

/* params: flat fp32 [NERF_MLP_NPARAMS]; packed: nerf_mlp_packed_bytes() bytes, 256-byte aligned.
 * streams: which weight streams to (re)write -- NERF_PACK_TRAIN for nerf_mlp_fwd with `saved`,
 * nerf_mlp_fwd_encoded and the backward, NERF_PACK_INFER for nerf_mlp_fwd without `saved`, or both. */
enum { NERF_PACK_TRAIN = 1, NERF_PACK_INFER = 2, NERF_PACK_NOVIEW = 4 /* params describe the network without view directions */ };
int nerf_mlp_pack(const float *params, int precision, int streams, void *packed, void *stream);
/* The same for up to NERF_WGRAD_MAX_NETS networks in one launch (the coarse and the fine network after an
 * optimizer step): HOST arrays of n_nets parameter pointers, stream selectors and packed buffers. */
int nerf_mlp_pack_multi(int n_nets, const float *const *params, int precision, const int *streams,
                        void *const *packed, void *stream);

/* bytes of activations the training forward saves per sample (for nerf_mlp_bwd) */
int64_t nerf_mlp_saved_bytes(int64_t n_samples);
/* bytes of workspace nerf_mlp_bwd needs */
int64_t nerf_mlp_bwd_workspace_bytes(int64_t n_samples);

/* Forward.  Sample m = ray*S + k sits at rays_o[ray] + rays_d[ray] * z[ray,k]
 * with view direction rays_d[ray] / |rays_d[ray]| (models/nerf.py:409-421).
 * raw [n_rays*S, 4] = (rgb_raw, sigma_raw), 16-byte aligned.
 * saved (nullable): nerf_mlp_saved_bytes(n_rays*S) bytes; when given, the
 * hidden activations are kept for nerf_mlp_bwd.
 */
int nerf_mlp_fwd(int64_t n_rays, int S, const float *rays_o, const float *rays_d, const float *z,
                 const void *packed, int precision, float *raw, void *saved, void *stream);

/* Forward on pre-encoded inputs (the literal NeRF.forward signature, models/nerf.py:155):
 * x_pts [M,63], x_views [M,27] fp32.  The rows are first rewritten as 16-bit fragment units: into
 * `saved` when it is given (they are two of its tensors), otherwise into `workspace`
 * (nerf_mlp_encoded_workspace_bytes(M) bytes, 256-byte aligned); one of the two must be non-NULL. */
int64_t nerf_mlp_encoded_workspace_bytes(int64_t n_samples);
int nerf_mlp_fwd_encoded(int64_t M, const float *x_pts, const float *x_views, const void *packed,
                         int precision, float *raw, void *saved, void *workspace, void *stream);

/* ------------------------------------------------------------------------
 * fp32-operand forward of NeRF.forward -- src/nerf_project/models/nerf.py:173-209 for any
 * architecture its constructor (:100-153) accepts within the limits below: the validation mode
 * (v_mfma_f32_32x32x2_f32, exact fp32 products and sums).  The network is handed over as a list of
 * layers.  A layer multiplies one or two input segments (skip connection, nerf.py:184: [encoded
 * points, hidden]; colour head, :198: [feature, encoded view direction]) and writes either one of the
 * two hidden buffers or columns dst_col .. of raw [M,4] (nerf.py:209).  x_pts [M,cp], x_views [M,cv]
 * (NULL when cv = 0) are the encoded inputs of nerf.py:155.
 */
#define NERF_FP32_MAX_WIDTH 256  /* widest hidden layer */
#define NERF_FP32_MAX_PTS 64     /* encoded point features, padded */
#define NERF_FP32_MAX_VIEWS 32   /* encoded view-direction features, padded */
#define NERF_FP32_MAX_LAYERS 16
enum { NERF_FP32_BUF_PTS = 0, NERF_FP32_BUF_VIEWS = 1, NERF_FP32_BUF_A = 2, NERF_FP32_BUF_B = 3, NERF_FP32_DST_RAW = 4 };
typedef struct {
    const float *weights[2]; /* per segment: nerf_mlp_fp32_pack_segment output (device) */
    int k_padded[2];         /* per segment: input features rounded up to a multiple of 8 */
    int src[2];              /* per segment: NERF_FP32_BUF_* the inputs are read from */
    int n_segments;          /* 1 or 2 */
    const float *bias;       /* [out_features] (device) */
    int out_features;
    int relu;                /* 1: max(., 0) */
    int dst;                 /* NERF_FP32_BUF_A / _B, or NERF_FP32_DST_RAW */
    int dst_col;             /* first column of raw when dst is NERF_FP32_DST_RAW */
} nerf_fp32_layer_t;
/* floats of one packed segment of a [rows, cols] weight block */
int64_t nerf_mlp_fp32_segment_floats(int rows, int cols);
/* w: row-major [rows, ld]; columns col0 .. col0+cols-1 are rewritten in MFMA fragment order, zero padded */
int nerf_mlp_fp32_pack_segment(const float *w, int rows, int ld, int col0, int cols, float *dst, void *stream);
/* layers: HOST array of n_layers descriptors (copied into the launch) */
int nerf_mlp_fp32_fwd(int64_t M, const float *x_pts, int cp, const float *x_views, int cv,
                      const nerf_fp32_layer_t *layers, int n_layers, float *raw, void *stream);

/* ------------------------------------------------------------------------
 * fp32 backward -- the autograd of src/nerf_project/models/nerf.py:173-209 with respect to the parameters,
 * as scripts/train_nerf.py:334-335 runs it in fp32, on v_mfma_f32_32x32x2_f32 for the same layer lists as
 * nerf_mlp_fp32_fwd.  nerf_mlp_fp32_fwd_train is that forward keeping, in `saved`
 * (nerf_mlp_fp32_saved_floats(...) floats, 16-byte aligned), the encoded inputs and every hidden layer's output.
 * nerf_mlp_fp32_bwd then takes d_raw [M,4] and writes every layer's weight and bias gradient:
 *   1. the backward-data chain dH_in = W^T dZ, dZ = dH * [y > 0], layer by layer in LDS like the forward (it is
 *      the forward kernel run on the transposed weights), keeping every layer's dZ in the workspace;
 *   2. per layer segment dW = sum over samples dZ x input^T, the sample axis split over workgroups into fp32
 *      partial blocks that a second kernel adds up in a fixed order (bitwise reproducible); db = row sums of dZ.
 * Gradients with respect to the encoded inputs are not produced.
 * grads: HOST array, one entry per layer of `layers`:
 *   wt[s]      per segment whose source is a hidden buffer: nerf_mlp_fp32_pack_segment output of the TRANSPOSED
 *              weight block, i.e. of the [cols[s], out_features] matrix W[:, col0[s] : col0[s] + cols[s]]^T with
 *              col0 = 0, cols = out_features; for a layer that writes raw (dst = NERF_FP32_DST_RAW) of the
 *              [cols[s], 4] matrix holding that block in columns dst_col .. and zeros elsewhere.  NULL for a
 *              segment read from the encoded inputs (no gradient flows there).
 *   d_weight   [out_features, ld] row-major (the layer's weight gradient, written, not accumulated);
 *   col0/cols  per segment: the columns of d_weight it owns (cols rounded up to 8 = the layer's k_padded);
 *   d_bias     [out_features].
 * workspace: nerf_mlp_fp32_bwd_workspace_floats(...) floats, 16-byte aligned.
 */
typedef struct {
    const float *wt[2];
    float *d_weight;
    int ld;
    int col0[2];
    int cols[2];
    float *d_bias;
} nerf_fp32_grad_t;
int64_t nerf_mlp_fp32_saved_floats(int64_t M, int cp, int cv, const nerf_fp32_layer_t *layers, int n_layers);
int64_t nerf_mlp_fp32_bwd_workspace_floats(int64_t M, int cp, int cv, const nerf_fp32_layer_t *layers, int n_layers);
int nerf_mlp_fp32_fwd_train(int64_t M, const float *x_pts, int cp, const float *x_views, int cv,
                            const nerf_fp32_layer_t *layers, int n_layers, float *raw, float *saved, void *stream);
int nerf_mlp_fp32_bwd(int64_t M, int cp, int cv, const nerf_fp32_layer_t *layers, const nerf_fp32_grad_t *grads,
                      int n_layers, const float *d_raw, const float *saved, float *workspace,
                      int64_t workspace_floats, void *stream);

/* ------------------------------------------------------------------------
 * The same layer programs with 16-bit operands (NERF_PREC_BF16 / NERF_PREC_FP16, fp32 accumulation, fp32 bias and
 * raw) on v_mfma_f32_32x32x16: NeRF.forward and its parameter gradients (src/nerf_project/models/nerf.py:173-209,
 * scripts/train_nerf.py:334-335) for the architectures the constructor (:100-153) accepts that the unrolled 16-bit
 * kernels above (nerf_mlp_fwd / nerf_mlp_bwd_*) are not compiled for, e.g. the reference's own test network D=4,
 * W=64, skips=[2] (tests/test_nerf_model.py:105-109).  Same arithmetic model as those kernels: weights and every
 * layer's input rounded to the operand type.  The entry points mirror the nerf_mlp_fp32_* ones and take the same
 * nerf_fp32_layer_t / nerf_fp32_grad_t lists, with these differences: a segment's input features are padded to a
 * multiple of 16 (k_padded, and the cols of a gradient segment rounded up to 16); `weights` and `wt` point to
 * nerf_mlp_p16_pack_segment outputs (16-bit, cast to the struct's pointer type); saved activations and the gradient
 * rows are kept in the operand type (sizes in bytes; saved 16-byte, workspace 256-byte aligned); grad_scale
 * (nullable, device): d_raw is multiplied by *grad_scale on entry and the gradients divided by it on exit, which
 * keeps fp16 gradients inside the format's range (a power of two keeps the result unchanged otherwise).
 */
int64_t nerf_mlp_p16_segment_bytes(int rows, int cols);
int nerf_mlp_p16_pack_segment(const float *w, int rows, int ld, int col0, int cols, int precision, void *dst, void *stream);
int nerf_mlp_p16_fwd(int64_t M, const float *x_pts, int cp, const float *x_views, int cv,
                     const nerf_fp32_layer_t *layers, int n_layers, int precision, float *raw, void *stream);
int64_t nerf_mlp_p16_saved_bytes(int64_t M, int cp, int cv, const nerf_fp32_layer_t *layers, int n_layers);
int64_t nerf_mlp_p16_bwd_workspace_bytes(int64_t M, int cp, int cv, const nerf_fp32_layer_t *layers, int n_layers);
int nerf_mlp_p16_fwd_train(int64_t M, const float *x_pts, int cp, const float *x_views, int cv,
                           const nerf_fp32_layer_t *layers, int n_layers, int precision, float *raw, void *saved,
                           void *stream);
int nerf_mlp_p16_bwd(int64_t M, int cp, int cv, const nerf_fp32_layer_t *layers, const nerf_fp32_grad_t *grads,
                     int n_layers, int precision, const float *d_raw, const void *saved, const float *grad_scale,
                     void *workspace, int64_t workspace_bytes, void *stream);

/* The same forward as nerf_mlp_fwd_encoded(saved = NULL) with the workspace inside the library.  It has no
 * counterpart in the reference: it exists for callers that cannot allocate device memory -- code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The call allocates nothing
 * (no hipMalloc / hipFree, no allocating library routine): its staging buffer is a static device array of
 * 12 MiB that every GPU the library runs on holds until the process ends, so M is limited to
 * NERF_MLP_FWD_ENCODED_STATIC_MAX samples (NERF_E_SHAPE beyond that).  Results are bit-identical to
 * nerf_mlp_fwd_encoded.  All calls on one GPU share the one array: calls that overlap in time on the same
 * GPU (two streams, two threads) race; calls on one stream are ordered by the stream.  The first call on a
 * device looks up that device's copy of the array (hipGetSymbolAddress, kept under a mutex: the library's only
 * mutable state): make it outside stream capture; later calls on that device can be captured into a HIP graph. */
#define NERF_MLP_FWD_ENCODED_STATIC_MAX 65536
int64_t nerf_mlp_fwd_encoded_static_max_samples(void);
int nerf_mlp_fwd_encoded_static(int64_t M, const float *x_pts, const float *x_views, const void *packed,
                                int precision, float *raw, void *stream);

/* Backward: d_raw [M,4] -> d_params flat fp32 [NERF_MLP_NPARAMS] (overwritten,
 * not accumulated).  saved is what nerf_mlp_fwd wrote for the same samples.
 * grad_scale (nullable): one device float s; the incoming gradient is multiplied
 * by s before it is rounded to 16 bits and the result divided by s, which keeps
 * small gradients inside the fp16 range (a power of two leaves bf16/fp16 mantissas
 * untouched). */
int nerf_mlp_bwd(int64_t M, const float *d_raw, const void *packed, int precision, const void *saved,
                 const float *grad_scale, void *workspace, int64_t workspace_bytes, float *d_params,
                 void *stream);

/* Weight gradients of up to NERF_WGRAD_MAX_NETS networks in one launch sequence -- the second half of
 * loss.backward() (scripts/train_nerf.py:334-335) for the coarse and the fine network together: one
 * contraction kernel over all their GEMMs, one reduction, one launch for the factored gradients.  Each
 * GEMM is then split over half as many workgroups as in nerf_mlp_bwd_weights, which halves the partial
 * sums written and read.  nets: HOST array; per network the arguments of nerf_mlp_bwd_weights (M >= 1; all
 * networks share the operand type, the architecture flag may differ).  partial:
 * nerf_mlp_bwd_weights_multi_bytes(n_nets, nets) bytes, 256-byte aligned.  Results are bit-identical from
 * call to call (fixed summation order), not to those of separate nerf_mlp_bwd_weights calls (the sample
 * axis is cut differently). */
typedef struct {
    int64_t M;               /* samples */
    const void *packed;      /* nerf_mlp_pack output of the network */
    const void *saved;       /* activations kept by nerf_mlp_fwd */
    const void *dz;          /* written by nerf_mlp_bwd_data */
    int precision;           /* NERF_PREC_* | NERF_ARCH_* */
    const float *grad_scale; /* nullable, as in nerf_mlp_bwd_weights */
    float *d_params;         /* [NERF_MLP_NPARAMS] or [NERF_MLP_NPARAMS_NOVIEW] */
} nerf_wgrad_net_t;
int64_t nerf_mlp_bwd_weights_multi_bytes(int n_nets, const nerf_wgrad_net_t *nets);
int nerf_mlp_bwd_weights_multi(int n_nets, const nerf_wgrad_net_t *nets, void *partial, int64_t partial_bytes,
                               void *stream);

/* The two halves of nerf_mlp_bwd as separate calls (same kernels), for callers that want to
 * time or overlap them: the backward-data chain writes every layer's dZ (16 bit,
 * nerf_mlp_dz_bytes(M) bytes); the weight-gradient pass contracts dZ with the saved activations
 * over the samples (workspace: nerf_mlp_partial_bytes(M) bytes of fp32 partial sums). */
int64_t nerf_mlp_dz_bytes(int64_t n_samples);
int64_t nerf_mlp_partial_bytes(int64_t n_samples);
int nerf_mlp_bwd_data(int64_t M, const float *d_raw, const void *packed, int precision, const void *saved,
                      const float *grad_scale, void *dz, void *stream);
int nerf_mlp_bwd_weights(int64_t M, const void *packed, const void *saved, const void *dz, int precision,
                         const float *grad_scale, void *partial, int64_t partial_bytes, float *d_params,
                         void *stream);

/* Adam on a flat fp32 buffer -- the optimizer.step() of scripts/train_nerf.py:336
 * (torch.optim.Adam, no weight decay, no amsgrad).  step is 1-based. */
int nerf_adam_step(int64_t n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                   float lr, float beta1, float beta2, float eps, int64_t step, float grad_scale,
                   void *stream);

/* Same update with the hyper-parameters in device memory, so that the call can be captured in a
 * HIP graph and replayed while lr changes: state = device float[8] = {lr, beta1, beta2, eps,
 * grad_scale, step, ticket, scratch}.  The call applies the update of step state[5] + 1 and leaves
 * state[5] advanced by one (it starts from 0); state[6] is a workgroup counter: it MUST be 0 before the
 * first call (zero-initialise the array) and is 0 again after every call.  With anything else there the
 * step counter stops advancing and the bias corrections are wrong from then on. */
int nerf_adam_step_dev(int64_t n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                       float *state, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */
