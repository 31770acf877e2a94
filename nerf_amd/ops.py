"""Python-side launchers for the HIP kernels: tensor validation, output allocation,
and the torch.autograd.Function wrappers.  Every function here requires GPU
(ROCm "cuda") fp32 tensors and enqueues on torch's current stream; nothing
falls back to PyTorch arithmetic.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _C
from ._C import check, lib


# ---------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------

def _gpu_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got device {t.device}); nerf_amd has no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(dev: torch.device):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# Optional per-kernel timing with events on the launch stream (used by bench.py for the roofline
# figure).  KERNEL_TIMES maps a kernel name to a list of (start_event, end_event, units).
KERNEL_TIMES = None
# A callable(stream) -> int, or None.  When set, only calls made while the stream is being captured into a HIP
# graph are recorded (those whose name is in its ``names`` attribute, when it has one): the callable enqueues a
# time-stamp kernel on the capture stream and returns the index of the slot it writes, so that every replay of the
# graph stamps the slots and the difference of a pair is the duration inside the replayed step (bench.py owns the
# kernel and the slots; ordinary HIP events cannot be recorded into a captured graph on ROCm).  KERNEL_TIMES then
# holds (slot before, slot after, units).
KERNEL_STAMP = None


class _timed:
    def __init__(self, name, dev, units):
        self.name, self.dev, self.units = name, dev, units
        self.on = False

    def __enter__(self):
        if KERNEL_TIMES is None:
            return
        if KERNEL_STAMP is not None:
            names = getattr(KERNEL_STAMP, "names", None)
            if torch.cuda.is_current_stream_capturing() and (names is None or self.name in names):
                self.on = True
                self.e0 = KERNEL_STAMP(torch.cuda.current_stream(self.dev))
            return
        self.on = True
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record(torch.cuda.current_stream(self.dev))

    def __exit__(self, *exc):
        if self.on:
            if KERNEL_STAMP is not None:
                self.e1 = KERNEL_STAMP(torch.cuda.current_stream(self.dev))
            else:
                self.e1.record(torch.cuda.current_stream(self.dev))
            KERNEL_TIMES.setdefault(self.name, []).append((self.e0, self.e1, self.units))
        return False


# ---------------------------------------------------------------------------
# rays / sample placement / positional encoding
# ---------------------------------------------------------------------------

def get_rays(H: int, W: int, focal: float, c2w: torch.Tensor, row0: int = 0,
             nrows: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rows [row0, row0+nrows) of the pinhole frame; (rays_o, rays_d) of shape [nrows, W, 3]."""
    c2w = _gpu_f32(c2w, "c2w")
    if c2w.dim() != 2 or c2w.shape[1] != 4 or c2w.shape[0] not in (3, 4):
        raise ValueError(f"c2w must be [4,4] or [3,4], got {tuple(c2w.shape)}")
    H, W = int(H), int(W)
    nrows = H - row0 if nrows is None else int(nrows)
    ro = torch.empty((nrows, W, 3), dtype=torch.float32, device=c2w.device)
    rd = torch.empty((nrows, W, 3), dtype=torch.float32, device=c2w.device)
    with torch.cuda.device(c2w.device), _timed("get_rays", c2w.device, nrows * W):
        check(lib.nerf_get_rays(H, W, float(focal), _p(c2w), 4, int(row0), nrows, _p(ro), _p(rd),
                                _stream(c2w.device)), "nerf_get_rays")
    return ro, rd


def gather_rays(pix: torch.Tensor, H: int, W: int, focal: float, poses: torch.Tensor,
                images: Optional[torch.Tensor] = None, lut: Optional[torch.Tensor] = None,
                white_bkgd: bool = False, out: Optional[Sequence[torch.Tensor]] = None):
    """Rays (and blended target colours when ``images`` is given) of the global pixel ids ``pix``
    (image*H*W + y*W + x) of an image bank: (rays_o, rays_d, target or None), each [n, 3].
    ``out`` = (rays_o, rays_d, target) writes into existing contiguous fp32 [n, 3] tensors instead."""
    poses = _gpu_f32(poses, "poses")
    if poses.dim() != 3 or poses.shape[1] not in (3, 4) or poses.shape[2] != 4:
        raise ValueError(f"poses must be [n_images, 4, 4] or [n_images, 3, 4], got {tuple(poses.shape)}")
    dev = poses.device
    if pix.device != dev or pix.dtype != torch.int64 or pix.dim() != 1:
        raise RuntimeError("pix must be a 1-D int64 tensor on the poses' device")
    pix = pix.contiguous()
    n, n_img = pix.numel(), poses.shape[0]
    if out is not None:
        for t in out:
            if t is not None and (t.device != dev or t.dtype != torch.float32 or tuple(t.shape) != (n, 3)
                                  or not t.is_contiguous()):
                raise RuntimeError(f"out tensors must be contiguous fp32 [{n}, 3] on the poses' device")
        ro, rd = out[0], out[1]
    else:
        ro = torch.empty((n, 3), dtype=torch.float32, device=dev)
        rd = torch.empty((n, 3), dtype=torch.float32, device=dev)
    tgt, C = None, 0
    if images is not None:
        if (images.device != dev or images.dtype != torch.uint8 or images.dim() != 4 or not images.is_contiguous()
                or tuple(images.shape[:3]) != (n_img, int(H), int(W)) or images.shape[3] not in (3, 4)):
            raise RuntimeError("images must be a contiguous uint8 [n_images, H, W, 3|4] tensor on the poses' device")
        lut = _gpu_f32(lut, "lut")
        if lut.numel() != 256:
            raise ValueError("lut must hold 256 floats")
        C = images.shape[3]
        tgt = out[2] if out is not None else torch.empty((n, 3), dtype=torch.float32, device=dev)
        if tgt is None:
            raise RuntimeError("out[2] is needed for the target colours")
    with torch.cuda.device(dev):
        check(lib.nerf_gather_rays(n, _p(pix), n_img, int(H), int(W), float(focal), _p(poses),
                                   poses.shape[1] * 4, _p(images), C, _p(lut), int(bool(white_bkgd)), _p(ro), _p(rd),
                                   _p(tgt), _stream(dev)), "nerf_gather_rays")
    return ro, rd, tgt


def posenc(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    x = _gpu_f32(x, "x")
    d = x.shape[-1]
    L = int(freqs.numel())
    out = torch.empty(x.shape[:-1] + (d * (1 + 2 * L),), dtype=torch.float32, device=x.device)
    n = x.numel() // d if d else 0
    fr = _gpu_f32(freqs, "freq_bands") if L else None
    with torch.cuda.device(x.device):
        check(lib.nerf_posenc(n, d, L, _p(fr), _p(x), _p(out), _stream(x.device)), "nerf_posenc")
    return out


def place_samples(n_rays: int, near, far, n_samples: int, lindisp: bool, device: torch.device,
                  z_vals_override: Optional[torch.Tensor] = None,
                  t_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """z_vals [n_rays, S]: linspace in depth/disparity (or the override) plus optional jitter."""
    near_t = far_t = None
    near_s = far_s = 0.0
    if z_vals_override is not None:
        zov = _gpu_f32(z_vals_override, "z_vals_override")
        if zov.dim() != 2 or zov.shape[0] != n_rays:
            raise ValueError(f"z_vals_override must be [N_rays, S], got {tuple(zov.shape)}")
        S = zov.shape[1]
        device = zov.device
    else:
        zov = None
        S = int(n_samples)

        def scal(v, nm):
            if torch.is_tensor(v):
                if v.numel() == 1:
                    return float(v), None
                vt = _gpu_f32(v.to(device), nm).reshape(-1)
                if vt.numel() != n_rays:
                    raise ValueError(f"{nm} must be a scalar or have one value per ray")
                return 0.0, vt
            return float(v), None
        near_s, near_t = scal(near, "near")
        far_s, far_t = scal(far, "far")
    if t_rand is not None:
        t_rand = _gpu_f32(t_rand, "t_rand")
        if tuple(t_rand.shape) != (n_rays, S):
            raise ValueError(f"t_rand must be [{n_rays}, {S}], got {tuple(t_rand.shape)}")
    z = torch.empty((n_rays, S), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        check(lib.nerf_place_samples(n_rays, S, near_s, far_s, _p(near_t), _p(far_t), int(bool(lindisp)),
                                     _p(zov), _p(t_rand), _p(z), _stream(device)), "nerf_place_samples")
    return z


# ---------------------------------------------------------------------------
# compositing
# ---------------------------------------------------------------------------

class _Composite(torch.autograd.Function):
    """raw [N,S,4], z [N,S] -> rgb_map, depth_map, disp_map, acc_map, weights, alpha."""

    @staticmethod
    def forward(ctx, raw, z, white_bkgd: bool):
        raw = _gpu_f32(raw, "raw")
        z = _gpu_f32(z, "z_vals")
        N, S = z.shape
        if raw.shape != (N, S, 4):
            raise ValueError(f"raw must be [{N},{S},4], got {tuple(raw.shape)}")
        dev = raw.device
        rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
        depth = torch.empty((N,), dtype=torch.float32, device=dev)
        disp = torch.empty((N,), dtype=torch.float32, device=dev)
        acc = torch.empty((N,), dtype=torch.float32, device=dev)
        weights = torch.empty((N, S), dtype=torch.float32, device=dev)
        alpha = torch.empty((N, S), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev), _timed(f"composite_fwd_S{S}", dev, N):
            check(lib.nerf_composite_fwd(N, S, _p(raw), _p(z), int(bool(white_bkgd)), _p(rgb), _p(depth),
                                         _p(disp), _p(acc), _p(weights), _p(alpha), _stream(dev)),
                  "nerf_composite_fwd")
        ctx.save_for_backward(raw, z)
        ctx.white = bool(white_bkgd)
        ctx.set_materialize_grads(False)      # unused outputs arrive as None, not as tensors of zeros
        return rgb, depth, disp, acc, weights, alpha

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_disp, g_acc, g_w, g_alpha):
        raw, z = ctx.saved_tensors
        N, S = z.shape
        dev = raw.device

        def prep(g, nm):
            return None if g is None else _gpu_f32(g, nm)
        g_rgb, g_depth, g_disp, g_acc = prep(g_rgb, "g_rgb"), prep(g_depth, "g_depth"), prep(g_disp, "g_disp"), prep(g_acc, "g_acc")
        g_w, g_alpha = prep(g_w, "g_weights"), prep(g_alpha, "g_alpha")
        d_raw = torch.empty_like(raw)
        with torch.cuda.device(dev), _timed(f"composite_bwd_S{S}", dev, N):
            check(lib.nerf_composite_bwd(N, S, _p(raw), _p(z), int(ctx.white), _p(g_rgb), _p(g_depth), _p(g_disp),
                                         _p(g_acc), _p(g_w), _p(g_alpha), _p(d_raw), _stream(dev)),
                  "nerf_composite_bwd")
        return d_raw, None, None


class _MsePair(torch.autograd.Function):
    """(total, loss_a, loss_b) of train_nerf.py:299, :330-331; only ``total`` is differentiable."""

    @staticmethod
    def forward(ctx, a, b, target, out):
        a, b, target = _gpu_f32(a, "rgb_coarse"), _gpu_f32(b, "rgb_fine"), _gpu_f32(target, "target_rgb")
        if a.shape != target.shape or b.shape != target.shape:
            raise ValueError(f"rgb maps and target must agree in shape: {tuple(a.shape)}, {tuple(b.shape)}, "
                             f"{tuple(target.shape)}")
        dev = a.device
        losses = out if out is not None else torch.empty((3,), dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        da = torch.empty_like(a) if need else None
        db = torch.empty_like(b) if need else None
        with torch.cuda.device(dev):
            check(lib.nerf_mse_pair(a.numel(), _p(a), _p(b), _p(target), _p(losses), _p(da), _p(db), _stream(dev)),
                  "nerf_mse_pair")
        ctx.save_for_backward(da, db)
        total, la, lb = losses[0], losses[1], losses[2]
        ctx.mark_non_differentiable(la, lb)
        return total, la, lb

    @staticmethod
    def backward(ctx, g_total, _ga, _gb):
        da, db = ctx.saved_tensors
        return g_total * da, g_total * db, None, None


def mse_pair(rgb_coarse: torch.Tensor, rgb_fine: torch.Tensor, target: torch.Tensor,
             out: Optional[torch.Tensor] = None):
    """Both photometric losses and their sum in one launch: (total, loss_coarse, loss_fine), scalars
    that are views of one 3-float tensor (``out`` if given).  Gradients flow through ``total``."""
    return _MsePair.apply(rgb_coarse, rgb_fine, target, out)


@torch.no_grad()
def mse_pair_grads(rgb_coarse: torch.Tensor, rgb_fine: torch.Tensor, target: torch.Tensor,
                   out: Optional[torch.Tensor] = None):
    """The same launch outside autograd: ((total, loss_coarse, loss_fine), d total/d rgb_coarse,
    d total/d rgb_fine).  ``rgb_fine=None`` is the coarse-only loss (total = loss_coarse, loss_fine = 0).  ``torch.autograd.backward([rgb_coarse, rgb_fine], [d_coarse, d_fine])`` is then
    ``total.backward()`` without the unit root gradient and the two products with it."""
    a, target = _gpu_f32(rgb_coarse, "rgb_coarse"), _gpu_f32(target, "target_rgb")
    b = None if rgb_fine is None else _gpu_f32(rgb_fine, "rgb_fine")     # coarse-only run: one loss
    if a.shape != target.shape or (b is not None and b.shape != target.shape):
        raise ValueError(f"rgb maps and target must agree in shape: {tuple(a.shape)}, "
                         f"{None if b is None else tuple(b.shape)}, {tuple(target.shape)}")
    dev = a.device
    losses = out if out is not None else torch.empty((3,), dtype=torch.float32, device=dev)
    da, db = torch.empty_like(a), (None if b is None else torch.empty_like(b))
    with torch.cuda.device(dev):
        check(lib.nerf_mse_pair(a.numel(), _p(a), _p(b), _p(target), _p(losses), _p(da), _p(db), _stream(dev)),
              "nerf_mse_pair")
    return (losses[0], losses[1], losses[2]), da, db


def composite(raw: torch.Tensor, z_vals: torch.Tensor, white_bkgd: bool = False):
    return _Composite.apply(raw, z_vals, white_bkgd)


# ---------------------------------------------------------------------------
# hierarchical sampling
# ---------------------------------------------------------------------------

def _rows(t: torch.Tensor, name: str) -> Tuple[torch.Tensor, int]:
    """2-D fp32 GPU tensor with unit inner stride; returns (tensor, row stride)."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); nerf_amd has no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if t.dim() != 2:
        raise ValueError(f"{name} must be 2-D [N_rays, n], got {tuple(t.shape)}")
    if t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t, t.stride(0)


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, N_samples: int, det: bool = False, *,
               u: Optional[torch.Tensor] = None, return_inds: bool = False):
    bins, bld = _rows(bins, "bins")
    weights, wld = _rows(weights, "weights")
    N, nb = bins.shape
    if weights.shape[0] != N or weights.shape[1] != nb - 1:
        raise ValueError(
            f"sample_pdf needs len(bins) == len(weights) + 1 per ray (got bins {tuple(bins.shape)}, "
            f"weights {tuple(weights.shape)}); the CDF has one more knot than there are weights")
    dev = bins.device
    if u is None and not det:
        u = torch.rand((N, N_samples), dtype=torch.float32, device=dev)
    if u is not None:
        u = _gpu_f32(u, "u")
        if tuple(u.shape) != (N, N_samples):
            raise ValueError(f"u must be [{N}, {N_samples}], got {tuple(u.shape)}")
    samples = torch.empty((N, N_samples), dtype=torch.float32, device=dev)
    inds = torch.empty((N, N_samples), dtype=torch.int32, device=dev) if return_inds else None
    with torch.cuda.device(dev):
        check(lib.nerf_sample_pdf(N, nb, _p(bins), bld, _p(weights), wld, int(N_samples), _p(u), _p(samples),
                                  _p(inds), _stream(dev)), "nerf_sample_pdf")
    return (samples, inds) if return_inds else samples


def sort_rows(a: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row-wise ascending sort of cat([a, b], -1)."""
    a = _gpu_f32(a, "a")
    nb = 0
    if b is not None:
        b = _gpu_f32(b, "b")
        nb = b.shape[1]
    N, na = a.shape
    out = torch.empty((N, na + nb), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        check(lib.nerf_sort_rows(N, na, _p(a), nb, _p(b), _p(out), _stream(a.device)), "nerf_sort_rows")
    return out


def sample_merge(z_coarse: torch.Tensor, weights_coarse: torch.Tensor, N_fine: int, det: bool, *,
                 u: Optional[torch.Tensor] = None, return_fine: bool = False):
    """Coarse->fine glue in one launch: sort(cat(z_coarse, sample_pdf(mid(z), w[1:-1]+1e-5)))."""
    zc = _gpu_f32(z_coarse.detach(), "z_coarse")
    wc = _gpu_f32(weights_coarse.detach(), "weights_coarse")
    N, Sc = zc.shape
    if wc.shape != zc.shape:
        raise ValueError("z_coarse and weights_coarse must have the same shape")
    dev = zc.device
    if u is None and not det:
        u = torch.rand((N, N_fine), dtype=torch.float32, device=dev)
    if u is not None:
        u = _gpu_f32(u, "u")
        if tuple(u.shape) != (N, N_fine):
            raise ValueError(f"u must be [{N}, {N_fine}], got {tuple(u.shape)}")
    out = torch.empty((N, Sc + N_fine), dtype=torch.float32, device=dev)
    zf = torch.empty((N, N_fine), dtype=torch.float32, device=dev) if return_fine else None
    with torch.cuda.device(dev), _timed("sample_merge_det" if u is None else "sample_merge_u", dev, N):
        check(lib.nerf_sample_merge(N, Sc, _p(zc), _p(wc), int(N_fine), _p(u), _p(out), _p(zf), _stream(dev)),
              "nerf_sample_merge")
    return (out, zf) if return_fine else out


# ---------------------------------------------------------------------------
# fused MLP
# ---------------------------------------------------------------------------

PACK_TRAIN, PACK_INFER = 1, 2


def _prec(precision: str, viewdirs: bool) -> int:
    """The `precision` argument of the C entry points: operand type plus the architecture flag."""
    return _C.PREC[precision] | (0 if viewdirs else _C.ARCH_NOVIEW)


def mlp_pack(flat_params: torch.Tensor, precision: str, streams: int = PACK_TRAIN | PACK_INFER,
             viewdirs: bool = True) -> torch.Tensor:
    """Weight streams of one network.  ``streams`` limits the work to what the caller will run:
    PACK_TRAIN (forward that keeps activations, encoded-input forward, backward) and / or PACK_INFER
    (the no_grad ray forward); streams that are not written hold garbage.  ``viewdirs=False``: the
    parameters are those of ``NeRF(use_viewdirs=False)``."""
    flat = _gpu_f32(flat_params, "flat_params")
    n = _C.NERF_MLP_NPARAMS if viewdirs else _C.NERF_MLP_NPARAMS_NOVIEW
    if flat.numel() != n:
        raise ValueError(f"expected {n} parameters, got {flat.numel()}")
    packed = torch.empty((int(lib.nerf_mlp_packed_bytes()),), dtype=torch.uint8, device=flat.device)
    with torch.cuda.device(flat.device):
        check(lib.nerf_mlp_pack(_p(flat), _prec(precision, viewdirs), int(streams), _p(packed), _stream(flat.device)),
              "nerf_mlp_pack")
    return packed


class _FusedMLP(torch.autograd.Function):
    """(rays_o, rays_d, z) or (x_pts, x_views) + 24 parameter tensors -> raw [M,4]."""

    @staticmethod
    def forward(ctx, mode, a0, a1, a2, precision, packed, need_grad, viewdirs, *params):
        dev = a0.device
        if mode == "rays":
            rays_o, rays_d, z = _gpu_f32(a0, "rays_o"), _gpu_f32(a1, "rays_d"), _gpu_f32(a2, "z_vals")
            N, S = z.shape
            M = N * S
            out_shape = (N, S, 4)
        else:
            x_pts = _gpu_f32(a0, "x_pts_encoded")
            x_views = _gpu_f32(a1, "x_views_encoded") if viewdirs else None
            M = x_pts.numel() // 63
            out_shape = x_pts.shape[:-1] + (4,)
        raw = torch.empty(out_shape, dtype=torch.float32, device=dev)
        saved = None
        if need_grad:
            saved = torch.empty((int(lib.nerf_mlp_saved_bytes(M)),), dtype=torch.uint8, device=dev)
        prec = _prec(precision, viewdirs)
        with torch.cuda.device(dev):
            if mode == "rays":
                with _timed("mlp_fwd_train" if need_grad else "mlp_fwd", dev, M):
                    check(lib.nerf_mlp_fwd(N, S, _p(rays_o), _p(rays_d), _p(z), _p(packed), prec, _p(raw), _p(saved),
                                           _stream(dev)), "nerf_mlp_fwd")
            else:
                work = None if need_grad else torch.empty((int(lib.nerf_mlp_encoded_workspace_bytes(M)),),
                                                          dtype=torch.uint8, device=dev)
                check(lib.nerf_mlp_fwd_encoded(M, _p(x_pts), _p(x_views), _p(packed), prec, _p(raw), _p(saved),
                                               _p(work), _stream(dev)), "nerf_mlp_fwd_encoded")
        ctx.M = M
        ctx.prec = prec
        ctx.n_params = _C.NERF_MLP_NPARAMS if viewdirs else _C.NERF_MLP_NPARAMS_NOVIEW
        ctx.fp16 = precision == "fp16"
        ctx.shapes = [p.shape for p in params]
        ctx.packed = packed
        ctx.saved_acts = saved
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        if ctx.saved_acts is None:
            raise RuntimeError("fused MLP forward ran without saving activations (no parameter required grad)")
        d_raw = _gpu_f32(d_raw, "d_raw")
        dev = d_raw.device
        M = ctx.M
        dz = torch.empty((int(lib.nerf_mlp_dz_bytes(M)),), dtype=torch.uint8, device=dev)
        scale = None
        if ctx.fp16 and M > 0:
            # keep gradients inside the fp16 range: a power-of-two scale that puts max|d_raw| near 2^10
            amax = d_raw.abs().amax().clamp_min(1e-30)
            scale = torch.exp2(torch.floor(torch.log2(1024.0 / amax)).clamp(-24.0, 24.0)).reshape(1)
        with torch.cuda.device(dev):
            st = _stream(dev)
            with _timed("mlp_bwd_data", dev, M):
                check(lib.nerf_mlp_bwd_data(M, _p(d_raw), _p(ctx.packed), ctx.prec, _p(ctx.saved_acts), _p(scale),
                                            _p(dz), st), "nerf_mlp_bwd_data")
            sink = None if GRAD_SINKS is None else GRAD_SINKS.get(ctx.packed.data_ptr())
            if sink is not None and DEFERRED_WGRAD is None:
                # the weight gradients go straight into the caller's buffer (a slice of an optimizer's bucket)
                part_bytes = int(lib.nerf_mlp_partial_bytes(M))
                part = torch.empty((part_bytes,), dtype=torch.uint8, device=dev)
                with _timed("mlp_bwd_weights", dev, M):
                    check(lib.nerf_mlp_bwd_weights(M, _p(ctx.packed), _p(ctx.saved_acts), _p(dz), ctx.prec, _p(scale), _p(part),
                                                   part_bytes, _p(sink), st), "nerf_mlp_bwd_weights")
                ctx.saved_acts = None
                return (None,) * (8 + len(ctx.shapes))
            if DEFERRED_WGRAD is not None:
                # the caller runs the weight gradients of all its networks in one launch (mlp_bwd_weights_multi)
                DEFERRED_WGRAD.append(dict(M=M, packed=ctx.packed, saved=ctx.saved_acts, dz=dz, prec=ctx.prec, scale=scale,
                                           n_params=ctx.n_params, shapes=ctx.shapes))
                ctx.saved_acts = None
                return (None,) * (8 + len(ctx.shapes))
            part_bytes = int(lib.nerf_mlp_partial_bytes(M))
            part = torch.empty((part_bytes,), dtype=torch.uint8, device=dev)
            d_flat = torch.empty((ctx.n_params,), dtype=torch.float32, device=dev)
            with _timed("mlp_bwd_weights", dev, M):
                check(lib.nerf_mlp_bwd_weights(M, _p(ctx.packed), _p(ctx.saved_acts), _p(dz), ctx.prec, _p(scale), _p(part),
                                               part_bytes, _p(d_flat), st), "nerf_mlp_bwd_weights")
        ctx.saved_acts = None
        grads = []
        off = 0
        for shp in ctx.shapes:
            n = 1
            for s in shp:
                n *= s
            grads.append(d_flat[off:off + n].view(shp))
            off += n
        return (None, None, None, None, None, None, None, None, *grads)


# When this is a list, _FusedMLP.backward stops after the backward-data pass and appends what the weight-gradient
# pass needs; mlp_bwd_weights_multi() then serves all entries in one launch sequence.
DEFERRED_WGRAD = None
# {data_ptr of a network's packed weights: flat fp32 tensor}: _FusedMLP.backward writes that network's weight
# gradients there and returns no gradients to autograd.
GRAD_SINKS = None


def mlp_bwd_weights_multi(entries, d_params: Sequence[torch.Tensor]) -> None:
    """Weight gradients of the deferred networks ``entries`` (at most two) into the flat fp32 tensors ``d_params``
    (one per entry, e.g. slices of an optimizer's gradient bucket)."""
    if not entries:
        return
    if len(entries) > _C.WGRAD_MAX_NETS or len(entries) != len(d_params):
        raise ValueError("mlp_bwd_weights_multi takes at most two networks and one gradient tensor per network")
    dev = entries[0]["dz"].device
    nets = (_C.WgradNet * len(entries))()
    for n, (e, d) in enumerate(zip(entries, d_params)):
        if not (d.is_cuda and d.dtype == torch.float32 and d.is_contiguous() and d.numel() == e["n_params"]):
            raise RuntimeError(f"gradient tensor {n} must be a contiguous fp32 GPU tensor of {e['n_params']} elements")
        nets[n].M, nets[n].packed, nets[n].saved, nets[n].dz = e["M"], e["packed"].data_ptr(), e["saved"].data_ptr(), e["dz"].data_ptr()
        nets[n].precision = e["prec"]
        nets[n].grad_scale = None if e["scale"] is None else e["scale"].data_ptr()
        nets[n].d_params = d.data_ptr()
    nbytes = int(lib.nerf_mlp_bwd_weights_multi_bytes(len(entries), ctypes.cast(nets, ctypes.c_void_p)))
    if nbytes < 0:
        raise ValueError("mlp_bwd_weights_multi: the networks must share one operand type")
    part = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _timed("mlp_bwd_weights", dev, sum(e["M"] for e in entries)):
        check(lib.nerf_mlp_bwd_weights_multi(len(entries), ctypes.cast(nets, ctypes.c_void_p), _p(part), nbytes, _stream(dev)),
              "nerf_mlp_bwd_weights_multi")


def mlp_pack_multi(flat_params: Sequence[torch.Tensor], precision: str, streams: int, viewdirs: Sequence[bool],
                   out: Optional[Sequence[torch.Tensor]] = None):
    """``mlp_pack`` for up to two networks in one launch; ``out``: packed buffers to overwrite."""
    n = len(flat_params)
    if not 1 <= n <= _C.WGRAD_MAX_NETS or len(viewdirs) != n:
        raise ValueError("mlp_pack_multi takes one or two networks")
    flats = [_gpu_f32(f, "flat_params") for f in flat_params]
    for f, v in zip(flats, viewdirs):
        want = _C.NERF_MLP_NPARAMS if v else _C.NERF_MLP_NPARAMS_NOVIEW
        if f.numel() != want:
            raise ValueError(f"expected {want} parameters, got {f.numel()}")
    dev = flats[0].device
    nb = int(lib.nerf_mlp_packed_bytes())
    if out is None:
        out = [torch.empty((nb,), dtype=torch.uint8, device=dev) for _ in range(n)]
    P = (ctypes.c_void_p * n)(*[f.data_ptr() for f in flats])
    S = (ctypes.c_int * n)(*[int(streams) | (0 if v else 4) for v in viewdirs])
    O_ = (ctypes.c_void_p * n)(*[o.data_ptr() for o in out])
    with torch.cuda.device(dev):
        check(lib.nerf_mlp_pack_multi(n, ctypes.cast(P, ctypes.c_void_p), _C.PREC[precision], ctypes.cast(S, ctypes.c_void_p),
                                      ctypes.cast(O_, ctypes.c_void_p), _stream(dev)), "nerf_mlp_pack_multi")
    return list(out)


def fused_mlp_rays(rays_o, rays_d, z_vals, precision: str, packed: torch.Tensor, params: Sequence[torch.Tensor],
                   viewdirs: bool = True):
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _FusedMLP.apply("rays", rays_o, rays_d, z_vals, precision, packed, need_grad, viewdirs, *params)


def fused_mlp_encoded(x_pts, x_views, precision: str, packed: torch.Tensor, params: Sequence[torch.Tensor],
                      viewdirs: bool = True):
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _FusedMLP.apply("encoded", x_pts, x_views, None, precision, packed, need_grad, viewdirs, *params)


# ---------------------------------------------------------------------------
# fp32-operand layer programs (validation mode, any architecture within the program's limits)
# ---------------------------------------------------------------------------

def fp32_pack_segment(weight: torch.Tensor, col0: int, cols: int) -> torch.Tensor:
    """Columns [col0, col0+cols) of a [rows, ld] weight matrix in the fragment order of the fp32 kernel."""
    w = _gpu_f32(weight.detach(), "weight")
    rows, ld = w.shape
    out = torch.empty((int(lib.nerf_mlp_fp32_segment_floats(rows, cols)),), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        check(lib.nerf_mlp_fp32_pack_segment(_p(w), rows, ld, int(col0), int(cols), _p(out), _stream(w.device)),
              "nerf_mlp_fp32_pack_segment")
    return out


def fp32_mlp_forward(x_pts: torch.Tensor, x_views: Optional[torch.Tensor], layers) -> torch.Tensor:
    """raw [M, 4] of the network described by ``layers`` (a ctypes array of ``_C.Fp32Layer``; the tensors its
    pointers refer to must be kept alive by the caller) on encoded inputs [M, cp] and [M, cv] or None."""
    xp = _gpu_f32(x_pts, "x_pts_encoded")
    M, cp = xp.shape
    xv, cv = None, 0
    if x_views is not None:
        xv = _gpu_f32(x_views, "x_views_encoded")
        if xv.shape[0] != M:
            raise ValueError("x_pts_encoded and x_views_encoded must have the same number of rows")
        cv = xv.shape[1]
    raw = torch.empty((M, 4), dtype=torch.float32, device=xp.device)
    with torch.cuda.device(xp.device), _timed("mlp_fwd_fp32", xp.device, M):
        check(lib.nerf_mlp_fp32_fwd(M, _p(xp), cp, _p(xv), cv, ctypes.cast(layers, ctypes.c_void_p), len(layers), _p(raw),
                                    _stream(xp.device)), "nerf_mlp_fp32_fwd")
    return raw


def fp32_mlp_forward_train(x_pts: torch.Tensor, x_views: Optional[torch.Tensor], layers):
    """``fp32_mlp_forward`` that also keeps the encoded inputs and every hidden layer's output for
    ``fp32_mlp_backward``: returns (raw [M, 4], saved)."""
    xp = _gpu_f32(x_pts, "x_pts_encoded")
    M, cp = xp.shape
    xv, cv = None, 0
    if x_views is not None:
        xv = _gpu_f32(x_views, "x_views_encoded")
        if xv.shape[0] != M:
            raise ValueError("x_pts_encoded and x_views_encoded must have the same number of rows")
        cv = xv.shape[1]
    lp = ctypes.cast(layers, ctypes.c_void_p)
    n_saved = int(lib.nerf_mlp_fp32_saved_floats(M, cp, cv, lp, len(layers)))
    if n_saved < 0:
        raise ValueError("fp32_mlp_forward_train: bad layer list")
    raw = torch.empty((M, 4), dtype=torch.float32, device=xp.device)
    saved = torch.empty((max(n_saved, 4),), dtype=torch.float32, device=xp.device)
    with torch.cuda.device(xp.device), _timed("mlp_fwd_fp32_train", xp.device, M):
        check(lib.nerf_mlp_fp32_fwd_train(M, _p(xp), cp, _p(xv), cv, lp, len(layers), _p(raw), _p(saved), _stream(xp.device)),
              "nerf_mlp_fp32_fwd_train")
    return raw, saved


def fp32_mlp_backward(M: int, cp: int, cv: int, layers, grads, d_raw: torch.Tensor, saved: torch.Tensor) -> None:
    """Weight and bias gradients of the network ``layers`` describes into the tensors ``grads`` (a ctypes array of
    ``_C.Fp32Grad``) points to, from ``d_raw`` [M, 4] and what ``fp32_mlp_forward_train`` kept."""
    d_raw = _gpu_f32(d_raw, "d_raw")
    if tuple(d_raw.shape) != (M, 4):
        raise ValueError(f"d_raw must be [{M}, 4], got {tuple(d_raw.shape)}")
    dev = d_raw.device
    lp, gp = ctypes.cast(layers, ctypes.c_void_p), ctypes.cast(grads, ctypes.c_void_p)
    n_work = int(lib.nerf_mlp_fp32_bwd_workspace_floats(M, cp, cv, lp, len(layers)))
    work = torch.empty((max(n_work, 4),), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _timed("mlp_bwd_fp32", dev, M):
        check(lib.nerf_mlp_fp32_bwd(M, cp, cv, lp, gp, len(layers), _p(d_raw), _p(saved), _p(work), work.numel(), _stream(dev)),
              "nerf_mlp_fp32_bwd")


# ---------------------------------------------------------------------------
# 16-bit-operand layer programs (any architecture within the program's limits, precision 'bf16' / 'fp16')
# ---------------------------------------------------------------------------

def p16_pack_segment(weight: torch.Tensor, col0: int, cols: int, precision: str) -> torch.Tensor:
    """Columns [col0, col0+cols) of a [rows, ld] weight matrix in the fragment order of the 16-bit program kernel."""
    w = _gpu_f32(weight.detach(), "weight")
    rows, ld = w.shape
    out = torch.empty((int(lib.nerf_mlp_p16_segment_bytes(rows, cols)),), dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        check(lib.nerf_mlp_p16_pack_segment(_p(w), rows, ld, int(col0), int(cols), _C.PREC[precision], _p(out),
                                            _stream(w.device)), "nerf_mlp_p16_pack_segment")
    return out


def _encoded_pair(x_pts, x_views):
    xp = _gpu_f32(x_pts, "x_pts_encoded")
    M, cp = xp.shape
    xv, cv = None, 0
    if x_views is not None:
        xv = _gpu_f32(x_views, "x_views_encoded")
        if xv.shape[0] != M:
            raise ValueError("x_pts_encoded and x_views_encoded must have the same number of rows")
        cv = xv.shape[1]
    return xp, xv, M, cp, cv


def p16_mlp_forward(x_pts: torch.Tensor, x_views: Optional[torch.Tensor], layers, precision: str) -> torch.Tensor:
    """raw [M, 4] of the network ``layers`` describes (``_C.Fp32Layer`` array whose weights were packed by
    ``p16_pack_segment``) with 16-bit operands."""
    xp, xv, M, cp, cv = _encoded_pair(x_pts, x_views)
    raw = torch.empty((M, 4), dtype=torch.float32, device=xp.device)
    with torch.cuda.device(xp.device), _timed("mlp_fwd_p16", xp.device, M):
        check(lib.nerf_mlp_p16_fwd(M, _p(xp), cp, _p(xv), cv, ctypes.cast(layers, ctypes.c_void_p), len(layers),
                                   _C.PREC[precision], _p(raw), _stream(xp.device)), "nerf_mlp_p16_fwd")
    return raw


def p16_mlp_forward_train(x_pts: torch.Tensor, x_views: Optional[torch.Tensor], layers, precision: str):
    """``p16_mlp_forward`` that also keeps the encoded inputs and every hidden layer's output (in the operand
    type) for ``p16_mlp_backward``: returns (raw [M, 4], saved)."""
    xp, xv, M, cp, cv = _encoded_pair(x_pts, x_views)
    lp = ctypes.cast(layers, ctypes.c_void_p)
    n_saved = int(lib.nerf_mlp_p16_saved_bytes(M, cp, cv, lp, len(layers)))
    if n_saved < 0:
        raise ValueError("p16_mlp_forward_train: bad layer list")
    raw = torch.empty((M, 4), dtype=torch.float32, device=xp.device)
    saved = torch.empty((max(n_saved, 16),), dtype=torch.uint8, device=xp.device)
    with torch.cuda.device(xp.device), _timed("mlp_fwd_p16_train", xp.device, M):
        check(lib.nerf_mlp_p16_fwd_train(M, _p(xp), cp, _p(xv), cv, lp, len(layers), _C.PREC[precision], _p(raw), _p(saved),
                                         _stream(xp.device)), "nerf_mlp_p16_fwd_train")
    return raw, saved


def p16_mlp_backward(M: int, cp: int, cv: int, layers, grads, d_raw: torch.Tensor, saved: torch.Tensor,
                     precision: str) -> None:
    """Weight and bias gradients of the network ``layers`` describes into the tensors ``grads`` (``_C.Fp32Grad``
    array) points to, from ``d_raw`` [M, 4] and what ``p16_mlp_forward_train`` kept.  In fp16 the gradients are
    scaled by a power of two that puts max|d_raw| near 2^10 and scaled back at the end."""
    d_raw = _gpu_f32(d_raw, "d_raw")
    if tuple(d_raw.shape) != (M, 4):
        raise ValueError(f"d_raw must be [{M}, 4], got {tuple(d_raw.shape)}")
    dev = d_raw.device
    scale = None
    if precision == "fp16" and M > 0:
        amax = d_raw.abs().amax().clamp_min(1e-30)
        scale = torch.exp2(torch.floor(torch.log2(1024.0 / amax)).clamp(-24.0, 24.0)).reshape(1)
    lp, gp = ctypes.cast(layers, ctypes.c_void_p), ctypes.cast(grads, ctypes.c_void_p)
    n_work = int(lib.nerf_mlp_p16_bwd_workspace_bytes(M, cp, cv, lp, len(layers)))
    work = torch.empty((max(n_work, 256),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _timed("mlp_bwd_p16", dev, M):
        check(lib.nerf_mlp_p16_bwd(M, cp, cv, lp, gp, len(layers), _C.PREC[precision], _p(d_raw), _p(saved), _p(scale),
                                   _p(work), work.numel(), _stream(dev)), "nerf_mlp_p16_bwd")


# ---------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------

def adam_step(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
              lr: float, beta1: float, beta2: float, eps: float, step: int, grad_scale: float = 1.0) -> None:
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f"{nm} must be a contiguous fp32 GPU tensor")
    n = params.numel()
    if not (grads.numel() == n and exp_avg.numel() == n and exp_avg_sq.numel() == n):
        raise ValueError("adam_step: size mismatch")
    with torch.cuda.device(params.device):
        check(lib.nerf_adam_step(n, _p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), float(lr), float(beta1),
                                 float(beta2), float(eps), int(step), float(grad_scale), _stream(params.device)),
              "nerf_adam_step")


def adam_step_dev(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                  state: torch.Tensor) -> None:
    """Adam with hyper-parameters in the device tensor ``state`` = [lr, beta1, beta2, eps, grad_scale, step, ticket, -]
    (the ticket, a workgroup counter, must be 0 before the first call; every call leaves it 0)."""
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"),
                  (state, "state")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f"{nm} must be a contiguous fp32 GPU tensor")
    n = params.numel()
    if not (grads.numel() == n and exp_avg.numel() == n and exp_avg_sq.numel() == n and state.numel() >= 8):
        raise ValueError("adam_step_dev: size mismatch")
    with torch.cuda.device(params.device):
        check(lib.nerf_adam_step_dev(n, _p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), _p(state),
                                     _stream(params.device)), "nerf_adam_step_dev")
