"""Drop-in replacements for the reference's model/render functions, backed by HIP kernels.

Mirrors the call surface of the reference's ``src/nerf_project/models/nerf.py``:
``PositionalEncoder`` (:9-76), ``NeRF`` (:79-209), ``sample_pdf`` (:213-315) and
``render_rays`` (:318-483), with the same positional/keyword signatures, the same
``state_dict`` keys and the same returned dict.  Keyword-only additions:
``u=`` on ``sample_pdf`` and ``t_rand=`` on ``render_rays`` inject the uniforms the
reference draws with ``torch.rand``.

All tensors must live on the GPU; there is no CPU path.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import ops

_PRECISIONS = ("bf16", "fp16", "fp32")
_FUSED_PRECISIONS = ("bf16", "fp16")


class PositionalEncoder(nn.Module):
    """gamma(x) = [x, sin(f0 x), cos(f0 x), ..., sin(f_{L-1} x), cos(f_{L-1} x)], f_l = 2^l (no pi).

    Same constructor, attributes and buffer as the reference (nerf.py:24-53).
    """

    def __init__(self, input_dims: int, num_freqs: int, log_sampling: bool = True):
        super().__init__()
        self.input_dims = input_dims
        self.num_freqs = num_freqs
        self.log_sampling = log_sampling
        if log_sampling:
            fb = 2.0 ** torch.linspace(0.0, num_freqs - 1, num_freqs)
        else:
            fb = torch.linspace(1.0, 2.0 ** (num_freqs - 1), num_freqs)
        self.register_buffer("freq_bands", fb)
        self.output_dims = self.input_dims * (1 + 2 * self.num_freqs)

    def is_standard(self, num_freqs: int) -> bool:
        """True when this encoder is the one the fused MLP kernel generates in-register."""
        return self.input_dims == 3 and self.num_freqs == num_freqs and self.log_sampling

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[-1] != self.input_dims:
            raise ValueError(f"expected last dim {self.input_dims}, got {x.shape[-1]}")
        return ops.posenc(x, self.freq_bands)


def _consecutive_view(ps) -> Optional[torch.Tensor]:
    """The parameters as one flat tensor without copying, when they already lie back to back in one
    storage (as they do once ``FlatAdam`` has moved them into its bucket); else None."""
    store = ps[0].untyped_storage().data_ptr()
    ptr, total = ps[0].data_ptr(), 0
    for p in ps:
        if (not p.is_contiguous() or p.dtype != torch.float32 or p.untyped_storage().data_ptr() != store
                or p.data_ptr() != ptr + 4 * total):
            return None
        total += p.numel()
    return ps[0].detach().as_strided((total,), (1,), ps[0].storage_offset())


class NeRF(nn.Module):
    """The NeRF MLP with the reference's parameters (nerf.py:126-153) and forward (nerf.py:173-209).

    The fused HIP kernels implement the architecture the reference's scripts instantiate
    (D=8, W=256, input_ch_pts=63, input_ch_views=27, skips=[4], use_viewdirs=True).

    ``precision`` selects the MFMA operand type ('bf16' or 'fp16'); accumulation is fp32.
    ``precision='fp32'`` is the validation mode: the layers run one by one as fp32 library GEMMs
    (rocBLAS through ``torch.nn.functional.linear``) under plain autograd; the encodings and
    compositing around it are still the HIP kernels.  An architecture outside the fused kernels
    (other D / W / skips / encodings) runs in that fp32 mode whatever ``precision`` says
    (``effective_precision()`` reports it).
    """

    def __init__(self, D: int = 8, W: int = 256, input_ch_pts: int = 63, input_ch_views: int = 27,
                 skips: List[int] = [4], use_viewdirs: bool = True, precision: str = "bf16"):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch_pts, self.input_ch_views = input_ch_pts, input_ch_views
        self.skips = skips
        self.use_viewdirs = use_viewdirs
        self.precision = precision
        self.pts_linears = nn.ModuleList([nn.Linear(input_ch_pts, W)])
        for i in range(D - 1):
            self.pts_linears.append(nn.Linear(W + input_ch_pts if i in self.skips else W, W))
        if use_viewdirs:
            self.sigma_linear = nn.Linear(W, 1)
            self.feature_linear = nn.Linear(W, W)
            self.rgb_layers = nn.Sequential(nn.Linear(W + input_ch_views, W // 2), nn.ReLU(), nn.Linear(W // 2, 3))
        else:
            self.output_linear = nn.Linear(W, 4)
        self._pack_key = None
        self._packed = None

    # -- kernel plumbing ------------------------------------------------------------------
    def kernel_supported(self) -> bool:
        return (self.D == 8 and self.W == 256 and self.input_ch_pts == 63 and self.input_ch_views == 27
                and list(self.skips) == [4] and self.use_viewdirs)

    def fused(self) -> bool:
        """True when calls go to the fused 16-bit MFMA kernels."""
        if self.precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {_PRECISIONS}, got {self.precision!r}")
        return self.precision in _FUSED_PRECISIONS and self.kernel_supported()

    def effective_precision(self) -> str:
        """The precision calls actually run in: ``precision``, or 'fp32' for an architecture outside the fused kernels."""
        return self.precision if self.fused() else "fp32"

    def _require_supported(self):
        if not self.kernel_supported():
            raise NotImplementedError(
                "the HIP MLP kernels implement D=8, W=256, input_ch_pts=63, input_ch_views=27, skips=[4], "
                f"use_viewdirs=True; got D={self.D}, W={self.W}, pts={self.input_ch_pts}, "
                f"views={self.input_ch_views}, skips={list(self.skips)}, use_viewdirs={self.use_viewdirs}")
        if self.precision not in _FUSED_PRECISIONS:
            raise ValueError(f"the fused kernels take precision 'bf16' or 'fp16', got {self.precision!r}")

    def ordered_params(self) -> List[torch.Tensor]:
        """Parameters in state_dict order (the order of the flat buffer the kernels index)."""
        return list(self.parameters())

    def invalidate_packed(self) -> None:
        """Forget the cached packed weights.  Needed after the weights were rewritten on the device behind
        PyTorch's back (a HIP-graph replay of an optimizer step, a collective on the optimizer's flat
        bucket): such writes do not advance the version counters the cache is keyed on."""
        self._pack_key = None

    def packed_weights(self) -> torch.Tensor:
        """MFMA fragment streams of the current weights.  While gradients are being recorded the
        weights are packed on every call (one small kernel; it also keeps the call sequence identical
        from step to step, which HIP-graph capture of a training step relies on); under no_grad the
        packed buffer is cached and re-made only when a parameter was modified or replaced."""
        self._require_supported()
        ps = self.ordered_params()
        training = torch.is_grad_enabled() and any(p.requires_grad for p in ps)
        key = None if training else (self.precision,) + tuple((p.data_ptr(), p._version) for p in ps)
        if training or key != self._pack_key:
            with torch.no_grad():
                flat = _consecutive_view(ps)
                if flat is None:
                    flat = torch.cat([p.detach().reshape(-1) for p in ps])
                # a training step only ever runs the training kernels on this snapshot of the weights
                self._packed = ops.mlp_pack(flat, self.precision,
                                            ops.PACK_TRAIN if training else ops.PACK_TRAIN | ops.PACK_INFER)
            self._pack_key = key
        return self._packed

    def forward_rays(self, rays_o: torch.Tensor, rays_d: torch.Tensor, z_vals: torch.Tensor) -> torch.Tensor:
        """Fused sample placement on the ray, both encodings and the MLP: raw [N_rays, S, 4]."""
        if not self.fused():
            raise NotImplementedError("forward_rays is the fused 16-bit path; in fp32 mode call render_rays or forward")
        return ops.fused_mlp_rays(rays_o, rays_d, z_vals, self.precision, self.packed_weights(),
                                  self.ordered_params())

    def forward(self, x_pts_encoded: torch.Tensor, x_views_encoded: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.use_viewdirs and x_views_encoded is None:
            raise ValueError("View directions (x_views_encoded) must be provided when use_viewdirs is True.")
        if not self.fused():
            return self._forward_fp32(x_pts_encoded, x_views_encoded)
        if x_pts_encoded.shape[-1] != 63 or x_views_encoded.shape[-1] != 27:
            raise ValueError("expected encoded inputs of width 63 and 27")
        if x_pts_encoded.shape[:-1] != x_views_encoded.shape[:-1]:
            raise ValueError("x_pts_encoded and x_views_encoded must agree on the leading dimensions")
        if x_pts_encoded.requires_grad or x_views_encoded.requires_grad:
            raise NotImplementedError("gradients with respect to the encoded inputs are not implemented")
        return ops.fused_mlp_encoded(x_pts_encoded, x_views_encoded, self.precision, self.packed_weights(),
                                     self.ordered_params())


    def _forward_fp32(self, x_pts_encoded: torch.Tensor, x_views_encoded: Optional[torch.Tensor]) -> torch.Tensor:
        """Layer-by-layer fp32 evaluation of nerf.py:173-209 (validation mode, any architecture)."""
        h = x_pts_encoded
        for i, layer in enumerate(self.pts_linears):
            h = torch.relu(layer(h))
            if i in self.skips:
                h = torch.cat((x_pts_encoded, h), dim=-1)
        if not self.use_viewdirs:
            return self.output_linear(h)
        sigma = self.sigma_linear(h)
        rgb = self.rgb_layers(torch.cat((self.feature_linear(h), x_views_encoded), dim=-1))
        return torch.cat((rgb, sigma), dim=-1)


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, N_samples: int, det: bool = False, *,
               u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inverse-CDF sampling (nerf.py:213-315).  ``bins`` needs one more entry per ray than
    ``weights`` (the CDF has len(weights)+1 knots and indexes ``bins``), as every real caller
    passes.  ``u`` [N_rays, N_samples] replaces the internal uniform draw."""
    return ops.sample_pdf(bins, weights, N_samples, det, u=u)


def render_rays(nerf_model, embed_fn_pts, embed_fn_views, rays_o: torch.Tensor, rays_d: torch.Tensor,
                near, far, N_samples: int, rand: bool = False, lindisp: bool = False,
                use_viewdirs: bool = True, white_bkgd: bool = False,
                z_vals_override: Optional[torch.Tensor] = None, *,
                t_rand: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Volumetric rendering of a batch of rays (nerf.py:318-483); returns the same 8-key dict.

    With this package's ``NeRF`` and standard ``PositionalEncoder``s the points, both
    encodings and the MLP run as one fused kernel; other callables are evaluated as given and
    only sample placement and compositing run on the HIP kernels.
    """
    if rays_o.dim() != 2 or rays_o.shape[-1] != 3 or rays_d.shape != rays_o.shape:
        raise RuntimeError(f"rays_o and rays_d must both be [N_rays, 3], got {tuple(rays_o.shape)} and "
                           f"{tuple(rays_d.shape)}")
    n_rays = rays_o.shape[0]
    dev = rays_o.device
    if rand and t_rand is None:
        S = z_vals_override.shape[-1] if z_vals_override is not None else N_samples
        t_rand = torch.rand((n_rays, S), dtype=torch.float32, device=dev)
    if not rand:
        t_rand = None
    if z_vals_override is not None and t_rand is None:
        z_vals = z_vals_override          # same aliasing as the reference (nerf.py:374, :480)
    else:
        z_vals = ops.place_samples(n_rays, near, far, N_samples, lindisp, dev, z_vals_override, t_rand)

    fused = (isinstance(nerf_model, NeRF) and nerf_model.fused() and use_viewdirs
             and isinstance(embed_fn_pts, PositionalEncoder) and embed_fn_pts.is_standard(10)
             and isinstance(embed_fn_views, PositionalEncoder) and embed_fn_views.is_standard(4))
    if fused:
        raw = nerf_model.forward_rays(rays_o, rays_d, z_vals)
    else:
        S = z_vals.shape[-1]
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        emb_pts = embed_fn_pts(pts.reshape(-1, 3))
        if use_viewdirs:
            vd = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
            emb_views = embed_fn_views(vd[:, None, :].expand_as(pts).reshape(-1, 3))
            raw = nerf_model(emb_pts, emb_views)
        else:
            raw = nerf_model(emb_pts)
        raw = raw.view(n_rays, S, raw.shape[-1])

    rgb_map, depth_map, disp_map, acc_map, weights, alpha = ops.composite(raw, z_vals, white_bkgd)
    return {"rgb_map": rgb_map, "depth_map": depth_map, "disp_map": disp_map, "acc_map": acc_map,
            "weights": weights, "z_vals": z_vals, "alpha": alpha, "raw_sigma": raw[..., 3]}


def render_hierarchical(model_coarse, model_fine, embed_fn_pts, embed_fn_views, rays_o, rays_d, near, far,
                        N_samples_coarse: int, N_samples_fine: int, *, rand: bool = False, det: bool = True,
                        lindisp: bool = False, use_viewdirs: bool = True, white_bkgd: bool = False,
                        t_rand_coarse=None, u_fine=None, t_rand_fine=None):
    """The coarse -> sample_pdf -> sort -> fine sequence that the reference's scripts repeat
    inline (scripts/train_nerf.py:293-329, scripts/render_nerf.py:275-303).  Returns
    (coarse_results, fine_results).  ``rand``/``det`` follow the scripts: training uses
    rand=True, det=False; rendering rand=False, det=True."""
    coarse = render_rays(model_coarse, embed_fn_pts, embed_fn_views, rays_o, rays_d, near, far, N_samples_coarse,
                         rand=rand, lindisp=lindisp, use_viewdirs=use_viewdirs, white_bkgd=white_bkgd,
                         t_rand=t_rand_coarse)
    with torch.no_grad():
        z_comb = ops.sample_merge(coarse["z_vals"], coarse["weights"], N_samples_fine, det, u=u_fine)
    fine = render_rays(model_fine, embed_fn_pts, embed_fn_views, rays_o, rays_d, near, far, z_comb.shape[-1],
                       rand=rand, lindisp=lindisp, use_viewdirs=use_viewdirs, white_bkgd=white_bkgd,
                       z_vals_override=z_comb, t_rand=t_rand_fine)
    return coarse, fine
