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

from .. import _C, ops

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

    The fused 16-bit HIP kernels implement the two networks the reference's scripts instantiate
    (train_nerf.py:189-200): D=8, W=256, input_ch_pts=63, skips=[4], with input_ch_views=27 and
    use_viewdirs=True, or with use_viewdirs=False (the script default).

    ``precision`` selects the MFMA operand type: 'bf16' or 'fp16' (fp32 accumulation), or 'fp32', the
    validation mode (exact fp32 products).  The two networks above in a 16-bit type run on the fused kernels.
    Everything else the constructor accepts -- any depth, width up to 256, any skips, and the two networks above
    in fp32 (``program_supported()``) -- runs on the layer-program kernels, which take the network as a list of
    layers, in the requested operand type: forward, backward-data chain and weight gradients are this library's
    MFMA kernels there as well.  Beyond those limits the layers are evaluated by library GEMMs in fp32
    (``effective_precision()`` reports it).  The encodings and compositing around the MLP are the same HIP
    kernels in every mode.
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
        self._prog = None
        self._external_pack = None

    # -- kernel plumbing ------------------------------------------------------------------
    def kernel_supported(self) -> bool:
        """The two networks the reference's scripts build (train_nerf.py:189-200): 8 x 256 with the skip after
        layer 4, with the view-direction head or (the script default) without it."""
        return (self.D == 8 and self.W == 256 and self.input_ch_pts == 63 and list(self.skips) == [4]
                and (self.input_ch_views == 27 or not self.use_viewdirs))

    def fused(self) -> bool:
        """True when calls go to the fused 16-bit MFMA kernels."""
        if self.precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {_PRECISIONS}, got {self.precision!r}")
        return self.precision in _FUSED_PRECISIONS and self.kernel_supported()

    def effective_precision(self) -> str:
        """The operand type calls actually run in: ``precision`` on the fused kernels and on the layer-program
        kernels, 'fp32' where only library GEMMs cover the architecture."""
        return self.precision if (self.fused() or self.program_supported()) else "fp32"

    def _require_supported(self):
        if not self.kernel_supported():
            raise NotImplementedError(
                "the 16-bit HIP MLP kernels implement D=8, W=256, input_ch_pts=63, skips=[4], with "
                f"input_ch_views=27 or use_viewdirs=False; got D={self.D}, W={self.W}, pts={self.input_ch_pts}, "
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
        if self._external_pack is not None:         # a training step packed all its networks in one launch
            return self._external_pack
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
                                            ops.PACK_TRAIN if training else ops.PACK_TRAIN | ops.PACK_INFER,
                                            viewdirs=self.use_viewdirs)
            self._pack_key = key
        return self._packed

    def forward_rays(self, rays_o: torch.Tensor, rays_d: torch.Tensor, z_vals: torch.Tensor) -> torch.Tensor:
        """Fused sample placement on the ray, both encodings and the MLP: raw [N_rays, S, 4]."""
        if not self.fused():
            raise NotImplementedError("forward_rays is the fused 16-bit path; in fp32 mode call render_rays or forward")
        return ops.fused_mlp_rays(rays_o, rays_d, z_vals, self.precision, self.packed_weights(),
                                  self.ordered_params(), viewdirs=self.use_viewdirs)

    def forward(self, x_pts_encoded: torch.Tensor, x_views_encoded: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.use_viewdirs and x_views_encoded is None:
            raise ValueError("View directions (x_views_encoded) must be provided when use_viewdirs is True.")
        if not self.fused():
            return self._forward_program(x_pts_encoded, x_views_encoded)
        if not self.use_viewdirs:
            x_views_encoded = None          # ignored, as nerf.py:202-206 does
        if x_pts_encoded.shape[-1] != 63 or (x_views_encoded is not None and x_views_encoded.shape[-1] != 27):
            raise ValueError("expected encoded inputs of width 63 and 27")
        if x_views_encoded is not None and x_pts_encoded.shape[:-1] != x_views_encoded.shape[:-1]:
            raise ValueError("x_pts_encoded and x_views_encoded must agree on the leading dimensions")
        if x_pts_encoded.requires_grad or (x_views_encoded is not None and x_views_encoded.requires_grad):
            raise NotImplementedError("gradients with respect to the encoded inputs are not implemented")
        return ops.fused_mlp_encoded(x_pts_encoded, x_views_encoded, self.precision, self.packed_weights(),
                                     self.ordered_params(), viewdirs=self.use_viewdirs)


    # -- layer-program kernels: fp32 validation mode, and 16-bit operands outside the fused architectures ----
    def program_supported(self) -> bool:
        """True when the layer-program MFMA kernels cover this architecture (their LDS-resident activations
        bound the widths; a skip connection after the last hidden layer is not a network the reference can run)."""
        n_layers = self.D + (4 if self.use_viewdirs else 1)
        return (1 <= self.W <= _C.FP32_MAX_WIDTH and 1 <= self.input_ch_pts <= _C.FP32_MAX_PTS
                and (not self.use_viewdirs or 1 <= self.input_ch_views <= _C.FP32_MAX_VIEWS)
                and n_layers <= _C.FP32_MAX_LAYERS and all(0 <= i < self.D - 1 for i in self.skips))

    fp32_kernel_supported = program_supported

    def program_mode(self) -> str:
        """Operand type of the layer-program kernels for this network: ``precision``."""
        if self.precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {_PRECISIONS}, got {self.precision!r}")
        return self.precision

    def _layer_program(self, mode: str, training: bool):
        """The layer list of the program kernels for the current weights in operand type ``mode``: (ctypes
        array, tensors it points to, per-layer description).  With ``training`` (the caller records a graph for
        backward) the weights are packed on every call -- which also keeps the call sequence identical from step to
        step, as HIP-graph capture of a training step relies on: a capture that hit the cache would replay stale
        packed weights; else the program is cached and re-made only when a parameter was modified or replaced.
        (The caller decides: inside ``autograd.Function.forward`` grad mode is always off.)"""
        ps = self.ordered_params()
        key = None if training else ("program", mode) + tuple((p.data_ptr(), p._version) for p in ps)
        if not training and key == self._pack_key and self._prog is not None:
            return self._prog
        keep, layers, spec = [], [], []
        kpad = 8 if mode == "fp32" else 16          # input features per weight chunk of the kernel
        pad8 = lambda v: (v + kpad - 1) // kpad * kpad
        pack = (lambda w, c0, c: ops.fp32_pack_segment(w, c0, c)) if mode == "fp32" else \
               (lambda w, c0, c: ops.p16_pack_segment(w, c0, c, mode))

        def add(lin, segments, relu, dst, dst_col=0):
            spec.append((lin, segments, dst, dst_col))
            L = _C.Fp32Layer()
            L.n_segments = len(segments)
            for s, (col0, cols, src) in enumerate(segments):
                w = pack(lin.weight, col0, cols)
                keep.append(w)
                L.weights[s], L.k_padded[s], L.src[s] = w.data_ptr(), pad8(cols), src
            b = lin.bias.detach().contiguous()
            keep.append(b)
            L.bias, L.out_features, L.relu, L.dst, L.dst_col = b.data_ptr(), lin.out_features, int(relu), dst, dst_col
            layers.append(L)

        with torch.no_grad():
            Cp, Cv, W = self.input_ch_pts, self.input_ch_views, self.W
            cur, other = _C.FP32_BUF_PTS, _C.FP32_BUF_A
            for i, lin in enumerate(self.pts_linears):
                if i > 0 and (i - 1) in self.skips:          # input is [encoded points, hidden] (nerf.py:184)
                    segs = [(0, Cp, _C.FP32_BUF_PTS), (Cp, W, cur)]
                else:
                    segs = [(0, lin.in_features, cur)]
                add(lin, segs, True, other)
                cur, other = other, (_C.FP32_BUF_B if other == _C.FP32_BUF_A else _C.FP32_BUF_A)
            if self.use_viewdirs:
                add(self.sigma_linear, [(0, W, cur)], False, _C.FP32_DST_RAW, 3)
                add(self.feature_linear, [(0, W, cur)], False, other)
                add(self.rgb_layers[0], [(0, W, other), (W, Cv, _C.FP32_BUF_VIEWS)], True, cur)
                add(self.rgb_layers[2], [(0, W // 2, cur)], False, _C.FP32_DST_RAW, 0)
            else:
                add(self.output_linear, [(0, W, cur)], False, _C.FP32_DST_RAW, 0)
        arr = (_C.Fp32Layer * len(layers))(*layers)
        self._prog = (arr, keep, spec)
        if not training:
            self._pack_key = key
        return self._prog

    @staticmethod
    def _transposed_weights(spec, mode: str):
        """Per layer and segment of ``spec`` (from ``_layer_program``) the packed TRANSPOSED weight block the
        backward-data chain multiplies with, or None for a segment read from the encoded inputs (nothing flows back
        there): the [cols, out] matrix W[:, col0:col0+cols]^T, for a layer that writes raw placed in columns
        dst_col.. of a [cols, 4] matrix (the chain reads d_raw [M, 4] as that layer's output gradient)."""
        pack = (lambda w, c0, c: ops.fp32_pack_segment(w, c0, c)) if mode == "fp32" else \
               (lambda w, c0, c: ops.p16_pack_segment(w, c0, c, mode))
        out = []
        with torch.no_grad():
            for lin, segments, dst, dst_col in spec:
                per_seg = []
                for col0, cols, src in segments:
                    if src in (_C.FP32_BUF_PTS, _C.FP32_BUF_VIEWS):
                        per_seg.append(None)
                        continue
                    wt = lin.weight.detach()[:, col0:col0 + cols].t()
                    if dst == _C.FP32_DST_RAW:
                        full = torch.zeros((cols, 4), dtype=torch.float32, device=wt.device)
                        full[:, dst_col:dst_col + lin.out_features] = wt
                        per_seg.append(pack(full, 0, 4))
                    else:
                        per_seg.append(pack(wt.contiguous(), 0, lin.out_features))
                out.append(per_seg)
        return out

    def _forward_program(self, x_pts_encoded: torch.Tensor, x_views_encoded: Optional[torch.Tensor]) -> torch.Tensor:
        """nerf.py:173-209 for any architecture: the layer-program MFMA kernels in the operand type of
        ``precision`` where they cover the architecture, else layer-by-layer library GEMMs in fp32."""
        if x_pts_encoded.shape[-1] != self.input_ch_pts:
            raise ValueError(f"expected encoded points of width {self.input_ch_pts}, got {x_pts_encoded.shape[-1]}")
        if not self.use_viewdirs:
            x_views_encoded = None
        elif x_views_encoded.shape[-1] != self.input_ch_views:
            raise ValueError(f"expected encoded view directions of width {self.input_ch_views}, got {x_views_encoded.shape[-1]}")
        if not self.program_supported():
            return self._forward_library(x_pts_encoded, x_views_encoded, None)
        lead = x_pts_encoded.shape[:-1]
        xp = x_pts_encoded.reshape(-1, self.input_ch_pts)
        xv = None if x_views_encoded is None else x_views_encoded.reshape(-1, self.input_ch_views)
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        raw = _ProgramForward.apply(self, xp, xv, want_grad, self.program_mode(), *self.ordered_params())
        return raw.view(*lead, 4)

    def _forward_library(self, x_pts_encoded, x_views_encoded, params) -> torch.Tensor:
        """nerf.py:173-209 layer by layer with ``torch.nn.functional.linear`` (rocBLAS), on ``params`` (a list in
        state_dict order) or on the module's own parameters."""
        F = torch.nn.functional
        it = iter(self.ordered_params() if params is None else params)
        lin = lambda x: F.linear(x, next(it), next(it))
        h = x_pts_encoded
        for i in range(self.D):
            h = torch.relu(lin(h))
            if i in self.skips:
                h = torch.cat((x_pts_encoded, h), dim=-1)
        if not self.use_viewdirs:
            return lin(h)
        sigma = lin(h)
        feat = lin(h)
        g = torch.relu(lin(torch.cat((feat, x_views_encoded), dim=-1)))
        rgb = lin(g)
        return torch.cat((rgb, sigma), dim=-1)


class _ProgramForward(torch.autograd.Function):
    """Forward and parameter gradients on the layer-program MFMA kernels of this library, in fp32 or with 16-bit
    operands (nerf_mlp_fp32_* / nerf_mlp_p16_*: the training forward keeps the activations, the backward runs the
    backward-data chain and the weight-gradient contractions).  Only when a gradient with respect to the encoded
    inputs is asked for -- which the reference's scripts never do -- are the layers evaluated again with library
    GEMMs (fp32) under autograd and differentiated there."""

    N_FIXED = 5          # arguments in front of the parameters

    @staticmethod
    def forward(ctx, model, x_pts, x_views, want_grad, mode, *params):
        layers, _keep, spec = model._layer_program(mode, want_grad)
        ctx.model, ctx.mode = model, mode
        ctx.has_views = x_views is not None
        need_inputs = ctx.needs_input_grad[1] or (x_views is not None and ctx.needs_input_grad[2])
        # (needs_input_grad is set for parameters under no_grad as well: the caller says whether a graph is recorded)
        ctx.kernel_backward = want_grad and any(ctx.needs_input_grad[_ProgramForward.N_FIXED:]) and not need_inputs
        if ctx.kernel_backward:
            if mode == "fp32":
                raw, ctx.acts = ops.fp32_mlp_forward_train(x_pts, x_views, layers)
            else:
                raw, ctx.acts = ops.p16_mlp_forward_train(x_pts, x_views, layers, mode)
            ctx.layers, ctx.keep, ctx.spec = layers, _keep, spec
            ctx.wt = NeRF._transposed_weights(spec, mode)       # of the weights this forward used
            ctx.param_ids = [id(p) for p in model.ordered_params()]
            ctx.M, ctx.cp = x_pts.shape[0], x_pts.shape[1]
            ctx.cv = x_views.shape[1] if x_views is not None else 0
        else:
            raw = ops.fp32_mlp_forward(x_pts, x_views, layers) if mode == "fp32" else \
                ops.p16_mlp_forward(x_pts, x_views, layers, mode)
            ctx.save_for_backward(x_pts, *([x_views] if x_views is not None else []), *params)
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        nf = _ProgramForward.N_FIXED
        if ctx.kernel_backward:
            n_par = len(ctx.param_ids)
            grads_out = [None] * n_par
            arr = (_C.Fp32Grad * len(ctx.spec))()
            hold = []
            for G, (lin, segments, _dst, _dst_col), wts in zip(arr, ctx.spec, ctx.wt):
                dw, db = torch.empty_like(lin.weight), torch.empty_like(lin.bias)
                hold += [dw, db]
                G.d_weight, G.d_bias, G.ld = dw.data_ptr(), db.data_ptr(), lin.in_features
                for s, ((col0, cols, _src), wt) in enumerate(zip(segments, wts)):
                    G.col0[s], G.cols[s] = col0, cols
                    G.wt[s] = None if wt is None else wt.data_ptr()
                grads_out[ctx.param_ids.index(id(lin.weight))] = dw
                grads_out[ctx.param_ids.index(id(lin.bias))] = db
            if ctx.mode == "fp32":
                ops.fp32_mlp_backward(ctx.M, ctx.cp, ctx.cv, ctx.layers, arr, d_raw.reshape(ctx.M, 4), ctx.acts)
            else:
                ops.p16_mlp_backward(ctx.M, ctx.cp, ctx.cv, ctx.layers, arr, d_raw.reshape(ctx.M, 4), ctx.acts, ctx.mode)
            ctx.acts = None
            need = ctx.needs_input_grad[nf:]
            return (*([None] * nf), *[g if n else None for g, n in zip(grads_out, need)])
        saved = ctx.saved_tensors
        x_pts = saved[0]
        x_views = saved[1] if ctx.has_views else None
        params = saved[2 if ctx.has_views else 1:]
        need = list(ctx.needs_input_grad[1:3]) + list(ctx.needs_input_grad[nf:])          # x_pts, x_views, params...
        with torch.enable_grad():
            xp = x_pts.detach().requires_grad_(need[0])
            xv = None if x_views is None else x_views.detach().requires_grad_(need[1])
            ps = [p.detach().requires_grad_(n) for p, n in zip(params, need[2:])]
            raw = ctx.model._forward_library(xp, xv, ps)
            wanted = [t for t, n in zip([xp, xv] + ps, need) if n and t is not None]
            got = iter(torch.autograd.grad(raw, wanted, d_raw.contiguous(), allow_unused=True)) if wanted else iter(())
        out = [next(got) if (n and t is not None) else None for t, n in zip([xp, xv] + ps, need)]
        return (None, out[0], out[1], None, None, *out[2:])


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
    rays_o, rays_d = ops._gpu_f32(rays_o, "rays_o"), ops._gpu_f32(rays_d, "rays_d")     # no CPU path: say so here
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

    # a network built without view directions ignores them whatever use_viewdirs says (nerf.py:202-206)
    fused = (isinstance(nerf_model, NeRF) and nerf_model.fused()
             and isinstance(embed_fn_pts, PositionalEncoder) and embed_fn_pts.is_standard(10)
             and (not nerf_model.use_viewdirs
                  or (use_viewdirs and isinstance(embed_fn_views, PositionalEncoder) and embed_fn_views.is_standard(4))))
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
