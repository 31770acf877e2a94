"""CPU oracle for the NeRF render path.  TEST INFRASTRUCTURE ONLY.

This module is a build-owned CPU restatement (plain torch CPU ops, fp32) of the
algorithm of the reference's hot path.  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; the
product package ``nerf_amd`` never imports it and has no CPU fallback.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference itself
(``/root/reference/src/nerf_project``) in the development container, checks
every function below against it, and writes the vectors under
``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks this file against
those vectors wherever the repo is run (the reference never travels).

Each function cites the reference lines it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------

def param_shapes(D: int = 8, W: int = 256, input_ch_pts: int = 63,
                 input_ch_views: int = 27, skips: Sequence[int] = (4,),
                 use_viewdirs: bool = True) -> List[Tuple[str, Tuple[int, ...]]]:
    """state_dict keys and shapes of the MLP, in registration order.

    Follows src/nerf_project/models/nerf.py:126-153 (pts_linears first layer,
    D-1 further layers with a widened input after each index in ``skips``,
    then sigma_linear / feature_linear / rgb_layers.{0,2} or output_linear).
    """
    out: List[Tuple[str, Tuple[int, ...]]] = []
    out.append(("pts_linears.0.weight", (W, input_ch_pts)))
    out.append(("pts_linears.0.bias", (W,)))
    for i in range(D - 1):
        fan_in = W + input_ch_pts if i in skips else W
        out.append((f"pts_linears.{i + 1}.weight", (W, fan_in)))
        out.append((f"pts_linears.{i + 1}.bias", (W,)))
    if use_viewdirs:
        out.append(("sigma_linear.weight", (1, W)))
        out.append(("sigma_linear.bias", (1,)))
        out.append(("feature_linear.weight", (W, W)))
        out.append(("feature_linear.bias", (W,)))
        out.append(("rgb_layers.0.weight", (W // 2, W + input_ch_views)))
        out.append(("rgb_layers.0.bias", (W // 2,)))
        out.append(("rgb_layers.2.weight", (3, W // 2)))
        out.append(("rgb_layers.2.bias", (3,)))
    else:
        out.append(("output_linear.weight", (4, W)))
        out.append(("output_linear.bias", (4,)))
    return out


def init_params(seed: int, **arch) -> Dict[str, torch.Tensor]:
    """Seeded random-init weights, reproducible on any machine.

    A numpy ``default_rng(seed)`` stream drawn in state_dict order with the
    bounds of ``nn.Linear``'s default init, U(-1/sqrt(fan_in), +1/sqrt(fan_in))
    for weight and bias alike (the reference uses the default ``nn.Linear``
    init, nerf.py:128-153; the stream itself is build-owned so that the
    weights can be regenerated instead of stored).
    """
    rng = np.random.default_rng(seed)
    params: Dict[str, torch.Tensor] = {}
    fan_in = 1
    for name, shape in param_shapes(**arch):
        if name.endswith(".weight"):
            fan_in = shape[1]
        bound = 1.0 / math.sqrt(fan_in)
        a = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        params[name] = torch.from_numpy(a)
    return params


def params_sha256(params: Dict[str, torch.Tensor]) -> str:
    h = hashlib.sha256()
    for k in params:
        h.update(k.encode())
        h.update(params[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


# --------------------------------------------------------------------------
# A1 positional encoding
# --------------------------------------------------------------------------

def freq_bands(num_freqs: int, log_sampling: bool = True) -> torch.Tensor:
    """nerf.py:41-46: 2**linspace(0, L-1, L) or linspace(1, 2**(L-1), L). No pi."""
    if log_sampling:
        return 2.0 ** torch.linspace(0.0, num_freqs - 1, num_freqs)
    return torch.linspace(1.0, 2.0 ** (num_freqs - 1), num_freqs)


def positional_encode(x: torch.Tensor, num_freqs: int, log_sampling: bool = True) -> torch.Tensor:
    """nerf.py:68-76: [x, sin(f0 x), cos(f0 x), sin(f1 x), cos(f1 x), ...] on the last dim."""
    outs = [x]
    for f in freq_bands(num_freqs, log_sampling):
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, dim=-1)


# --------------------------------------------------------------------------
# A3 MLP
# --------------------------------------------------------------------------

def mlp_forward(params: Dict[str, torch.Tensor], x_pts: torch.Tensor,
                x_views: Optional[torch.Tensor] = None, D: int = 8,
                skips: Sequence[int] = (4,), use_viewdirs: bool = True) -> torch.Tensor:
    """nerf.py:173-209: D x (Linear, ReLU) with [x, h] concat after each skip
    index; sigma and (activation-free) feature heads from the last hidden
    state; rgb = Linear(ReLU(Linear([feature, views]))); returns [rgb, sigma]."""
    h = x_pts
    for i in range(D):
        h = F.relu(F.linear(h, params[f"pts_linears.{i}.weight"], params[f"pts_linears.{i}.bias"]))
        if i in skips:
            h = torch.cat([x_pts, h], dim=-1)
    if use_viewdirs:
        if x_views is None:
            raise ValueError("View directions (x_views_encoded) must be provided when use_viewdirs is True.")
        sigma = F.linear(h, params["sigma_linear.weight"], params["sigma_linear.bias"])
        feat = F.linear(h, params["feature_linear.weight"], params["feature_linear.bias"])
        comb = torch.cat([feat, x_views], dim=-1)
        g = F.relu(F.linear(comb, params["rgb_layers.0.weight"], params["rgb_layers.0.bias"]))
        rgb = F.linear(g, params["rgb_layers.2.weight"], params["rgb_layers.2.bias"])
    else:
        o = F.linear(h, params["output_linear.weight"], params["output_linear.bias"])
        rgb, sigma = o[..., :3], o[..., 3:4]
    return torch.cat([rgb, sigma], dim=-1)


# --------------------------------------------------------------------------
# A4 sample placement
# --------------------------------------------------------------------------

def place_samples(n_rays: int, near, far, n_samples: int, lindisp: bool = False,
                  z_vals_override: Optional[torch.Tensor] = None,
                  t_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nerf.py:373-402.  ``t_rand`` (uniforms in [0,1), shape of z_vals) stands in
    for the reference's ``torch.rand`` draw at nerf.py:400; ``None`` means rand=False."""
    if z_vals_override is not None:
        z = z_vals_override
    else:
        t = torch.linspace(0.0, 1.0, steps=n_samples)
        _near = torch.as_tensor(near).view(-1, 1)
        _far = torch.as_tensor(far).view(-1, 1)
        if _near.shape[0] == 1 and n_rays > 1:
            _near = _near.expand(n_rays, 1)
        if _far.shape[0] == 1 and n_rays > 1:
            _far = _far.expand(n_rays, 1)
        if lindisp:
            z = 1.0 / (1.0 / _near * (1.0 - t) + 1.0 / _far * t)
        else:
            z = _near * (1.0 - t) + _far * t
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], dim=-1)
        lower = torch.cat([z[..., :1], mids], dim=-1)
        z = lower + (upper - lower) * t_rand
    return z


# --------------------------------------------------------------------------
# A5 compositing
# --------------------------------------------------------------------------

def composite(raw: torch.Tensor, z_vals: torch.Tensor, white_bkgd: bool = False) -> Dict[str, torch.Tensor]:
    """nerf.py:430-483 on raw [N,S,4], z_vals [N,S]."""
    n_rays = raw.shape[0]
    rgb = torch.sigmoid(raw[..., :3])
    sigma_a = F.relu(raw[..., 3])
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], dim=-1)
    alpha = 1.0 - torch.exp(-sigma_a * dists)
    trans = torch.cumprod(torch.cat([torch.ones((n_rays, 1)), 1.0 - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, dim=-2)
    depth_map = torch.sum(weights * z_vals, dim=-1)
    acc_map = torch.sum(weights, dim=-1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / (acc_map + 1e-10))
    disp_map = torch.nan_to_num(disp_map, nan=0.0)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return {"rgb_map": rgb_map, "depth_map": depth_map, "disp_map": disp_map, "acc_map": acc_map,
            "weights": weights, "z_vals": z_vals, "alpha": alpha, "raw_sigma": raw[..., 3]}


# --------------------------------------------------------------------------
# A6 hierarchical sampling
# --------------------------------------------------------------------------

def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, n_samples: int, det: bool = False,
               u: Optional[torch.Tensor] = None, return_inds: bool = False):
    """nerf.py:237-315.  ``u`` replaces the ``torch.rand`` draw at nerf.py:255.
    Requires bins.shape[-1] == weights.shape[-1] + 1 (the real callers pass 63
    and 62; the gather at nerf.py:295 indexes ``bins`` with CDF indices)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    if u is None:
        if not det:
            raise ValueError("the oracle needs explicit uniforms for stochastic sampling")
        u = torch.linspace(0.0, 1.0, steps=n_samples)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], dim=-1)
    b = cdf.shape[0]
    cdf_g = torch.gather(cdf, 1, inds_g.view(b, -1)).view(b, n_samples, 2)
    bins_g = torch.gather(bins, 1, inds_g.view(b, -1)).view(b, n_samples, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    samples = bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])
    if return_inds:
        return samples, inds, cdf
    return samples


def merge_samples(z_coarse: torch.Tensor, weights_coarse: torch.Tensor, n_fine: int, det: bool,
                  u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Coarse->fine glue, scripts/train_nerf.py:304-320 (= render_nerf.py:285-294):
    midpoints, weights[1:-1] + 1e-5, sample_pdf, sort(cat)."""
    z_mid = 0.5 * (z_coarse[..., 1:] + z_coarse[..., :-1])
    w_pdf = weights_coarse[..., 1:-1] + 1e-5
    z_fine = sample_pdf(z_mid, w_pdf, n_fine, det=det, u=u)
    z_comb, _ = torch.sort(torch.cat([z_coarse, z_fine], dim=-1), dim=-1)
    return z_comb


# --------------------------------------------------------------------------
# A8 rays / poses
# --------------------------------------------------------------------------

def get_rays(H: int, W: int, focal: float, c2w: torch.Tensor):
    """src/nerf_project/utils/ray_utils.py:38-65."""
    py, px = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing="ij")
    dx = (px - W * 0.5) / focal
    dy = (py - H * 0.5) / focal
    dz = -torch.ones_like(dx)
    dirs = torch.stack([dx, dy, dz], dim=-1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], dim=-1)
    rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o.contiguous(), rays_d.contiguous()


def pose_spherical(theta: float, phi: float, radius: float, target_height: float = 0.0) -> torch.Tensor:
    """scripts/render_nerf.py:92-124: c2w = shift_y(h) @ R_y(theta) @ R_x(phi) @ T_z(radius), degrees."""
    t = torch.eye(4, dtype=torch.float32)
    t[2, 3] = radius
    rp = torch.eye(4, dtype=torch.float32)
    p = phi / 180.0 * np.pi
    rp[1, 1] = np.cos(p); rp[1, 2] = -np.sin(p); rp[2, 1] = np.sin(p); rp[2, 2] = np.cos(p)
    rt = torch.eye(4, dtype=torch.float32)
    th = theta / 180.0 * np.pi
    rt[0, 0] = np.cos(th); rt[0, 2] = np.sin(th); rt[2, 0] = -np.sin(th); rt[2, 2] = np.cos(th)
    c2w = rt @ rp @ t
    sh = torch.eye(4, dtype=torch.float32)
    sh[1, 3] = target_height
    return sh @ c2w


def focal_from_fov(width: int, camera_angle_x: float) -> float:
    """src/nerf_project/datasets/blender.py:113."""
    return 0.5 * width / np.tan(0.5 * camera_angle_x)


# --------------------------------------------------------------------------
# render_rays and the two-pass pipeline
# --------------------------------------------------------------------------

def render_rays(params: Dict[str, torch.Tensor], rays_o: torch.Tensor, rays_d: torch.Tensor,
                near, far, n_samples: int, lindisp: bool = False, white_bkgd: bool = False,
                z_vals_override: Optional[torch.Tensor] = None, t_rand: Optional[torch.Tensor] = None,
                L_pts: int = 10, L_views: int = 4, use_viewdirs: bool = True,
                D: int = 8, skips: Sequence[int] = (4,)) -> Dict[str, torch.Tensor]:
    """nerf.py:369-483 with PositionalEncoder(3, L_pts) / (3, L_views) as embedders."""
    n_rays = rays_o.shape[0]
    z = place_samples(n_rays, near, far, n_samples, lindisp, z_vals_override, t_rand)
    s = z.shape[-1]
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    e_pts = positional_encode(pts.reshape(-1, 3), L_pts)
    if use_viewdirs:
        vd = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
        vd = vd[:, None, :].expand_as(pts)
        e_views = positional_encode(vd.reshape(-1, 3), L_views)
        raw = mlp_forward(params, e_pts, e_views, D=D, skips=skips, use_viewdirs=True)
    else:
        raw = mlp_forward(params, e_pts, None, D=D, skips=skips, use_viewdirs=False)
    raw = raw.view(n_rays, s, raw.shape[-1])
    out = composite(raw, z, white_bkgd)
    out["raw"] = raw
    return out


def render_hierarchical(params_coarse, params_fine, rays_o, rays_d, near, far, n_coarse: int, n_fine: int,
                        det: bool = True, lindisp: bool = False, white_bkgd: bool = False,
                        t_rand_coarse=None, u_fine=None, t_rand_fine=None, **arch):
    """Two-pass render: scripts/render_nerf.py:275-303 (det, no jitter) or
    scripts/train_nerf.py:293-329 (jitter on both passes, stochastic PDF samples)."""
    coarse = render_rays(params_coarse, rays_o, rays_d, near, far, n_coarse, lindisp, white_bkgd,
                         t_rand=t_rand_coarse, **arch)
    with torch.no_grad():
        z_comb = merge_samples(coarse["z_vals"].detach(), coarse["weights"].detach(), n_fine, det, u_fine)
    fine = render_rays(params_fine, rays_o, rays_d, near, far, z_comb.shape[-1], lindisp, white_bkgd,
                       z_vals_override=z_comb, t_rand=t_rand_fine, **arch)
    return coarse, fine


def train_step_loss(params_coarse, params_fine, rays_o, rays_d, target_rgb, near, far, n_coarse, n_fine,
                    t_rand_coarse, u_fine, t_rand_fine, lindisp=False, white_bkgd=False):
    """Loss of scripts/train_nerf.py:299-331: mean((rgb_c - t)^2) + mean((rgb_f - t)^2)."""
    coarse, fine = render_hierarchical(params_coarse, params_fine, rays_o, rays_d, near, far, n_coarse, n_fine,
                                       det=False, lindisp=lindisp, white_bkgd=white_bkgd,
                                       t_rand_coarse=t_rand_coarse, u_fine=u_fine, t_rand_fine=t_rand_fine)
    loss_c = torch.mean((coarse["rgb_map"] - target_rgb) ** 2)
    loss_f = torch.mean((fine["rgb_map"] - target_rgb) ** 2)
    return loss_c + loss_f, loss_c, loss_f, coarse, fine


# --------------------------------------------------------------------------
# synthetic benchmark scene (shared by tests and the cpu_baseline timing)
# --------------------------------------------------------------------------

CAMERA_ANGLE_X = 0.6911112070083618


def synthetic_rays(n: int, seed: int = 0, H: int = 800, W: int = 800, theta: float = 0.0,
                   phi: float = -30.0, radius: float = 4.0):
    """First ``n`` pixels of a seeded permutation of an H x W frame seen from pose_spherical."""
    focal = focal_from_fov(W, CAMERA_ANGLE_X)
    c2w = pose_spherical(theta, phi, radius)
    ro, rd = get_rays(H, W, float(focal), c2w)
    perm = np.random.default_rng(seed).permutation(H * W)[:n]
    idx = torch.from_numpy(perm.astype(np.int64))
    return ro.reshape(-1, 3)[idx].contiguous(), rd.reshape(-1, 3)[idx].contiguous()
