"""Frame rendering: the chunk loop of scripts/render_nerf.py:260-314, sharded over ranks by image rows."""
from __future__ import annotations

import math
from typing import Optional

import torch

from .models.nerf import render_hierarchical, render_rays
from .parallel import gather_rows, shard_rows
from .utils.ray_utils import get_rays


@torch.no_grad()
def render_image(model_coarse, model_fine, embed_fn_pts, embed_fn_views, H: int, W: int, focal: float,
                 c2w: torch.Tensor, near: float = 2.0, far: float = 6.0, N_samples_coarse: int = 64,
                 N_samples_fine: int = 128, chunk: int = 65536, lindisp: bool = False, white_bkgd: bool = False,
                 rank: int = 0, world_size: int = 1, gather: bool = True) -> torch.Tensor:
    """Deterministic two-pass render of one camera (rand=False, det=True, as render_nerf.py:275-303).

    Each rank generates and renders only its block of image rows; with ``gather`` the blocks are
    all-gathered into the full [H, W, 3] frame on every rank, otherwise this rank's
    [nrows, W, 3] block is returned.  ``chunk`` is the number of rays per launch group (the
    reference's eval_batch_size, render_nerf.py:55; the kernels prefer larger chunks than 4096).
    ``model_fine=None`` renders the coarse pass only (render_nerf.py:282-285)."""
    # a checkpoint trained without view directions has no view encoder (render_nerf.py:278, :300 pass
    # use_viewdirs=use_viewdirs_model)
    use_viewdirs = embed_fn_views is not None and getattr(model_coarse, "use_viewdirs", True)
    row0, nrows = shard_rows(H, rank, world_size)
    rays_o, rays_d = get_rays(H, W, focal, c2w, row0=row0, nrows=nrows)
    rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    out = torch.empty((nrows * W, 3), dtype=torch.float32, device=rays_o.device)
    for s in range(0, nrows * W, chunk):
        ro, rd = rays_o[s:s + chunk], rays_d[s:s + chunk]
        if model_fine is None:
            res = render_rays(model_coarse, embed_fn_pts, embed_fn_views, ro, rd, near, far, N_samples_coarse,
                              rand=False, lindisp=lindisp, use_viewdirs=use_viewdirs, white_bkgd=white_bkgd)
        else:
            _, res = render_hierarchical(model_coarse, model_fine, embed_fn_pts, embed_fn_views, ro, rd, near, far,
                                         N_samples_coarse, N_samples_fine, rand=False, det=True, lindisp=lindisp,
                                         use_viewdirs=use_viewdirs, white_bkgd=white_bkgd)
        out[s:s + chunk] = res["rgb_map"]
    block = out.reshape(nrows, W, 3)
    return gather_rows(block, H, rank, world_size) if gather else block


def to_uint8(img: torch.Tensor) -> torch.Tensor:
    """clip to [0,1] then the truncating float -> uint8 cast of the reference
    (render_nerf.py:309 + utils/image_utils.py:52-92), on the device, so that only bytes cross PCIe."""
    return (img.clamp(0.0, 1.0) * 255.0).to(torch.uint8)


@torch.no_grad()
def render_path(model_coarse, model_fine, embed_fn_pts, embed_fn_views, H: int, W: int, focal: float,
                poses: torch.Tensor, *, rank: int = 0, world_size: int = 1, output_dir: Optional[str] = None,
                **render_kw) -> dict:
    """The frame loop of scripts/render_nerf.py:254-314 over a camera path (``poses`` [n, 4, 4], e.g.
    ``utils.synthetic.camera_path(120)``).  Frames are independent, so they are dealt round-robin to the
    ranks (no collective; every rank renders whole frames with all its CUs); each frame is clipped and
    converted to uint8 on the device and only those bytes are copied to the host.  With ``output_dir``
    every rank writes its frames as ``frame_%04d.png``.  Returns {frame index: uint8 ndarray [H, W, 3]}
    for this rank's frames.  ``render_kw`` goes to ``render_image`` (near, far, N_samples_*, chunk,
    lindisp, white_bkgd)."""
    from .parallel import shard_items
    from .utils.png import write_png
    import os
    frames = {}
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
    for i in shard_items(poses.shape[0], rank, world_size):
        img = render_image(model_coarse, model_fine, embed_fn_pts, embed_fn_views, H, W, focal, poses[i],
                           rank=0, world_size=1, **render_kw)
        frame = to_uint8(img).cpu().numpy()
        frames[i] = frame
        if output_dir is not None:
            write_png(os.path.join(output_dir, f"frame_{i:04d}.png"), frame)
    return frames


@torch.no_grad()
def evaluate(model_coarse, model_fine, embed_fn_pts, embed_fn_views, dataset, *, rank: int = 0, world_size: int = 1,
             **render_kw) -> dict:
    """Image-by-image PSNR over a dataset split: the evaluation loop of scripts/evaluate_nerf.py:201-270
    without its third-party metrics (SSIM, LPIPS).  ``dataset`` is a ``BlenderDataset`` (val / test
    split); images are dealt round-robin to the ranks.  Returns {"psnr": {image index: dB}, "mse": {...}}
    for this rank's images; MSE and PSNR are computed on the device from the unclipped render, as the
    reference does (utils/image_utils.py:8-49)."""
    from .parallel import shard_items
    out = {"psnr": {}, "mse": {}}
    for i in shard_items(len(dataset), rank, world_size):
        img = render_image(model_coarse, model_fine, embed_fn_pts, embed_fn_views, dataset.height, dataset.width,
                           dataset.focal, dataset.poses[i], rank=0, world_size=1, **render_kw)
        mse = torch.mean((img - dataset.image_rgb(i)) ** 2).item()
        out["mse"][i] = mse
        out["psnr"][i] = float("inf") if mse == 0.0 else -10.0 * math.log10(mse)
    return out
