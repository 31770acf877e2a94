"""Ray generation (drop-in for the reference's src/nerf_project/utils/ray_utils.py)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import ops


def _get_rays_host(H: int, W: int, focal: float, c2w: torch.Tensor, row0: int, nrows: int
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """The same rays for a pose held in host memory, in torch CPU ops (ray_utils.py:38-65).  The reference's
    loader calls get_rays on CPU tensors inside DataLoader worker processes (datasets/blender.py:151), which
    must not touch HIP.  Operation order follows the reference, so the results are bit-identical to it:
    camera-space direction ((x - W/2)/f, (y - H/2)/f, -1), the three products with a row of the rotation
    summed in index order, division by the Euclidean norm."""
    rot = c2w[:3, :3].to(torch.float32)
    ys = torch.arange(row0, row0 + nrows, dtype=torch.float32).view(nrows, 1).expand(nrows, W)
    xs = torch.arange(W, dtype=torch.float32).view(1, W).expand(nrows, W)
    cam = torch.stack(((xs - W * 0.5) / focal, (ys - H * 0.5) / focal, torch.full((nrows, W), -1.0)), dim=-1)
    world = (cam.unsqueeze(-2) * rot).sum(dim=-1)
    rays_d = world / torch.linalg.vector_norm(world, dim=-1, keepdim=True)
    rays_o = c2w[:3, 3].to(torch.float32).expand(nrows, W, 3)
    return rays_o.contiguous(), rays_d.contiguous()


def get_rays(H, W, focal, c2w: torch.Tensor, *, row0: int = 0, nrows: Optional[int] = None
             ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Ray origins and unit directions for every pixel of an H x W pinhole camera
    (ray_utils.py:6-65): returns contiguous fp32 ``rays_o, rays_d`` of shape [H, W, 3] on
    ``c2w.device``.  ``c2w`` is [4,4] or [3,4].  Conventions follow the reference: no
    half-pixel offset, y is not negated, directions are normalised.

    A pose on the GPU runs the HIP kernel; a pose in host memory (the reference's loader workers,
    datasets/blender.py:151) is served by torch CPU ops and never initialises the GPU.

    ``row0``/``nrows`` (keyword-only) restrict the output to a block of image rows, which is how
    a frame is sharded across GPUs.
    """
    if isinstance(c2w, torch.Tensor) and not c2w.is_cuda:
        if c2w.dim() != 2 or c2w.shape[1] != 4 or c2w.shape[0] not in (3, 4):
            raise ValueError(f"c2w must be [4,4] or [3,4], got {tuple(c2w.shape)}")
        H, W = int(H), int(W)
        return _get_rays_host(H, W, float(focal), c2w, int(row0), H - row0 if nrows is None else int(nrows))
    return ops.get_rays(H, W, focal, c2w, row0=row0, nrows=nrows)
