"""Ray generation (drop-in for the reference's src/nerf_project/utils/ray_utils.py)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import ops


def get_rays(H, W, focal, c2w: torch.Tensor, *, row0: int = 0, nrows: Optional[int] = None
             ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Ray origins and unit directions for every pixel of an H x W pinhole camera
    (ray_utils.py:6-65): returns contiguous fp32 ``rays_o, rays_d`` of shape [H, W, 3] on
    ``c2w.device``.  ``c2w`` is [4,4] or [3,4].  Conventions follow the reference: no
    half-pixel offset, y is not negated, directions are normalised.

    ``row0``/``nrows`` (keyword-only) restrict the output to a block of image rows, which is how
    a frame is sharded across GPUs.
    """
    return ops.get_rays(H, W, focal, c2w, row0=row0, nrows=nrows)
