"""nerf_amd: the NeRF render path (positional encoding, 8x256 MLP, hierarchical sampling, alpha
compositing, ray generation) as hand-written HIP kernels for AMD MI355X (gfx950), behind the
Python call surface of hadleyhzy34/nerf.

    from nerf_amd.models.nerf import PositionalEncoder, NeRF, render_rays, sample_pdf
    from nerf_amd.utils.ray_utils import get_rays

Importing the package loads nerf_amd/lib/libnerf_hip.so and fails if it has not been built.
"""
from . import _C  # noqa: F401  (loads the shared library; raises if missing)
from .models.nerf import NeRF, PositionalEncoder, render_hierarchical, render_rays, sample_pdf
from .utils.ray_utils import get_rays

__all__ = ["NeRF", "PositionalEncoder", "render_rays", "render_hierarchical", "sample_pdf", "get_rays"]
__version__ = "0.1.0"
