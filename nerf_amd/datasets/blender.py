"""Blender-scene loader with the image bank and the ray generator on the GPU.

Counterpart of the reference's ``BlenderDataset`` (src/nerf_project/datasets/blender.py:34-164): same
constructor arguments, same ``focal`` / ``height`` / ``width`` / ``poses`` / ``image_paths``
attributes, same ``__len__`` and the same per-image dict from ``__getitem__``.  What differs is where the
data lives.  The reference keeps every image as fp32 RGB on the host and calls ``get_rays`` for a whole
image inside DataLoader workers; here the images stay **uint8 in HBM** exactly as decoded (a 100-image
800x800 RGBA scene is 256 MB instead of 768 MB of fp32 RGB plus 1.5 GB of rays per epoch), and
``sample_rays`` / ``gather`` turn any list of pixel ids straight into (rays_o, rays_d, target_rgb) with
one kernel -- which is what a training step over 1024 rays drawn across all images needs
(scripts/train_nerf.py:280-283 as intended).
"""
from __future__ import annotations

import json
import os
import warnings
from typing import Dict, Optional

import numpy as np
import torch

from .. import ops
from ..utils.png import read_png


def _unit_lut() -> np.ndarray:
    """v/255 for v = 0..255 the way blender.py:92 rounds it (float64 divide, then one cast to fp32)."""
    return (np.arange(256, dtype=np.float64) / 255.0).astype(np.float32)


class BlenderDataset:
    def __init__(self, dataset_path: str, split: str, transform=None, data_white_bkgd: bool = False,
                 device: str = "cuda"):
        self.dataset_path, self.split, self.transform = dataset_path, split, transform
        self.data_white_bkgd = bool(data_white_bkgd)
        self.device = torch.device(device)
        self.image_paths, poses, images = [], [], []
        self.focal: Optional[float] = None
        self.height: Optional[int] = None
        self.width: Optional[int] = None

        json_path = os.path.join(dataset_path, f"transforms_{split}.json")
        with open(json_path, "r") as f:              # FileNotFoundError propagates, as in blender.py:58-63
            meta = json.load(f)
        camera_angle_x = float(meta["camera_angle_x"])
        for idx, frame in enumerate(meta["frames"]):
            rel = frame["file_path"]
            if rel.startswith("./"):
                rel = rel[2:]
            path = os.path.join(dataset_path, rel + ".png")
            try:
                img = read_png(path)
            except FileNotFoundError:                # blender.py:79-86: warn and skip the frame
                warnings.warn(f"skipping frame {idx}: image not found at {path}")
                continue
            if img.shape[2] not in (3, 4):
                raise ValueError(f"{path}: expected an RGB or RGBA image, got {img.shape[2]} channel(s)")
            if self.height is None:
                self.height, self.width = int(img.shape[0]), int(img.shape[1])
            if img.shape[:2] != (self.height, self.width) or (images and img.shape[2] != images[0].shape[2]):
                raise ValueError(f"{path}: image shape {img.shape} differs from the first image's")
            self.image_paths.append(path)
            poses.append(np.asarray(frame["transform_matrix"], dtype=np.float64))
            images.append(img)

        if images:
            self.focal = float(0.5 * self.width / np.tan(0.5 * camera_angle_x))       # blender.py:113
            self.poses = torch.from_numpy(np.stack(poses)).float().to(self.device)  # [n,4,4]
            self.images = torch.from_numpy(np.stack(images)).to(self.device)        # uint8 [n,H,W,C]
        else:
            warnings.warn("no image could be loaded; focal length is undefined")
            self.poses = torch.zeros((0, 4, 4), device=self.device)
            self.images = torch.zeros((0, 0, 0, 3), dtype=torch.uint8, device=self.device)
        self._lut = torch.from_numpy(_unit_lut()).to(self.device)
        self._all_rgbs: Optional[torch.Tensor] = None

    def __len__(self) -> int:
        return len(self.image_paths)

    @property
    def pixels_per_image(self) -> int:
        return self.height * self.width

    @property
    def num_rays(self) -> int:
        return len(self) * self.pixels_per_image

    # -- ray batches ------------------------------------------------------------------------------
    def gather(self, pix: torch.Tensor):
        """(rays_o, rays_d, target_rgb), each [n, 3], of global pixel ids image*H*W + y*W + x."""
        if pix.numel() and (int(pix.min()) < 0 or int(pix.max()) >= self.num_rays):
            raise IndexError("pixel id out of range")
        return ops.gather_rays(pix, self.height, self.width, self.focal, self.poses, self.images, self._lut,
                               self.data_white_bkgd)

    def sample_rays(self, n: int, generator: Optional[torch.Generator] = None, rank: int = 0, world_size: int = 1,
                    out=None):
        """n rays drawn uniformly (with replacement) over every pixel of every image.  With
        ``world_size`` > 1 all ranks must pass generators in the same state: one draw of
        n*world_size ids is made and rank r keeps its slice, so ranks see disjoint parts of one batch.
        ``out`` = (rays_o, rays_d, target) receives the batch in place (a training step's input buffers)."""
        total = n * world_size
        pix = torch.randint(0, self.num_rays, (total,), device=self.device, generator=generator)
        return ops.gather_rays(pix[rank * n:(rank + 1) * n], self.height, self.width, self.focal, self.poses,
                               self.images, self._lut, self.data_white_bkgd, out=out)

    # -- the reference's per-image view -------------------------------------------------------------
    def image_rgb(self, idx: int) -> torch.Tensor:
        """fp32 [H, W, 3] target of image idx (blender.py:92-104)."""
        hw = self.pixels_per_image
        pix = torch.arange(idx * hw, (idx + 1) * hw, device=self.device)
        return self.gather(pix)[2].view(self.height, self.width, 3)

    @property
    def all_rgbs(self) -> torch.Tensor:
        """fp32 [n_images, H, W, 3]; materialised on first use (blender.py:109)."""
        if self._all_rgbs is None:
            self._all_rgbs = torch.stack([self.image_rgb(i) for i in range(len(self))]) if len(self) else \
                torch.zeros((0, 0, 0, 3), device=self.device)
        return self._all_rgbs

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        if not -len(self) <= idx < len(self):
            raise IndexError(idx)
        idx %= len(self)
        pose = self.poses[idx]
        rays_o, rays_d = ops.get_rays(self.height, self.width, self.focal, pose)
        rgb = self.image_rgb(idx)
        if self.split == "train":                    # blender.py:153-161
            return {"rays_o": rays_o.reshape(-1, 3), "rays_d": rays_d.reshape(-1, 3), "target_rgb": rgb.reshape(-1, 3)}
        return {"rays_o": rays_o, "rays_d": rays_d, "target_rgb": rgb, "pose": pose}
