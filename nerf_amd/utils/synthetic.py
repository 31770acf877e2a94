"""Synthetic Blender-style scene set-up used by the benchmark and the smoke test (there is no dataset
or checkpoint in the build environment): camera intrinsics and poses with the reference's formulas,
seeded random-init weights, seeded ray batches."""
from __future__ import annotations

import math

import numpy as np
import torch

from .ray_utils import get_rays

CAMERA_ANGLE_X = 0.6911112070083618   # the Blender "lego" transforms file value


def focal_from_fov(width: int, camera_angle_x: float = CAMERA_ANGLE_X) -> float:
    """src/nerf_project/datasets/blender.py:113."""
    return 0.5 * width / math.tan(0.5 * camera_angle_x)


def pose_spherical(theta: float, phi: float, radius: float, target_height: float = 0.0) -> torch.Tensor:
    """scripts/render_nerf.py:72-124: shift_y(h) @ R_y(theta) @ R_x(phi) @ T_z(radius); angles in degrees.
    (Unlike the original NeRF code there is no final axis-swap matrix.)"""
    t = np.eye(4, dtype=np.float32); t[2, 3] = radius
    p = phi / 180.0 * np.pi
    rp = np.eye(4, dtype=np.float32)
    rp[1, 1] = np.cos(p); rp[1, 2] = -np.sin(p); rp[2, 1] = np.sin(p); rp[2, 2] = np.cos(p)
    th = theta / 180.0 * np.pi
    rt = np.eye(4, dtype=np.float32)
    rt[0, 0] = np.cos(th); rt[0, 2] = np.sin(th); rt[2, 0] = -np.sin(th); rt[2, 2] = np.cos(th)
    sh = np.eye(4, dtype=np.float32); sh[1, 3] = target_height
    return torch.from_numpy(sh @ (rt @ rp @ t))


def camera_path(num_frames: int, phi: float = -30.0, radius: float = 4.0, target_height: float = 0.0) -> torch.Tensor:
    """scripts/render_nerf.py:145-149: theta_i = i / num_frames * 360 degrees at fixed elevation."""
    return torch.stack([pose_spherical(i / num_frames * 360.0, phi, radius, target_height) for i in range(num_frames)])


@torch.no_grad()
def seed_parameters_(model: torch.nn.Module, seed: int) -> torch.nn.Module:
    """Overwrite the parameters with a machine-independent seeded draw: numpy default_rng(seed), in
    state_dict order, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases (nn.Linear's bounds)."""
    rng = np.random.default_rng(seed)
    fan_in = 1
    for name, p in model.named_parameters():
        if name.endswith(".weight"):
            fan_in = p.shape[1]
        bound = 1.0 / math.sqrt(fan_in)
        a = rng.uniform(-bound, bound, size=tuple(p.shape)).astype(np.float32)
        p.copy_(torch.from_numpy(a).to(p.device))
    return model


def synthetic_rays(n: int, device, seed: int = 0, H: int = 800, W: int = 800, theta: float = 0.0,
                   phi: float = -30.0, radius: float = 4.0):
    """First n pixels of a seeded permutation of an H x W frame seen from pose_spherical(theta, phi, radius)."""
    c2w = pose_spherical(theta, phi, radius).to(device)
    ro, rd = get_rays(H, W, focal_from_fov(W), c2w)
    idx = torch.from_numpy(np.random.default_rng(seed).permutation(H * W)[:n].astype(np.int64)).to(device)
    return ro.reshape(-1, 3)[idx].contiguous(), rd.reshape(-1, 3)[idx].contiguous()
