"""Shared helpers for the tests (the oracle is imported here and only here / in tests)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import nerf_oracle as O  # noqa: E402


def T(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def amd_model(params, precision="bf16", device="cuda", **kw):
    """nerf_amd.NeRF loaded with oracle-generated parameters."""
    from nerf_amd.models.nerf import NeRF
    m = NeRF(precision=precision, **kw)
    m.load_state_dict(params, strict=True)
    return m.to(device)


def maxabs(a, b):
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()


def make_blender_scene(root, n_images=5, H=10, W=12, channels=4, seed=0, splits=("train", "val"), missing=()):
    """A tiny scene in the layout of the Blender NeRF data (transforms_<split>.json + PNG files), written
    with the product's PNG writer; poses from the oracle's pose_spherical.  Returns the decoded images."""
    from nerf_amd.utils.png import write_png
    import json
    rng = np.random.default_rng(seed)
    images = {}
    for split in splits:
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for i in range(n_images):
            img = rng.integers(0, 256, size=(H, W, channels), dtype=np.uint8)
            if channels == 4:
                img[: H // 3, :, 3] = 0          # fully transparent band
                img[H // 3: 2 * H // 3, :, 3] = 255
            rel = f"./{split}/r_{i}" if i % 2 == 0 else f"{split}/r_{i}"
            if i not in missing:
                write_png(os.path.join(root, split, f"r_{i}.png"), img)
                images[(split, i)] = img
            pose = O.pose_spherical(37.0 * i + 5.0, -30.0 + 3.0 * i, 4.0)
            frames.append({"file_path": rel, "rotation": 0.0123, "transform_matrix": pose.double().tolist()})
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, f)
    return images


def make_sphere_scene(root, n_train=12, H=64, W=64):
    """A scene with real 3-D structure in the Blender layout: a unit sphere at the origin whose colour is its
    surface position (0.5 + 0.5 p), transparent elsewhere, intersected analytically with the oracle's rays;
    n_train views on a ring at two elevations plus two held-out views ("val")."""
    import json
    import math
    from nerf_amd.utils.png import write_png
    angle = 0.6911112070083618
    focal = 0.5 * W / math.tan(0.5 * angle)
    for split, thetas in (("train", [360.0 * i / n_train for i in range(n_train)]), ("val", [15.0, 195.0])):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for i, th in enumerate(thetas):
            c2w = O.pose_spherical(th, -30.0 if i % 2 == 0 else -50.0, 4.0)
            ro, rd = O.get_rays(H, W, focal, c2w)
            ro, rd = ro.reshape(-1, 3).double(), rd.reshape(-1, 3).double()
            b, c = (ro * rd).sum(-1), (ro * ro).sum(-1) - 1.0
            disc = b * b - c
            hit = disc > 0
            p = ro + rd * (-b - torch.sqrt(disc.clamp_min(0)))[:, None]
            rgba = torch.cat([torch.where(hit[:, None], 0.5 + 0.5 * p, torch.zeros_like(p)), hit.double()[:, None]], -1)
            img = (rgba.reshape(H, W, 4) * 255.0 + 0.5).clamp(0, 255).to(torch.uint8).numpy()
            write_png(os.path.join(root, split, f"r_{i}.png"), img)
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": c2w.double().tolist()})
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": angle, "frames": frames}, f)
