"""CPU restatement of the reference's Blender loader.  TEST INFRASTRUCTURE ONLY: imported by tests,
never by the product package.

Follows src/nerf_project/datasets/blender.py: JSON and path handling :56-86, x/255 and the alpha blend
:92-104, focal :113, the per-image item :146-164.  The reference module itself cannot be imported in
this environment (it needs ``imageio``, which is not installed) and the reference ships no dataset
files or loader fixtures, so this restatement is checked against hand-computed values only:
PARITY UNPINNED for the loader.  Image decoding is passed in (``read_image``) so that the oracle does
not depend on the product's PNG reader.
"""
import json
import os

import numpy as np
import torch

from . import nerf_oracle as O


def blend_images(img_u8: np.ndarray, white: bool) -> np.ndarray:
    """uint8 [...,3|4] -> float32 [...,3] (blender.py:92-104)."""
    image = (np.array(img_u8) / 255.0).astype(np.float32)
    if image.shape[-1] == 4:
        alpha, rgb = image[..., -1:], image[..., :3]
        image = rgb * alpha + (1.0 - alpha) if white else rgb * alpha
    return image


def load_blender(dataset_path: str, split: str, data_white_bkgd: bool, read_image):
    with open(os.path.join(dataset_path, f"transforms_{split}.json"), "r") as f:
        meta = json.load(f)
    angle = float(meta["camera_angle_x"])
    paths, poses, rgbs = [], [], []
    for frame in meta["frames"]:
        rel = frame["file_path"]
        if rel.startswith("./"):
            rel = rel[2:]
        path = os.path.join(dataset_path, rel + ".png")
        if not os.path.exists(path):
            continue                                   # blender.py:79-86
        paths.append(path)
        poses.append(np.array(frame["transform_matrix"]))
        rgbs.append(blend_images(read_image(path), data_white_bkgd))
    H, W = rgbs[0].shape[:2]
    return {"image_paths": paths, "poses": torch.from_numpy(np.array(poses)).float(),
            "all_rgbs": torch.from_numpy(np.array(rgbs)).float(), "height": H, "width": W,
            "focal": 0.5 * W / np.tan(0.5 * angle)}


def item(ds, idx: int, split: str):
    """blender.py:146-164."""
    rays_o, rays_d = O.get_rays(ds["height"], ds["width"], ds["focal"], ds["poses"][idx])
    rgb = ds["all_rgbs"][idx]
    if split == "train":
        return {"rays_o": rays_o.reshape(-1, 3), "rays_d": rays_d.reshape(-1, 3), "target_rgb": rgb.reshape(-1, 3)}
    return {"rays_o": rays_o, "rays_d": rays_d, "target_rgb": rgb, "pose": ds["poses"][idx]}
