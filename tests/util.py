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
