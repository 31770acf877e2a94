"""Checkpoint files in the reference's wire format.

``scripts/train_nerf.py:118-131`` stores one ``torch.save`` dict per checkpoint with the keys
``epoch``, ``global_step``, ``args_saved_at_checkpoint``, ``H``, ``W``, ``focal``,
``model_coarse_state_dict``, ``optimizer_state_dict`` and (hierarchical runs) ``model_fine_state_dict``;
``scripts/render_nerf.py:184-252`` rebuilds encoders and models from it.  The functions here write and
read that dict, so checkpoints move between the two code bases in both directions.  Files are always
read with ``torch.load(..., weights_only=True)``: nothing in the file is executed.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Mapping, Optional, Tuple

import torch

from .models.nerf import NeRF, PositionalEncoder

#: model-defining entries of ``args_saved_at_checkpoint`` and the defaults of train_nerf.py:60-79
ARCH_DEFAULTS = {"net_depth": 8, "net_width": 256, "input_ch_pts_L": 10, "input_ch_views_L": 4,
                 "use_viewdirs": False, "use_hierarchical": False}
_PLAIN = (bool, int, float, str, type(None))


def _plain(v):
    """args as the reference saves them are argparse values: keep what a safe loader accepts."""
    if isinstance(v, _PLAIN):
        return v
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, Mapping):
        return {str(k): _plain(x) for k, x in v.items()}
    return str(v)


def checkpoint_name(epoch: int, global_step: int, is_final: bool = False) -> str:
    """File names of train_nerf.py:112-114."""
    stem = "checkpoint_final" if is_final else "checkpoint"
    return f"{stem}_epoch_{epoch:04d}_step_{global_step:07d}.pth"


def save_checkpoint(path: str, *, epoch: int, global_step: int, args: Mapping[str, Any], H: int, W: int,
                    focal: float, model_coarse: torch.nn.Module, optimizer, model_fine: Optional[torch.nn.Module] = None) -> str:
    """Write one checkpoint file (train_nerf.py:118-131); tensors are stored on the CPU."""
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}   # noqa: E731
    opt = optimizer.state_dict()
    opt["state"] = {k: {n: (t.detach().cpu().clone() if torch.is_tensor(t) else t) for n, t in st.items()}
                    for k, st in opt["state"].items()}
    d: Dict[str, Any] = {"epoch": int(epoch), "global_step": int(global_step),
                         "args_saved_at_checkpoint": _plain(dict(args)), "H": int(H), "W": int(W),
                         "focal": float(focal), "model_coarse_state_dict": cpu(model_coarse.state_dict()),
                         "optimizer_state_dict": opt}
    if model_fine is not None:
        d["model_fine_state_dict"] = cpu(model_fine.state_dict())
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    torch.save(d, tmp)
    os.replace(tmp, path)                     # a crash never leaves a half-written checkpoint behind
    return path


def load_checkpoint(path: str, map_location="cpu") -> Dict[str, Any]:
    """Read a checkpoint written by either code base.  ``weights_only=True``: a file holding anything
    but tensors and plain containers is refused by torch rather than unpickled."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    if not isinstance(ckpt, dict) or "model_coarse_state_dict" not in ckpt:
        raise ValueError(f"{path} is not a NeRF checkpoint (no 'model_coarse_state_dict')")
    saved = ckpt.get("args_saved_at_checkpoint") or {}
    if not isinstance(saved, dict):
        saved = vars(saved)
    ckpt["args_saved_at_checkpoint"] = saved
    return ckpt


def build_models(ckpt: Mapping[str, Any], device="cuda", precision: str = "bf16",
                 overrides: Optional[Mapping[str, Any]] = None
                 ) -> Tuple[NeRF, Optional[NeRF], PositionalEncoder, Optional[PositionalEncoder], Dict[str, Any]]:
    """Encoders and models as render_nerf.py:188-252 rebuilds them: the architecture comes from the saved
    args (``overrides`` fills what they lack, then the training script's defaults), the fine model is
    returned only when the run was hierarchical and its weights are in the file.
    Returns (model_coarse, model_fine or None, embed_fn_pts, embed_fn_views or None, resolved args)."""
    saved = dict(ckpt.get("args_saved_at_checkpoint") or {})
    cfg = {k: saved.get(k, (overrides or {}).get(k, dflt)) for k, dflt in ARCH_DEFAULTS.items()}
    pe = PositionalEncoder(3, int(cfg["input_ch_pts_L"]), True).to(device)
    pv = PositionalEncoder(3, int(cfg["input_ch_views_L"]), True).to(device) if cfg["use_viewdirs"] else None
    kw = dict(D=int(cfg["net_depth"]), W=int(cfg["net_width"]), input_ch_pts=pe.output_dims,
              input_ch_views=pv.output_dims if pv is not None else 0, skips=[int(cfg["net_depth"]) // 2],
              use_viewdirs=bool(cfg["use_viewdirs"]))
    # (an architecture outside the fused kernels keeps the requested operand type: it runs on the layer-program
    # kernels; NeRF.effective_precision() says what a network beyond those runs in)
    mc = NeRF(precision=precision, **kw)
    mc.load_state_dict(ckpt["model_coarse_state_dict"])
    mc = mc.to(device).eval()
    mf = None
    if cfg["use_hierarchical"] and ckpt.get("model_fine_state_dict") is not None:
        mf = NeRF(precision=precision, **kw)
        mf.load_state_dict(ckpt["model_fine_state_dict"])
        mf = mf.to(device).eval()
    cfg["precision"] = mc.effective_precision()
    return mc, mf, pe, pv, cfg
