"""ctypes binding of libnerf_hip.so (the C ABI declared in include/nerf_hip.h).

There is no fallback: if the library has not been built, importing this module
raises, and every wrapper raises on a non-zero return code.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnerf_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP library first "
        "(`make -C nerf_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "nerf_amd has no CPU or PyTorch fallback."
    )

lib = ctypes.CDLL(LIB_PATH)

P = c_void_p
_SIGS = {
    "nerf_version": (c_int, []),
    "nerf_error_string": (c_char_p, [c_int]),
    "nerf_get_rays": (c_int, [c_int, c_int, c_float, P, c_int, c_int, c_int, P, P, P]),
    "nerf_gather_rays": (c_int, [c_int64, P, c_int, c_int, c_int, c_float, P, c_int, P, c_int, P, c_int, P, P, P, P]),
    "nerf_posenc": (c_int, [c_int64, c_int, c_int, P, P, P, P]),
    "nerf_place_samples": (c_int, [c_int64, c_int, c_float, c_float, P, P, c_int, P, P, P, P]),
    "nerf_composite_fwd": (c_int, [c_int64, c_int, P, P, c_int, P, P, P, P, P, P, P]),
    "nerf_composite_bwd": (c_int, [c_int64, c_int, P, P, c_int, P, P, P, P, P, P, P, P]),
    "nerf_sample_pdf": (c_int, [c_int64, c_int, P, c_int64, P, c_int64, c_int, P, P, P, P]),
    "nerf_sample_merge": (c_int, [c_int64, c_int, P, P, c_int, P, P, P, P]),
    "nerf_sort_rows": (c_int, [c_int64, c_int, P, c_int, P, P, P]),
    "nerf_mlp_packed_bytes": (c_int64, []),
    "nerf_mlp_pack": (c_int, [P, c_int, c_int, P, P]),
    "nerf_mlp_pack_multi": (c_int, [c_int, P, c_int, P, P, P]),
    "nerf_mlp_bwd_weights_multi_bytes": (c_int64, [c_int, P]),
    "nerf_mlp_bwd_weights_multi": (c_int, [c_int, P, P, c_int64, P]),
    "nerf_mlp_saved_bytes": (c_int64, [c_int64]),
    "nerf_mlp_bwd_workspace_bytes": (c_int64, [c_int64]),
    "nerf_mlp_fwd": (c_int, [c_int64, c_int, P, P, P, P, c_int, P, P, P]),
    "nerf_mlp_encoded_workspace_bytes": (c_int64, [c_int64]),
    "nerf_mlp_fwd_encoded": (c_int, [c_int64, P, P, P, c_int, P, P, P, P]),
    "nerf_mlp_fwd_encoded_static_max_samples": (c_int64, []),
    "nerf_mlp_fwd_encoded_static": (c_int, [c_int64, P, P, P, c_int, P, P]),
    "nerf_mlp_bwd": (c_int, [c_int64, P, P, c_int, P, P, P, c_int64, P, P]),
    "nerf_mlp_dz_bytes": (c_int64, [c_int64]),
    "nerf_mlp_partial_bytes": (c_int64, [c_int64]),
    "nerf_mlp_bwd_data": (c_int, [c_int64, P, P, c_int, P, P, P, P]),
    "nerf_mlp_bwd_weights": (c_int, [c_int64, P, P, P, c_int, P, P, c_int64, P, P]),
    "nerf_mlp_fp32_segment_floats": (c_int64, [c_int, c_int]),
    "nerf_mlp_fp32_pack_segment": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "nerf_mlp_fp32_fwd": (c_int, [c_int64, P, c_int, P, c_int, P, c_int, P, P]),
    "nerf_mlp_fp32_saved_floats": (c_int64, [c_int64, c_int, c_int, P, c_int]),
    "nerf_mlp_fp32_bwd_workspace_floats": (c_int64, [c_int64, c_int, c_int, P, c_int]),
    "nerf_mlp_fp32_fwd_train": (c_int, [c_int64, P, c_int, P, c_int, P, c_int, P, P, P]),
    "nerf_mlp_fp32_bwd": (c_int, [c_int64, c_int, c_int, P, P, c_int, P, P, P, c_int64, P]),
    "nerf_mlp_p16_segment_bytes": (c_int64, [c_int, c_int]),
    "nerf_mlp_p16_pack_segment": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "nerf_mlp_p16_fwd": (c_int, [c_int64, P, c_int, P, c_int, P, c_int, c_int, P, P]),
    "nerf_mlp_p16_saved_bytes": (c_int64, [c_int64, c_int, c_int, P, c_int]),
    "nerf_mlp_p16_bwd_workspace_bytes": (c_int64, [c_int64, c_int, c_int, P, c_int]),
    "nerf_mlp_p16_fwd_train": (c_int, [c_int64, P, c_int, P, c_int, P, c_int, c_int, P, P, P]),
    "nerf_mlp_p16_bwd": (c_int, [c_int64, c_int, c_int, P, P, c_int, c_int, P, P, P, P, c_int64, P]),
    "nerf_adam_step_dev": (c_int, [c_int64, P, P, P, P, P, P]),
    "nerf_mse_pair": (c_int, [c_int64, P, P, P, P, P, P, P]),
    "nerf_adam_step": (c_int, [c_int64, P, P, P, P, c_float, c_float, c_float, c_float, c_int64, c_float, P]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here = header and library disagree
    _fn.restype = _res
    _fn.argtypes = _args

SYMBOLS = tuple(_SIGS)


class WgradNet(ctypes.Structure):
    """nerf_wgrad_net_t of include/nerf_hip.h: one network of a multi-network weight-gradient call."""
    _fields_ = [("M", c_int64), ("packed", c_void_p), ("saved", c_void_p), ("dz", c_void_p), ("precision", c_int),
                ("grad_scale", c_void_p), ("d_params", c_void_p)]


WGRAD_MAX_NETS = 2


class Fp32Layer(ctypes.Structure):
    """nerf_fp32_layer_t of include/nerf_hip.h: one layer of the fp32 forward's program."""
    _fields_ = [("weights", c_void_p * 2), ("k_padded", c_int * 2), ("src", c_int * 2), ("n_segments", c_int),
                ("bias", c_void_p), ("out_features", c_int), ("relu", c_int), ("dst", c_int), ("dst_col", c_int)]


class Fp32Grad(ctypes.Structure):
    """nerf_fp32_grad_t of include/nerf_hip.h: where one layer's gradients go, and its transposed weights."""
    _fields_ = [("wt", c_void_p * 2), ("d_weight", c_void_p), ("ld", c_int), ("col0", c_int * 2), ("cols", c_int * 2),
                ("d_bias", c_void_p)]


FP32_MAX_WIDTH, FP32_MAX_PTS, FP32_MAX_VIEWS, FP32_MAX_LAYERS = 256, 64, 32, 16
FP32_BUF_PTS, FP32_BUF_VIEWS, FP32_BUF_A, FP32_BUF_B, FP32_DST_RAW = 0, 1, 2, 3, 4
NERF_MLP_NPARAMS = 595844
NERF_MLP_NPARAMS_NOVIEW = 494084
ARCH_NOVIEW = 0x100          # or-ed into the precision argument: the network without view directions
PREC = {"bf16": 0, "fp16": 1}


class NerfHipError(RuntimeError):
    pass


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib.nerf_error_string(int(code))
        raise NerfHipError(f"{what} failed with code {code}: {msg.decode() if msg else '?'}")
