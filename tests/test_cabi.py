"""CPU: the shared library loads without a GPU and exports every symbol that include/nerf_hip.h declares,
argument validation works before any launch, and the Python binding table matches the header."""
import ctypes
import os
import re

import pytest

from util import ROOT

HEADER = os.path.join(ROOT, "include", "nerf_hip.h")
LIB = os.path.join(ROOT, "nerf_amd", "lib", "libnerf_hip.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nerf_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for s in ("nerf_get_rays", "nerf_composite_fwd", "nerf_composite_bwd", "nerf_sample_pdf", "nerf_sort_rows",
              "nerf_sample_merge", "nerf_mlp_fwd", "nerf_mlp_bwd", "nerf_mlp_pack", "nerf_adam_step", "nerf_version"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build the library first: make -C nerf_amd/csrc"
    lib = ctypes.CDLL(LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.nerf_version.restype = ctypes.c_int
    assert lib.nerf_version() == 100
    lib.nerf_error_string.restype = ctypes.c_char_p
    assert lib.nerf_error_string(-2).decode().startswith("a size")


def test_binding_table_matches_header():
    from nerf_amd import _C
    assert sorted(_C.SYMBOLS) == declared_symbols()


def test_argument_errors_without_gpu():
    """negative return codes come from host-side validation: nothing is launched, no GPU is needed"""
    lib = ctypes.CDLL(LIB)
    lib.nerf_composite_fwd.restype = ctypes.c_int
    lib.nerf_composite_fwd.argtypes = [ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 7
    assert lib.nerf_composite_fwd(4, 2000, None, None, 0, None, None, None, None, None, None, None) == -2   # S too large
    assert lib.nerf_composite_fwd(4, 64, None, None, 0, None, None, None, None, None, None, None) == -1     # NULL
    lib.nerf_sort_rows.restype = ctypes.c_int
    lib.nerf_sort_rows.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.nerf_sort_rows(1, 2000, ctypes.c_void_p(16), 0, None, ctypes.c_void_p(16), None) == -2
    lib.nerf_mlp_packed_bytes.restype = ctypes.c_int64
    assert lib.nerf_mlp_packed_bytes() == (1216 + 1120) * 1024 + 13312 + 1184 * 1024 + (256 * 256 + 256 + 128 * 256) * 4
    lib.nerf_mlp_saved_bytes.restype = ctypes.c_int64
    lib.nerf_mlp_saved_bytes.argtypes = [ctypes.c_int64]
    assert lib.nerf_mlp_saved_bytes(1) == 8 * (142 + 9) * 1024 and lib.nerf_mlp_saved_bytes(257) == 16 * (142 + 9) * 1024


def test_cpu_tensors_are_rejected():
    """the product has no CPU path: CPU tensors raise before reaching the library"""
    import torch
    import nerf_amd
    with pytest.raises(RuntimeError):
        nerf_amd.get_rays(4, 4, 5.0, torch.eye(4))
    with pytest.raises(RuntimeError):
        nerf_amd.PositionalEncoder(3, 10)(torch.zeros(2, 3))
    m = nerf_amd.NeRF()
    assert [k for k, _ in m.named_parameters()][:2] == ["pts_linears.0.weight", "pts_linears.0.bias"]
    with pytest.raises(ValueError):
        m(torch.zeros(2, 63))          # same error as the reference when view encodings are missing


def test_hand_issued_loads_are_not_touched_before_their_wait(tmp_path):
    """The training forward loads the next pass's ray inputs by inline asm, which the compiler does not count:
    its disassembly must not read, copy or spill a destination register between the load and the counted wait
    (nerf_amd/csrc/audit_asm_loads.py; `make audit` does the same).  Needs the object files of the build."""
    import shutil
    import subprocess
    import sys
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    objs = [os.path.join(ROOT, "build", "obj", "mlp_fwd.tu%d.o" % t) for t in (1, 2)]
    if not all(os.path.exists(o) for o in objs) or not os.path.exists(objdump):
        pytest.skip("object files of the build are not here")
    for t, obj in enumerate(objs):
        d = tmp_path / str(t)
        d.mkdir()
        shutil.copy(obj, d / "k.o")
        subprocess.run([objdump, "--offloading", "k.o"], cwd=d, check=True, capture_output=True)
        code = [f for f in os.listdir(d) if f.startswith("k.o.") and "gfx950" in f]
        assert len(code) == 1, os.listdir(d)
        with open(d / "k.s", "w") as f:
            subprocess.run([objdump, "-d", code[0]], cwd=d, check=True, stdout=f)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "nerf_amd", "csrc", "audit_asm_loads.py"), str(d / "k.s")],
                           capture_output=True, text=True)
        assert r.returncode == 0 and " 0 early uses" in r.stdout, r.stdout + r.stderr
        assert int(r.stdout.split()[0]) >= 28, r.stdout        # the loads were found at all
