"""CPU: the shared library loads without a GPU and exports every symbol that include/nerf_hip.h declares,
argument validation works before any launch, and the Python binding table matches the header."""
import ctypes
import os
import re

import pytest

from util import ROOT

HEADER = os.path.join(ROOT, "include", "nerf_hip.h")
# NERF_TEST_LIB: another build of the same library (the host-side sanitizer build of `make asan`)
LIB = os.environ.get("NERF_TEST_LIB") or os.path.join(ROOT, "nerf_amd", "lib", "libnerf_hip.so")


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
    assert lib.nerf_composite_fwd(4, 1, None, None, 0, None, None, None, None, None, None, None) == -2      # one sample
    assert lib.nerf_composite_fwd(-1, 64, None, None, 0, None, None, None, None, None, None, None) == -2
    assert lib.nerf_composite_fwd(4, 64, None, None, 0, None, None, None, None, None, None, None) == -1     # NULL
    assert lib.nerf_composite_fwd(4, 5000, None, None, 0, None, None, None, None, None, None, None) == -1   # no upper limit on S
    assert lib.nerf_composite_fwd(4, 64, ctypes.c_void_p(8), *([ctypes.c_void_p(16)] * 1), 0,
                                  *([ctypes.c_void_p(16)] * 6), None) == -3                                   # raw not 16-byte aligned
    lib.nerf_sort_rows.restype = ctypes.c_int
    lib.nerf_sort_rows.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.nerf_sort_rows(1, 2000, ctypes.c_void_p(16), 0, None, ctypes.c_void_p(16), None) == -2
    lib.nerf_mlp_packed_bytes.restype = ctypes.c_int64
    assert lib.nerf_mlp_packed_bytes() == (1216 + 1120) * 1024 + 13312 + 1184 * 1024 + (256 * 256 + 256 + 128 * 256) * 4 + 4096
    lib.nerf_mlp_saved_bytes.restype = ctypes.c_int64
    lib.nerf_mlp_saved_bytes.argtypes = [ctypes.c_int64]
    assert lib.nerf_mlp_saved_bytes(1) == 8 * (142 + 9) * 1024 and lib.nerf_mlp_saved_bytes(257) == 16 * (142 + 9) * 1024


def test_fp32_program_is_validated_on_the_host():
    """nerf_mlp_fp32_fwd checks every index of the layer list before it launches: a layer that reads more features
    of a hidden buffer than its producer wrote, a source that nothing wrote, or a program that leaves a column of
    raw unwritten is refused with NERF_E_ARCH."""
    from nerf_amd import _C
    lib = ctypes.CDLL(LIB)
    lib.nerf_mlp_fp32_fwd.restype = ctypes.c_int
    lib.nerf_mlp_fp32_fwd.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    FAKE = 0x10000

    def layer(k, src, out, dst, col=0):
        L = _C.Fp32Layer()
        L.n_segments = 1
        L.weights[0], L.k_padded[0], L.src[0] = FAKE, k, src
        L.bias, L.out_features, L.relu, L.dst, L.dst_col = FAKE, out, 1, dst, col
        return L

    def run(layers):
        arr = (_C.Fp32Layer * len(layers))(*layers)
        return lib.nerf_mlp_fp32_fwd(64, FAKE, 63, None, 0, ctypes.cast(arr, ctypes.c_void_p), len(layers), FAKE, None)

    A, B, RAW, PTS = _C.FP32_BUF_A, _C.FP32_BUF_B, _C.FP32_DST_RAW, _C.FP32_BUF_PTS
    assert run([layer(64, PTS, 100, A), layer(136, A, 4, RAW)]) == -4       # A holds 128 features (4 row tiles), not 136
    assert run([layer(64, PTS, 100, A), layer(104, B, 4, RAW)]) == -4       # nothing wrote B
    assert run([layer(64, PTS, 100, A), layer(104, A, 3, RAW)]) == -4       # column 3 of raw stays unwritten
    assert run([layer(72, PTS, 100, A), layer(104, A, 4, RAW)]) == -4       # 63 encoded features pad to 64, not 72
    assert run([layer(64, PTS, 100, A), layer(104, A, 3, RAW, 2)]) == -4    # columns 2..4


def test_fp32_backward_arguments_are_validated_on_the_host():
    """nerf_mlp_fp32_bwd: sizes, the gradient descriptors and the workspace are checked before anything is launched"""
    from nerf_amd import _C
    lib = ctypes.CDLL(LIB)
    lib.nerf_mlp_fp32_bwd.restype = ctypes.c_int
    lib.nerf_mlp_fp32_bwd.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    for fn in (lib.nerf_mlp_fp32_saved_floats, lib.nerf_mlp_fp32_bwd_workspace_floats):
        fn.restype = ctypes.c_int64
        fn.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    FAKE = 0x10000
    A, RAW, PTS = _C.FP32_BUF_A, _C.FP32_DST_RAW, _C.FP32_BUF_PTS
    layers = (_C.Fp32Layer * 2)()
    grads = (_C.Fp32Grad * 2)()
    for L, (k, src, out, dst) in zip(layers, ((64, PTS, 100, A), (104, A, 4, RAW))):
        L.n_segments = 1
        L.weights[0], L.k_padded[0], L.src[0] = FAKE, k, src
        L.bias, L.out_features, L.relu, L.dst, L.dst_col = FAKE, out, 1, dst, 0
    for G, (ld, cols) in zip(grads, ((63, 63), (100, 100))):
        G.d_weight, G.d_bias, G.ld = FAKE, FAKE, ld
        G.col0[0], G.cols[0] = 0, cols
    grads[1].wt[0] = FAKE
    lp, gp = ctypes.cast(layers, ctypes.c_void_p), ctypes.cast(grads, ctypes.c_void_p)
    rows = 64 + 0 + 100                                   # padded points, no views, one hidden layer
    assert lib.nerf_mlp_fp32_saved_floats(130, 63, 0, lp, 2) == rows * 192      # 130 samples -> 3 tiles of 64
    need = lib.nerf_mlp_fp32_bwd_workspace_floats(130, 63, 0, lp, 2)
    assert need > (8 + 100) * 192
    call = lambda M, ws, n: lib.nerf_mlp_fp32_bwd(M, 63, 0, lp, gp, 2, FAKE, FAKE, ws, n, None)   # noqa: E731
    assert call(-1, FAKE, need) == -2
    assert call(130, FAKE, need - 1) == -5                # workspace too small
    assert call(130, None, need) == -1
    assert call(130, FAKE + 4, need) == -3
    grads[1].wt[0] = None                                 # a hidden source needs its transposed weights
    assert call(130, FAKE, need) == -1
    grads[1].wt[0] = FAKE
    grads[0].cols[0] = 50                                 # does not round up to the layer's k_padded
    assert call(130, FAKE, need) == -4
    grads[0].cols[0] = 63
    grads[1].ld = 90                                      # the segment's columns do not fit the row
    assert call(130, FAKE, need) == -4


def test_p16_program_arguments_are_validated_on_the_host():
    """The 16-bit layer-program entry points: the same layer lists with input features padded to 16, sizes in bytes,
    and the operand type checked before anything else."""
    from nerf_amd import _C
    lib = ctypes.CDLL(LIB)
    i64, i32, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
    lib.nerf_mlp_p16_fwd.restype = i32
    lib.nerf_mlp_p16_fwd.argtypes = [i64, vp, i32, vp, i32, vp, i32, i32, vp, vp]
    lib.nerf_mlp_p16_bwd.restype = i32
    lib.nerf_mlp_p16_bwd.argtypes = [i64, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp]
    lib.nerf_mlp_p16_segment_bytes.restype = i64
    lib.nerf_mlp_p16_segment_bytes.argtypes = [i32, i32]
    lib.nerf_mlp_p16_pack_segment.restype = i32
    lib.nerf_mlp_p16_pack_segment.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    for fn in (lib.nerf_mlp_p16_saved_bytes, lib.nerf_mlp_p16_bwd_workspace_bytes):
        fn.restype = i64
        fn.argtypes = [i64, i32, i32, vp, i32]
    FAKE = 0x10000
    A, RAW, PTS = _C.FP32_BUF_A, _C.FP32_DST_RAW, _C.FP32_BUF_PTS
    assert lib.nerf_mlp_p16_segment_bytes(100, 63) == 4 * 4 * 1024            # 4 row tiles x 4 chunks of 16 inputs
    assert lib.nerf_mlp_p16_segment_bytes(0, 63) == 0
    assert lib.nerf_mlp_p16_pack_segment(FAKE, 100, 63, 0, 63, 7, FAKE, None) == -6      # unknown operand type
    assert lib.nerf_mlp_p16_pack_segment(FAKE, 100, 63, 1, 63, 0, FAKE, None) == -2      # columns beyond the row
    layers = (_C.Fp32Layer * 2)()
    grads = (_C.Fp32Grad * 2)()
    for L, (k, src, out, dst) in zip(layers, ((64, PTS, 100, A), (112, A, 4, RAW))):
        L.n_segments = 1
        L.weights[0], L.k_padded[0], L.src[0] = FAKE, k, src
        L.bias, L.out_features, L.relu, L.dst, L.dst_col = FAKE, out, 1, dst, 0
    for G, (ld, cols) in zip(grads, ((63, 63), (100, 100))):
        G.d_weight, G.d_bias, G.ld = FAKE, FAKE, ld
        G.col0[0], G.cols[0] = 0, cols
    grads[1].wt[0] = FAKE
    lp, gp = ctypes.cast(layers, vp), ctypes.cast(grads, vp)
    fwd = lambda prec: lib.nerf_mlp_p16_fwd(64, FAKE, 63, None, 0, lp, 2, prec, FAKE, None)   # noqa: E731
    assert fwd(2) == -6
    layers[1].k_padded[0] = 104                           # a multiple of 8 is not enough here
    assert fwd(0) == -4
    layers[1].k_padded[0] = 144                           # A holds 128 features (4 row tiles)
    assert fwd(0) == -4
    layers[1].k_padded[0] = 112
    rows = 64 + 0 + 100                                   # padded points, no views, one hidden layer
    assert lib.nerf_mlp_p16_saved_bytes(130, 63, 0, lp, 2) == rows * 256 * 2       # 130 samples -> 2 tiles of 128
    need = lib.nerf_mlp_p16_bwd_workspace_bytes(130, 63, 0, lp, 2)
    assert need > (16 + 100) * 256 * 2 and need % 4 == 0
    call = lambda M, ws, n, prec=1: lib.nerf_mlp_p16_bwd(M, 63, 0, lp, gp, 2, prec, FAKE, FAKE, None, ws, n, None)   # noqa: E731
    assert call(-1, FAKE, need) == -2
    assert call(130, FAKE, need, 5) == -6
    assert call(130, FAKE, need - 1) == -5                # workspace too small
    assert call(130, None, need) == -1
    assert call(130, FAKE + 16, need) == -3               # 256-byte alignment
    grads[1].wt[0] = None                                 # a hidden source needs its transposed weights
    assert call(130, FAKE, need) == -1
    grads[1].wt[0] = FAKE
    grads[0].cols[0] = 40                                 # does not round up to the layer's k_padded
    assert call(130, FAKE, need) == -4


def test_cpu_tensors_are_rejected():
    """the kernels have no CPU path: CPU tensors raise before reaching the library (get_rays alone serves a pose
    in host memory, with torch CPU ops, as the reference's loader workers need)"""
    import torch
    import nerf_amd
    from nerf_amd import ops
    with pytest.raises(RuntimeError):
        ops.get_rays(4, 4, 5.0, torch.eye(4))
    assert nerf_amd.get_rays(4, 4, 5.0, torch.eye(4))[0].device.type == "cpu"
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
