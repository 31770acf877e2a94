"""Host-side sweep run under the sanitizer runtime by tests/test_asan_host.py (not collected by pytest).

Calls the sizing functions and the argument-error paths of the library named by NERF_TEST_LIB over sample
counts from 1 to 10^6, both architectures, one and two networks.  Every call stays on the host: sizing
functions launch nothing, and the weight-gradient entry point is handed a workspace of zero bytes, which it
rejects after it has divided the work (the job table is what indexes fixed arrays from run-time sizes)."""
import ctypes
import os
import sys
from ctypes import c_int, c_int64, c_void_p

lib = ctypes.CDLL(os.environ["NERF_TEST_LIB"])


class WgradNet(ctypes.Structure):      # nerf_wgrad_net_t
    _fields_ = [("M", c_int64), ("packed", c_void_p), ("saved", c_void_p), ("dz", c_void_p), ("precision", c_int),
                ("grad_scale", c_void_p), ("d_params", c_void_p)]


for name in ("nerf_mlp_saved_bytes", "nerf_mlp_dz_bytes", "nerf_mlp_partial_bytes", "nerf_mlp_bwd_workspace_bytes",
             "nerf_mlp_encoded_workspace_bytes"):
    getattr(lib, name).restype = c_int64
    getattr(lib, name).argtypes = [c_int64]
lib.nerf_mlp_bwd_weights_multi_bytes.restype = c_int64
lib.nerf_mlp_bwd_weights_multi_bytes.argtypes = [c_int, c_void_p]
lib.nerf_mlp_bwd_weights_multi.restype = c_int
lib.nerf_mlp_bwd_weights_multi.argtypes = [c_int, c_void_p, c_void_p, c_int64, c_void_p]
lib.nerf_mlp_bwd.restype = c_int
lib.nerf_mlp_bwd.argtypes = [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]

NOVIEW = 0x100
FAKE = 0x10000          # a non-NULL, 256-byte aligned address that is never dereferenced on these paths
ms = set(range(1, 600)) | {10 ** 6, 10 ** 6 - 1, 999_937}
m = 600
while m < 10 ** 6:
    ms |= {m - 1, m, m + 1, (m // 256) * 256, (m // 256) * 256 + 1, (m // 32) * 32 - 1}
    m = m * 9 // 8 + 7
ms = sorted(x for x in ms if 1 <= x <= 10 ** 6)
calls = 0
for M in ms:
    sv, dz, pb, wb = (lib.nerf_mlp_saved_bytes(M), lib.nerf_mlp_dz_bytes(M), lib.nerf_mlp_partial_bytes(M),
                      lib.nerf_mlp_bwd_workspace_bytes(M))
    tiles = (M + 255) // 256
    assert sv == tiles * 8 * 151 * 1024 and dz == tiles * 8 * 138 * 1024 and wb == dz + pb and pb > 0, (M, sv, dz, pb, wb)
    assert lib.nerf_mlp_encoded_workspace_bytes(M) == tiles * 8 * 6 * 1024
    for archs in ((0,), (NOVIEW,), (0, 0), (0, NOVIEW), (NOVIEW, 0), (NOVIEW, NOVIEW)):
        nets = (WgradNet * len(archs))()
        for n, a in enumerate(archs):
            # the second network of a pair gets another size, as coarse and fine do (64 and 192 samples per ray)
            nets[n].M = M if n == 0 else 3 * M
            nets[n].precision = a | (1 if M % 2 else 0)
            nets[n].packed = nets[n].saved = nets[n].dz = nets[n].d_params = FAKE
        nb = lib.nerf_mlp_bwd_weights_multi_bytes(len(archs), ctypes.cast(nets, c_void_p))
        assert nb > 0 and nb % 256 == 0, (M, archs, nb)
        if len(archs) == 1:
            assert nb <= pb
        # the same job table inside the launching entry point, stopped by the workspace check
        rc = lib.nerf_mlp_bwd_weights_multi(len(archs), ctypes.cast(nets, c_void_p), FAKE, 0, None)
        assert rc == -5, (M, archs, rc)
        rc = lib.nerf_mlp_bwd_weights_multi(len(archs), ctypes.cast(nets, c_void_p), FAKE, nb - 1, None)
        assert rc == -5, (M, archs, rc)
        calls += 3
    assert lib.nerf_mlp_bwd(M, FAKE, FAKE, 0, FAKE, None, FAKE, wb - 1, FAKE, None) == -5
# layer-program kernels (fp32 and 16-bit operands): sizing and the backward's host-side program construction, stopped
# by the workspace check, for a network with a skip connection and the colour head


class Layer(ctypes.Structure):         # nerf_fp32_layer_t
    _fields_ = [("weights", c_void_p * 2), ("k_padded", c_int * 2), ("src", c_int * 2), ("n_segments", c_int),
                ("bias", c_void_p), ("out_features", c_int), ("relu", c_int), ("dst", c_int), ("dst_col", c_int)]


class Grad(ctypes.Structure):          # nerf_fp32_grad_t
    _fields_ = [("wt", c_void_p * 2), ("d_weight", c_void_p), ("ld", c_int), ("col0", c_int * 2), ("cols", c_int * 2),
                ("d_bias", c_void_p)]


PTS, VIEWS, A, B, RAW = 0, 1, 2, 3, 4
for fam, kpad, unit in (("fp32", 8, "floats"), ("p16", 16, "bytes")):
    pad = lambda v: (v + kpad - 1) // kpad * kpad      # noqa: E731
    W, CP, CV = 100, 63, 27
    # (segments [(cols, src)], out, relu, dst, dst_col)
    net = [([(CP, PTS)], W, 1, A, 0), ([(CP, PTS), (W, A)], W, 1, B, 0), ([(W, B)], 1, 0, RAW, 3), ([(W, B)], W, 0, A, 0),
           ([(W, A), (CV, VIEWS)], W // 2, 1, B, 0), ([(W // 2, B)], 3, 0, RAW, 0)]
    layers, grads = (Layer * len(net))(), (Grad * len(net))()
    for L, G, (segs, out, relu, dst, col) in zip(layers, grads, net):
        L.n_segments = len(segs)
        c0 = 0
        for s, (cols, src) in enumerate(segs):
            L.weights[s], L.k_padded[s], L.src[s] = FAKE, pad(cols), src
            G.wt[s], G.col0[s], G.cols[s] = (FAKE if src in (A, B) else None), c0, cols
            c0 += cols
        L.bias, L.out_features, L.relu, L.dst, L.dst_col = FAKE, out, relu, dst, col
        G.d_weight, G.d_bias, G.ld = FAKE, FAKE, c0
    lp, gp = ctypes.cast(layers, c_void_p), ctypes.cast(grads, c_void_p)
    saved_fn, work_fn = getattr(lib, f"nerf_mlp_{fam}_saved_{unit}"), getattr(lib, f"nerf_mlp_{fam}_bwd_workspace_{unit}")
    for fn in (saved_fn, work_fn):
        fn.restype = c_int64
        fn.argtypes = [c_int64, c_int, c_int, c_void_p, c_int]
    bwd = getattr(lib, f"nerf_mlp_{fam}_bwd")
    bwd.restype = c_int
    if fam == "fp32":
        bwd.argtypes = [c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
    else:
        bwd.argtypes = [c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                        c_void_p]
    tile, esz = (64, 1) if fam == "fp32" else (128, 2)
    rows = pad(CP) + pad(CV) + 3 * W + W // 2
    for M in ms[::7]:
        ldm = (M + tile - 1) // tile * tile
        assert saved_fn(M, CP, CV, lp, len(net)) == rows * ldm * esz, (fam, M)
        need = work_fn(M, CP, CV, lp, len(net))
        assert need > (kpad + 3 * W + W // 2) * ldm * esz, (fam, M, need)
        if fam == "fp32":
            rc = bwd(M, CP, CV, lp, gp, len(net), FAKE, FAKE, FAKE, need - 1, None)
        else:
            rc = bwd(M, CP, CV, lp, gp, len(net), M % 2, FAKE, FAKE, None, FAKE, need - 1, None)
        assert rc == -5, (fam, M, rc)
        calls += 1
# argument errors
nets = (WgradNet * 2)()
nets[0].M, nets[1].M = 100, 100
nets[0].precision, nets[1].precision = 0, 1
assert lib.nerf_mlp_bwd_weights_multi_bytes(2, ctypes.cast(nets, c_void_p)) == -1       # mixed operand types
assert lib.nerf_mlp_bwd_weights_multi_bytes(3, ctypes.cast(nets, c_void_p)) == -1       # too many networks
assert lib.nerf_mlp_bwd_weights_multi_bytes(0, ctypes.cast(nets, c_void_p)) == -1
assert lib.nerf_mlp_bwd_weights_multi_bytes(1, None) == -1
nets[0].M = 0
assert lib.nerf_mlp_bwd_weights_multi_bytes(1, ctypes.cast(nets, c_void_p)) == -1
assert lib.nerf_mlp_bwd(-1, FAKE, FAKE, 0, FAKE, None, FAKE, 0, FAKE, None) == -2
assert lib.nerf_mlp_bwd(5, FAKE, FAKE, 7, FAKE, None, FAKE, 0, FAKE, None) == -6
assert lib.nerf_mlp_bwd(5, FAKE, FAKE, 0, FAKE, None, None, 0, FAKE, None) == -1
assert lib.nerf_mlp_bwd(5, FAKE, FAKE, 0, FAKE, None, FAKE + 8, 1 << 40, FAKE, None) == -3
print("asan sweep ok: %d sample counts, %d job tables" % (len(ms), calls))
sys.exit(0)
