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
