"""Stand-alone timings of the three training MLP kernels at the coarse and fine sizes of a 1024-ray step
(HIP events, real activations), and -- with a library built with -DNERF_WG_STAMP -- the per-workgroup
start/end stamps of the weight-gradient kernel."""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
from util import O, amd_model
from nerf_amd import ops, _C
lib, _p, check = ops.lib, ops._p, ops.check
DEV = torch.device('cuda:0')
prec = 'bf16'
m = amd_model(O.init_params(0), prec)
packed = m.packed_weights()
st = torch.cuda.current_stream().cuda_stream
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def ev(fn, iters=ITERS):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

for N, S in ((1024, 64), (1024, 192), (4096, 192)):
    M = N * S
    ro = torch.randn(N, 3, device=DEV); rd = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    z = torch.sort(torch.rand(N, S, device=DEV) * 4 + 2, dim=-1).values.contiguous()
    raw = torch.empty(N, S, 4, device=DEV)
    saved = torch.empty(int(lib.nerf_mlp_saved_bytes(M)), dtype=torch.uint8, device=DEV)
    dz = torch.empty(int(lib.nerf_mlp_dz_bytes(M)), dtype=torch.uint8, device=DEV)
    import ctypes
    one = (_C.WgradNet * 1)()
    one[0].M, one[0].precision = M, _C.PREC[prec]
    pb = int(lib.nerf_mlp_bwd_weights_multi_bytes(1, ctypes.cast(one, ctypes.c_void_p)))      # stamps follow what the library uses
    part = torch.zeros(pb + 64 * 1024, dtype=torch.uint8, device=DEV)
    d_raw = torch.randn(N, S, 4, device=DEV) * 1e-3
    d_flat = torch.empty(_C.NERF_MLP_NPARAMS, device=DEV)
    P = _C.PREC[prec]
    f = lambda: check(lib.nerf_mlp_fwd(N, S, _p(ro), _p(rd), _p(z), _p(packed), P, _p(raw), _p(saved), st), "fwd")
    d = lambda: check(lib.nerf_mlp_bwd_data(M, _p(d_raw), _p(packed), P, _p(saved), None, _p(dz), st), "dgrad")
    w = lambda: check(lib.nerf_mlp_bwd_weights(M, _p(packed), _p(saved), _p(dz), P, None, _p(part), pb + 64 * 1024, _p(d_flat), st), "wgrad")
    tf, td, tw = ev(f), ev(d), ev(w)
    print(f"M={M:7d}  fwd_train {tf:7.1f} us ({M*4832/tf/1e3:6.0f} GB/s wr)  dgrad {td:7.1f} us ({M*4416/td/1e3:6.0f} GB/s wr)  "
          f"wgrad+reduce+post {tw:7.1f} us ({M*8960/tw/1e3:6.0f} GB/s rd)", flush=True)
    torch.cuda.synchronize()
    stamps = part[pb:pb + 16 * 1024].view(torch.int64).cpu()
    if bool((stamps != 0).any()):
        n_jobs = int((stamps[0::2] > 0).sum())
        s0, s1 = stamps[0:2 * n_jobs:2], stamps[1:2 * n_jobs:2]
        xcc = s0 >> 56
        s0 = s0 & ((1 << 56) - 1)
        t0 = int(s0.min())
        dur = (s1 - s0).double() / 100.0
        end = (s1 - t0).double() / 100.0
        print(f"   wgrad stamps: jobs {n_jobs}, kernel span {float(end.max()):.1f} us, job duration min/median/max "
              f"{float(dur.min()):.1f}/{float(dur.median()):.1f}/{float(dur.max()):.1f} us")
        # per-GEMM cost per wave tile from a least-squares fit of the pieces' durations (the library's cut rule)
        import numpy as np
        cost = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [67, 131, 131, 131, 131, 146, 131, 131, 127, 24]
        tot = sum(cost); n_wt = (M + 255) // 256 * 8
        bound = lambda p: p * n_wt * tot // 256
        def b2w(g, b):
            before = sum(cost[:g])
            if b < n_wt * before: return 0
            return min(n_wt, (b - n_wt * before) // cost[g])
        A = np.zeros((256, 10))
        for pc in range(256):
            for g in range(10):
                A[pc, g] = max(0, b2w(g, bound(pc + 1)) - b2w(g, bound(pc)))
        d = dur.numpy()[:256]
        fit, *_ = np.linalg.lstsq(A, d, rcond=None)
        print("     fitted us per wave tile:", " ".join(f"{x:.3f}" for x in fit))
        print("     as cost (gemm 1 = 131): ", ",".join(f"{x / fit[1] * 131:.0f}" for x in fit))
        print("     piece durations (every 8th):", " ".join(f"{x:.0f}" for x in d[::8]))
        for x in range(8):
            mk = xcc == x
            print(f"     xcc {x}: {int(mk.sum())} jobs, mean duration {float(dur[mk].mean()):.1f}")
