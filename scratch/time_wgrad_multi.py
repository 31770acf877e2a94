"""Weight-gradient pass of a 1024-ray step: two separate calls (coarse 65 536 + fine 196 608 samples) against one
merged call; with a -DNERF_WG_STAMP library also the per-workgroup spans of the merged launch."""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import ctypes, torch
from util import O, amd_model
from nerf_amd import ops, _C
lib, _p, check = ops.lib, ops._p, ops.check
DEV = torch.device('cuda:0')
prec = 'bf16'
P = _C.PREC[prec]
st = torch.cuda.current_stream().cuda_stream
nets = []
for seed, (N, S) in enumerate(((1024, 64), (1024, 192))):
    m = amd_model(O.init_params(seed), prec)
    packed = m.packed_weights()
    M = N * S
    ro = torch.randn(N, 3, device=DEV); rd = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    z = torch.sort(torch.rand(N, S, device=DEV) * 4 + 2, dim=-1).values.contiguous()
    raw = torch.empty(N, S, 4, device=DEV)
    saved = torch.empty(int(lib.nerf_mlp_saved_bytes(M)), dtype=torch.uint8, device=DEV)
    dz = torch.empty(int(lib.nerf_mlp_dz_bytes(M)), dtype=torch.uint8, device=DEV)
    d_raw = torch.randn(N, S, 4, device=DEV) * 1e-3
    check(lib.nerf_mlp_fwd(N, S, _p(ro), _p(rd), _p(z), _p(packed), P, _p(raw), _p(saved), st), "fwd")
    check(lib.nerf_mlp_bwd_data(M, _p(d_raw), _p(packed), P, _p(saved), None, _p(dz), st), "dgrad")
    nets.append(dict(M=M, packed=packed, saved=saved, dz=dz, d=torch.empty(_C.NERF_MLP_NPARAMS, device=DEV)))

def ev(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

parts = [torch.empty(int(lib.nerf_mlp_partial_bytes(n["M"])), dtype=torch.uint8, device=DEV) for n in nets]
def separate():
    for n, part in zip(nets, parts):
        check(lib.nerf_mlp_bwd_weights(n["M"], _p(n["packed"]), _p(n["saved"]), _p(n["dz"]), P, None, _p(part), part.numel(), _p(n["d"]), st), "w")
arr = (_C.WgradNet * 2)()
for i, n in enumerate(nets):
    arr[i].M, arr[i].packed, arr[i].saved, arr[i].dz = n["M"], n["packed"].data_ptr(), n["saved"].data_ptr(), n["dz"].data_ptr()
    arr[i].precision, arr[i].grad_scale, arr[i].d_params = P, None, n["d"].data_ptr()
nb = int(lib.nerf_mlp_bwd_weights_multi_bytes(2, ctypes.cast(arr, ctypes.c_void_p)))
part2 = torch.zeros(nb + 64 * 1024, dtype=torch.uint8, device=DEV)
def merged():
    check(lib.nerf_mlp_bwd_weights_multi(2, ctypes.cast(arr, ctypes.c_void_p), _p(part2), part2.numel(), st), "wm")
separate(); torch.cuda.synchronize()
g_sep = [n["d"].clone() for n in nets]
merged(); torch.cuda.synchronize()
for a, n in zip(g_sep, nets):
    print("merged vs separate gradients: max rel diff %.2e" % ((a - n["d"]).abs().max() / a.abs().max()).item())
print(f"separate {ev(separate):7.1f} us   merged {ev(merged):7.1f} us   (partial bytes {sum(p.numel() for p in parts)/1e6:.1f} MB vs {nb/1e6:.1f} MB)")
stamps = part2[nb:nb + 16 * 1024].view(torch.int64).cpu()
if bool((stamps != 0).any()):
    s0, s1 = stamps[0:512:2], stamps[1:512:2]
    s0 = s0 & ((1 << 56) - 1)
    t0 = int(s0.min())
    dur = (s1 - s0).double() / 100.0
    print("merged launch: span %.1f us, piece duration min/median/max %.1f/%.1f/%.1f" % (float((s1 - t0).max()) / 100.0, float(dur.min()), float(dur.median()), float(dur.max())))
    print("piece durations:", " ".join(f"{float(x):.0f}" for x in dur[::4]))
