import ctypes, sys, torch
from ctypes import c_int64, c_int, c_void_p
def run(path, M):
    lib = ctypes.CDLL(path)
    for n in ("nerf_mlp_saved_bytes",): pass
    lib.nerf_mlp_dz_bytes.restype = c_int64; lib.nerf_mlp_dz_bytes.argtypes=[c_int64]
    lib.nerf_mlp_partial_bytes.restype = c_int64; lib.nerf_mlp_partial_bytes.argtypes=[c_int64]
    lib.nerf_mlp_bwd_weights.restype = c_int
    lib.nerf_mlp_bwd_weights.argtypes=[c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
    n_wt = (M + 255)//256*8
    saved = torch.randint(0, 255, (n_wt*158*1024,), dtype=torch.uint8, device='cuda')
    dz = torch.randint(0, 255, (lib.nerf_mlp_dz_bytes(M),), dtype=torch.uint8, device='cuda')
    pb = lib.nerf_mlp_partial_bytes(M)
    part = torch.empty(pb, dtype=torch.uint8, device='cuda'); dp = torch.empty(595844, device='cuda')
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: c_void_p(t.data_ptr())
    for _ in range(3): assert lib.nerf_mlp_bwd_weights(M, P(saved), P(dz), 0, None, P(part), pb, P(dp), st) == 0
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): lib.nerf_mlp_bwd_weights(M, P(saved), P(dz), 0, None, P(part), pb, P(dp), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/10
    gb = n_wt*356*1024/1e9
    print(f"{path.split('/')[-1]} M={M}: {ms*1e3:.1f} us, {gb/ms*1e3:.0f} GB/s (356 KB/wave-tile), partial {pb/1e6:.1f} MB")
for M in (65536, 196608, 1048576):
    for p in sys.argv[1:]:
        run(p, M)
