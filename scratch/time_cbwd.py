"""Device rate of the compositing backward at HBM-scale sizes (direct C-ABI calls)."""
import sys, torch
sys.path.insert(0, '.')
from nerf_amd import ops, _C
lib, check, _p = _C.lib, ops.check, ops._p
dev = torch.device('cuda:0')
for n, S in ((640000, 64), (250000, 192)):
    raw = torch.randn(n, S, 4, device=dev); z = 2.0 + torch.cumsum(torch.rand(n, S, device=dev), -1) * (4.0 / S)
    g_rgb = torch.randn(n, 3, device=dev); g_w = torch.randn(n, S, device=dev)
    d_raw = torch.empty_like(raw)
    st = torch.cuda.current_stream().cuda_stream
    def run(with_w):
        check(lib.nerf_composite_bwd(n, S, _p(raw), _p(z), 0, _p(g_rgb), None, None, None, _p(g_w) if with_w else None, None, _p(d_raw), st), "bwd")
    for with_w in (False, True):
        run(with_w); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run(with_w)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 5 * 1e3
        byt = n * (S * (16 + 4 + 16 + (4 if with_w else 0)) + 12)
        print(f"composite_bwd n={n} S={S} g_w={with_w}: {us:8.1f} us {byt / us / 1e3:7.1f} GB/s", flush=True)
