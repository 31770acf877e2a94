"""Event timings of the bandwidth-bound kernels at several launch sizes."""
import sys, torch
sys.path.insert(0, '.')
from nerf_amd import ops

dev = torch.device('cuda:0')
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

for n in (4096, 65536, 640000):
    for S in (64, 192):
        raw = torch.randn(n, S, 4, device=dev)
        z = torch.sort(torch.rand(n, S, device=dev) * 4 + 2, dim=-1).values
        with torch.no_grad():
            us = timeit(lambda: ops.composite(raw, z, False))
        byt = n * S * (16 + 4 + 4 + 4) + n * 24   # raw, z in; weights, alpha out; maps
        print(f"composite_fwd n={n} S={S}: {us:8.1f} us  {byt / us / 1e3:7.1f} GB/s", flush=True)
        if n <= 65536:
            raw.requires_grad_(True)
            out = ops.composite(raw, z, False)
            g = [torch.randn_like(o) for o in out]
            def bw():
                raw.grad = None
                torch.autograd.backward(out, g, retain_graph=True)
            us = timeit(bw)
            print(f"composite_bwd n={n} S={S}: {us:8.1f} us", flush=True)
        del raw, z
for n in (4096, 65536):
    zc = torch.sort(torch.rand(n, 64, device=dev) * 4 + 2, dim=-1).values
    w = torch.rand(n, 64, device=dev)
    with torch.no_grad():
        us = timeit(lambda: ops.sample_merge(zc, w, 128, True))
    print(f"sample_merge det n={n}: {us:8.1f} us", flush=True)
    with torch.no_grad():
        us = timeit(lambda: ops.sample_merge(zc, w, 128, False))
    print(f"sample_merge rand n={n}: {us:8.1f} us", flush=True)
c2w = torch.eye(4, device=dev)[:3]
for H in (800, 3200):
    us = timeit(lambda: ops.get_rays(H, H, 1111.0, c2w))
    print(f"get_rays {H}x{H}: {us:8.1f} us  {H * H * 24 / us / 1e3:7.1f} GB/s", flush=True)
