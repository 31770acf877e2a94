import sys, torch
sys.path.insert(0, '.')
from nerf_amd import ops
dev = torch.device('cuda:0')
c2w = torch.eye(4, device=dev)[:3].contiguous()
for H in (800, 1600, 3200, 6400):
    for _ in range(3): ops.get_rays(H, H, 1111.0, c2w)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.get_rays(H, H, 1111.0, c2w)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 10 * 1e3
    print(f"get_rays {H}x{H}: {us:8.1f} us  {H * H * 24 / us / 1e3:7.1f} GB/s", flush=True)
