"""Time the fp32 validation kernel on 262144 samples of the 8x256 network (and check it against torch)."""
import sys, torch
sys.path.insert(0, ".")
from nerf_amd.models.nerf import NeRF
from nerf_amd.utils.synthetic import seed_parameters_
dev = torch.device("cuda:0")
m = seed_parameters_(NeRF(precision="fp32").to(dev), 0)
xp, xv = torch.randn(262144, 63, device=dev), torch.randn(262144, 27, device=dev)
with torch.no_grad():
    out = m(xp, xv)
    ref = m._forward_library(xp[:4096], xv[:4096], None) if hasattr(m, "_forward_library") else None
if ref is not None:
    print("max abs diff vs library GEMMs on 4096 samples: %.3e" % (out[:4096] - ref).abs().max().item())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
with torch.no_grad():
    for _ in range(2): m(xp, xv)
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(5): m(xp, xv)
    ev[1].record(); torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3 / 5
print("fp32 kernel: %.1f us  %.1f TFLOP/s  (%.3f of 157.3)" % (us, 262144 * 1186816 / us / 1e6, 262144 * 1186816 / us / 1e6 / 157.3))
