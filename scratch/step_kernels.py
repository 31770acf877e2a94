"""Kernel-by-kernel listing of one eager training step (torch profiler)."""
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch
from torch.profiler import profile, ProfilerActivity
from util import O, amd_model
import nerf_amd as amd
from nerf_amd.optim import FlatAdam
from nerf_amd.train import TrainStep
DEV='cuda'
mc, mf = amd_model(O.init_params(0), 'bf16'), amd_model(O.init_params(1), 'bf16')
pe, pv = amd.PositionalEncoder(3,10).to(DEV), amd.PositionalEncoder(3,4).to(DEV)
opt = FlatAdam(list(mc.parameters())+list(mf.parameters()))
N=1024
ro, rd = O.synthetic_rays(N, seed=0)
ro, rd = ro.to(DEV), rd.to(DEV)
tgt = torch.rand(N,3,device=DEV)
step = TrainStep(mc, mf, pe, pv, opt, N, use_graph=False)
for _ in range(3): step(ro, rd, tgt, lr=5e-4)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(ro, rd, tgt, lr=5e-4)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
tot = 0
for e in evs:
    print(f"{e.time_range.elapsed_us():9.1f} us  {e.name[:110]}")
    tot += e.time_range.elapsed_us()
print("kernels", len(evs), "sum us", tot)
