"""Rate of the 16-bit layer-program kernels on the 8x256 network, 262144 samples (HIP events around the C calls)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import ops
from nerf_amd.models.nerf import NeRF, _ProgramForward
from nerf_amd.utils.synthetic import seed_parameters_
dev = "cuda"
M = 262144
xp, xv, cot = torch.randn(M, 63, device=dev), torch.randn(M, 27, device=dev), torch.randn(M, 4, device=dev)
for prec in ("bf16", "fp32"):
    mp = seed_parameters_(NeRF(precision=prec).to(dev), 0)
    ops.KERNEL_TIMES = {}
    for _ in range(4):
        with torch.no_grad():
            _ProgramForward.apply(mp, xp, xv, False, prec, *mp.ordered_params())
        with torch.enable_grad():
            mp.zero_grad(set_to_none=True)
            torch.autograd.backward([_ProgramForward.apply(mp, xp, xv, True, prec, *mp.ordered_params())], [cot])
    torch.cuda.synchronize()
    evs, ops.KERNEL_TIMES = ops.KERNEL_TIMES, None
    print(prec, {k: round(sum(a.elapsed_time(b) * 1e3 for a, b, _ in v[1:]) / (len(v) - 1), 1) for k, v in evs.items()})
