"""One rank of the two-rank data-parallel training check of test_gpu_system.py (launched by
torch.distributed.run; both ranks share GPU 0 and talk over gloo).

argv: out_dir mode, mode = "same" (both ranks hold the same rays and targets) or "split" (rank 1's
targets differ).  Each rank saves its flat weights and per-step losses after three graph-replayed steps.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import O, amd_model  # noqa: E402


def run_steps(world, rank, mode, n=256, steps=3):
    import nerf_amd
    from nerf_amd.optim import FlatAdam
    from nerf_amd.train import TrainStep
    from nerf_amd.utils.synthetic import synthetic_rays
    dev = "cuda"
    mc, mf = amd_model(O.init_params(0), "bf16"), amd_model(O.init_params(1), "bf16")
    pe, pv = nerf_amd.PositionalEncoder(3, 10).to(dev), nerf_amd.PositionalEncoder(3, 4).to(dev)
    opt = FlatAdam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4)
    ro, rd = synthetic_rays(n, dev, seed=0)
    tgt = torch.rand(n, 3, generator=torch.Generator().manual_seed(7 if (mode == "split" and rank == 1) else 0)).to(dev)
    step = TrainStep(mc, mf, pe, pv, opt, n, world_size=world, use_graph=True)
    step.prepare()
    torch.manual_seed(123)                  # the same jitter draws on every rank and in the single-process run
    losses = [step(ro, rd, tgt, lr=5e-4).item() for _ in range(steps)]
    torch.cuda.synchronize()
    return opt.flat.detach().cpu(), losses


if __name__ == "__main__":
    out_dir, mode = sys.argv[1], sys.argv[2]
    from nerf_amd.parallel import init_distributed
    rank, world, _ = init_distributed("gloo")
    flat, losses = run_steps(world, rank, mode)
    torch.save({"flat": flat, "losses": losses}, os.path.join(out_dir, f"{mode}_rank{rank}.pt"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
