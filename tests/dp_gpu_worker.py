"""One rank of the two-rank data-parallel training check of test_gpu_system.py (launched by
torch.distributed.run; both ranks share GPU 0 and talk over gloo).

argv: out_dir mode [backend], mode = "same" (both ranks hold the same rays and targets) or "split" (rank 1's
targets differ); backend "gloo" (default: both ranks share GPU 0) or "nccl" (RCCL, one GPU per rank).  Each
rank saves its flat weights and per-step losses after three graph-replayed steps, and under nccl also a
row-sharded, all-gathered frame.  Mode "solo" is one rank in a process group of its own (RANK=0, WORLD_SIZE=1 in
the environment) that runs the data-parallel step -- three graphs, two asynchronous all-reduces -- and the frame's
all_gather on that group: the way to run the collective code on RCCL with one GPU.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import O, amd_model  # noqa: E402


def run_steps(world, rank, mode, n=256, steps=3, step_world=None):
    """`step_world`: the world size the step is built for when it differs from the group's (the split step on a
    lone rank); the gradient scale is then set back to the group's 1 / world."""
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
    # the single-process reference keeps the per-network launches of the data-parallel step: merged launches
    # cut the sample axis differently and agree only to rounding
    step = TrainStep(mc, mf, pe, pv, opt, n, world_size=step_world or world, use_graph=True, joint=False)
    if step_world is not None:
        opt.device_state()[4] = 1.0 / world
    step.prepare()
    torch.manual_seed(123)                  # the same jitter draws on every rank and in the single-process run
    losses = [step(ro, rd, tgt, lr=5e-4).item() for _ in range(steps)]
    torch.cuda.synchronize()
    return opt.flat.detach().cpu(), losses


def render_frame(rank, world, H=96, W=64):
    """one frame, rows sharded over the ranks, gathered on every rank"""
    import nerf_amd
    from nerf_amd.render import render_image
    from nerf_amd.utils.synthetic import focal_from_fov, pose_spherical
    dev = "cuda"
    mc, mf = amd_model(O.init_params(0), "fp16"), amd_model(O.init_params(1), "fp16")
    pe, pv = nerf_amd.PositionalEncoder(3, 10).to(dev), nerf_amd.PositionalEncoder(3, 4).to(dev)
    return render_image(mc, mf, pe, pv, H, W, focal_from_fov(W), pose_spherical(25.0, -30.0, 4.0).to(dev), chunk=1500,
                        rank=rank, world_size=world)


if __name__ == "__main__":
    out_dir, mode = sys.argv[1], sys.argv[2]
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    from nerf_amd.parallel import init_distributed
    rank, world, _ = init_distributed(backend)
    if mode == "solo":
        import torch.distributed as dist
        assert world == 1 and not dist.is_initialized()
        dist.init_process_group(backend, rank=0, world_size=1)
        flat, losses = run_steps(1, 0, "same", step_world=2)
    else:
        flat, losses = run_steps(world, rank, mode)
    out = {"flat": flat, "losses": losses}
    if backend == "nccl":
        out["frame"] = render_frame(rank, world).cpu()
    torch.save(out, os.path.join(out_dir, f"{mode}_rank{rank}.pt"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
