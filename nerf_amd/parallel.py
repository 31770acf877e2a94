"""One-process-per-GPU helpers: ray-sharded rendering and data-parallel training.

Rays are independent (nothing in models/nerf.py:318-483 couples two rays), so a frame is cut
into contiguous blocks of image rows, one per rank, each rank generates only its own rays, and the
only collective of a render is one all_gather of the [rows, W, 3] colour blocks.  Training is
data-parallel over rays: every rank draws its own batch, and the only collective of a step is one
summing all-reduce of the flat gradient bucket (4.77 MB for both networks), followed by an Adam
step that divides by the world size.  The backend is "nccl" (RCCL over xGMI) on GPUs and "gloo"
on CPUs; the helpers themselves only use torch.distributed calls and work with either.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the process group
    when WORLD_SIZE > 1 and selects this rank's GPU."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world)
    return rank, world, local_rank


def shard_rows(H: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of image rows of this rank: (row0, nrows); blocks differ by at most one row."""
    base, rem = divmod(H, world)
    row0 = rank * base + min(rank, rem)
    return row0, base + (1 if rank < rem else 0)


def shard_items(n: int, rank: int, world: int) -> range:
    """Round-robin assignment of n independent items (frames of a camera path) to ranks."""
    return range(rank, n, world)


def gather_rows(block: torch.Tensor, H: int, rank: int, world: int) -> torch.Tensor:
    """all_gather of per-rank row blocks [nrows_r, W, C] into the full [H, W, C] frame (on every rank).
    A lone rank has nothing to gather -- unless it runs inside a process group of size one, where the collective is
    issued all the same (that is how a one-GPU machine exercises this path on RCCL)."""
    if world == 1 and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() == 1):
        return block
    max_rows = shard_rows(H, 0, world)[1]
    pad = torch.zeros((max_rows,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    pad[: block.shape[0]] = block
    staged = _host_staged(pad)
    src = pad.cpu() if staged else pad
    out = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(out, src)
    frame = torch.cat([out[r][: shard_rows(H, r, world)[1]] for r in range(world)], dim=0)
    return frame.to(block.device) if staged else frame


def render_frame_sharded(render_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                         get_rays_fn: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor]],
                         H: int, W: int, rank: int, world: int, chunk: int = 65536) -> torch.Tensor:
    """Render one frame across ranks.  ``get_rays_fn(row0, nrows)`` returns this rank's rays
    ([nrows, W, 3] each); ``render_fn(rays_o, rays_d)`` maps flat rays [n,3] to colours [n,3].
    Returns the full [H, W, 3] frame on every rank (the chunk loop of scripts/render_nerf.py:270-305,
    applied to this rank's rows)."""
    row0, nrows = shard_rows(H, rank, world)
    ro, rd = get_rays_fn(row0, nrows)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    outs: List[torch.Tensor] = []
    for s in range(0, ro.shape[0], chunk):
        outs.append(render_fn(ro[s:s + chunk], rd[s:s + chunk]))
    block = torch.cat(outs, 0).reshape(nrows, W, 3) if outs else ro.new_zeros((0, W, 3))
    return gather_rows(block, H, rank, world)


def _host_staged(t: torch.Tensor) -> bool:
    """gloo moves device tensors through pinned host copies on streams of its own; with several ranks
    sharing one GPU (the only way to rehearse N > 1 on a one-GPU machine) that path stalls for seconds to
    minutes, for any program.  Under gloo, device tensors are therefore staged through the host here."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_(t: torch.Tensor, op=None) -> torch.Tensor:
    """In-place all-reduce of ``t`` (sum by default) on the default group."""
    op = dist.ReduceOp.SUM if op is None else op
    if _host_staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


class _Done:
    def wait(self):
        return True


def all_reduce_async_(t: torch.Tensor):
    """Start a summing all-reduce of ``t`` in place and return a handle whose ``wait()`` orders the current
    stream after it.  Under nccl (RCCL) the collective runs on the communicator's own stream, behind the
    work already queued on the current stream, so kernels launched next overlap with it."""
    if _host_staged(t):
        all_reduce_(t)
        return _Done()
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_gradients_(flat_grad: torch.Tensor, world: int) -> float:
    """Summing all-reduce of the gradient bucket in place; returns the factor the optimizer must
    apply (1/world) so that the update equals that of the global batch (equal shard sizes)."""
    if world > 1:
        all_reduce_(flat_grad)
    return 1.0 / world


def broadcast_parameters_(flat_params: torch.Tensor, world: int) -> None:
    """Make every rank start from rank 0's weights."""
    if world > 1:
        if _host_staged(flat_params):
            h = flat_params.cpu()
            dist.broadcast(h, src=0)
            flat_params.copy_(h)
        else:
            dist.broadcast(flat_params, src=0)
