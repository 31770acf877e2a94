"""CPU, world_size 2 over gloo: the multi-process logic of nerf_amd.parallel -- row sharding and the
frame gather of a ray-sharded render, and the equivalence of data-parallel gradient averaging with the
global-batch gradient.  The GPU kernels are not available here, so the per-rank compute is the CPU
oracle; what is under test is the partitioning and the collectives."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import O, ROOT


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    from nerf_amd.parallel import init_distributed
    return init_distributed("gloo")


def _render_worker(rank, world, port, H, W, out):
    from nerf_amd.parallel import render_frame_sharded, shard_rows
    _init(rank, world, port)
    c2w = O.pose_spherical(40.0, -30.0, 4.0)
    focal = 0.5 * W / 0.36
    pc = O.init_params(0)

    def get_rays_block(row0, nrows):
        ro, rd = O.get_rays(H, W, focal, c2w)
        return ro[row0:row0 + nrows], rd[row0:row0 + nrows]

    def render(ro, rd):
        with torch.no_grad():
            return O.render_rays(pc, ro, rd, 2.0, 6.0, 8)["rgb_map"]

    frame = render_frame_sharded(render, get_rays_block, H, W, rank, world, chunk=37)
    if rank == 0:
        torch.save(frame, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [8, 9])
def test_sharded_render_equals_single(tmp_path, H):
    W, world = 6, 2
    out = str(tmp_path / "frame.pt")
    mp.spawn(_render_worker, args=(world, 29611 + H, H, W, out), nprocs=world, join=True)
    frame = torch.load(out)
    c2w = O.pose_spherical(40.0, -30.0, 4.0)
    ro, rd = O.get_rays(H, W, 0.5 * W / 0.36, c2w)
    with torch.no_grad():
        ref = O.render_rays(O.init_params(0), ro.reshape(-1, 3), rd.reshape(-1, 3), 2.0, 6.0, 8)["rgb_map"]
    assert frame.shape == (H, W, 3)
    assert torch.allclose(frame.reshape(-1, 3), ref, atol=1e-6)


def test_shard_rows_partition():
    from nerf_amd.parallel import shard_items, shard_rows
    for H in (1, 7, 8, 800, 801):
        for world in (1, 2, 3, 8):
            blocks = [shard_rows(H, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(n for _, n in blocks) == H
            assert all(blocks[i][0] + blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1
    assert sorted(sum((list(shard_items(120, r, 8)) for r in range(8)), [])) == list(range(120))


def _dp_worker(rank, world, port, out):
    from nerf_amd.parallel import allreduce_gradients_, broadcast_parameters_
    _init(rank, world, port)
    n = 16
    ro, rd = O.synthetic_rays(2 * n, seed=0, H=32, W=32)
    gen = torch.Generator().manual_seed(1)
    tgt = torch.rand(2 * n, 3, generator=gen)
    t1, uu, t2 = torch.rand(2 * n, 8, generator=gen), torch.rand(2 * n, 8, generator=gen), torch.rand(2 * n, 16, generator=gen)
    sl = slice(rank * n, (rank + 1) * n)
    pc = {k: v.requires_grad_(True) for k, v in O.init_params(0).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.init_params(1 + rank).items()}     # ranks start different ...
    flat_p = torch.cat([v.detach().reshape(-1) for v in pf.values()])
    broadcast_parameters_(flat_p, world)                                             # ... and adopt rank 0's
    off = 0
    for v in pf.values():
        v.data.copy_(flat_p[off:off + v.numel()].view_as(v)); off += v.numel()
    tot, *_ = O.train_step_loss(pc, pf, ro[sl], rd[sl], tgt[sl], 2.0, 6.0, 8, 8, t1[sl], uu[sl], t2[sl])
    tot.backward()
    flat = torch.cat([v.grad.reshape(-1) for v in list(pc.values()) + list(pf.values())])
    scale = allreduce_gradients_(flat, world)
    if rank == 0:
        torch.save(flat * scale, out)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_equals_global_batch(tmp_path):
    world, n = 2, 16
    out = str(tmp_path / "grad.pt")
    mp.spawn(_dp_worker, args=(world, 29631, out), nprocs=world, join=True)
    got = torch.load(out)
    ro, rd = O.synthetic_rays(2 * n, seed=0, H=32, W=32)
    gen = torch.Generator().manual_seed(1)
    tgt = torch.rand(2 * n, 3, generator=gen)
    t1, uu, t2 = torch.rand(2 * n, 8, generator=gen), torch.rand(2 * n, 8, generator=gen), torch.rand(2 * n, 16, generator=gen)
    pc = {k: v.requires_grad_(True) for k, v in O.init_params(0).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.init_params(1).items()}
    tot, *_ = O.train_step_loss(pc, pf, ro, rd, tgt, 2.0, 6.0, 8, 8, t1, uu, t2)
    tot.backward()
    ref = torch.cat([v.grad.reshape(-1) for v in list(pc.values()) + list(pf.values())])
    assert got.shape == ref.shape == (2 * 595844,)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-7 * ref.abs().max().item() + 1e-9)
