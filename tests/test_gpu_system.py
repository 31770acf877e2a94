"""GPU: the pieces around the kernels -- the flat Adam optimizer, the HIP-graph training step, frame
rendering with row sharding."""
import numpy as np
import pytest
import torch

from util import O, T, amd_model, maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(precision="bf16"):
    import nerf_amd
    from nerf_amd.optim import FlatAdam
    mc, mf = amd_model(O.init_params(0), precision), amd_model(O.init_params(1), precision)
    pe, pv = nerf_amd.PositionalEncoder(3, 10).to(DEV), nerf_amd.PositionalEncoder(3, 4).to(DEV)
    opt = FlatAdam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4)
    return mc, mf, pe, pv, opt


def test_flat_adam_matches_torch_adam():
    from nerf_amd.optim import FlatAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in ((17, 5), (33,), (4, 4, 2))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, opt_ref = FlatAdam(ps, lr=1e-2), torch.optim.Adam(ref, lr=1e-2)
    for step in range(4):
        for p, r in zip(ps, ref):
            g = torch.randn_like(p)
            p.grad, r.grad = g.clone(), g.clone()
        lr = 1e-2 * 0.9 ** step
        for o in (opt, opt_ref):
            for pg in o.param_groups:           # the reference rewrites lr every step (train_nerf.py:289-290)
                pg["lr"] = lr
        opt.step(); opt_ref.step()
        for p, r in zip(ps, ref):
            assert maxabs(p, r) < 1e-5
    assert ps[0].data_ptr() == opt.flat.data_ptr()          # parameters are views of the bucket
    # device-state variant (graph-safe) takes the same steps
    ps2 = [torch.nn.Parameter(torch.randn(50, device=DEV))]
    ref2 = [torch.nn.Parameter(ps2[0].detach().clone())]
    o2, r2 = FlatAdam(ps2, lr=3e-3), torch.optim.Adam(ref2, lr=3e-3)
    for _ in range(3):
        g = torch.randn(50, device=DEV)
        ps2[0].grad, ref2[0].grad = g.clone(), g.clone()
        o2.step_dev(); r2.step()
    assert maxabs(ps2[0], ref2[0]) < 1e-5


def test_train_step_graph_and_eager():
    """The HIP-graph replay and the eager issue of the same step: both train (loss falls on a fixed batch),
    the learning rate written between replays is honoured, and lr = 0 leaves the weights untouched."""
    from nerf_amd.train import TrainStep
    from nerf_amd.utils.synthetic import synthetic_rays
    n = 256
    ro, rd = synthetic_rays(n, DEV, seed=0)
    tgt = torch.rand(n, 3, generator=torch.Generator().manual_seed(0)).to(DEV)
    losses = {}
    for use_graph in (True, False):
        mc, mf, pe, pv, opt = _setup()
        step = TrainStep(mc, mf, pe, pv, opt, n, use_graph=use_graph)
        w0 = opt.flat.clone()
        l0 = step(ro, rd, tgt, lr=0.0).item()
        torch.cuda.synchronize()
        assert torch.equal(opt.flat, w0), "lr = 0 must not move the weights"
        ls = [step(ro, rd, tgt, lr=5e-4).item() for _ in range(30)]
        assert not torch.equal(opt.flat, w0)
        assert np.isfinite(ls).all() and np.mean(ls[-5:]) < 0.8 * l0, (l0, ls[-5:])
        assert abs(float(opt.device_state()[5].item()) - 31.0) < 0.5      # device-side step counter
        losses[use_graph] = (l0, np.mean(ls[-5:]))
    # same weights, same rays, different jitter draws: the first losses agree to the jitter noise
    assert abs(losses[True][0] - losses[False][0]) < 0.02 * losses[False][0]


def test_render_image_and_row_sharding():
    import nerf_amd
    from nerf_amd.render import render_image, to_uint8
    from nerf_amd.utils.synthetic import focal_from_fov, pose_spherical
    mc, mf, pe, pv, _ = _setup("fp16")
    H = W = 40
    focal = focal_from_fov(W)
    c2w = pose_spherical(30.0, -30.0, 4.0).to(DEV)
    full = render_image(mc, mf, pe, pv, H, W, focal, c2w, chunk=600, white_bkgd=True, lindisp=True)
    assert full.shape == (H, W, 3) and torch.isfinite(full).all()
    # two "ranks" rendering their row blocks give the same pixels as the single-rank frame
    parts = [render_image(mc, mf, pe, pv, H, W, focal, c2w, chunk=333, white_bkgd=True, lindisp=True, rank=r,
                          world_size=3, gather=False) for r in range(3)]
    assert torch.equal(torch.cat(parts, 0), full)
    # against the fp32 oracle at 1e-3 on rays with a stable last-sample sign
    ro, rd = O.get_rays(H, W, focal, c2w.cpu())
    with torch.no_grad():
        oc, of = O.render_hierarchical(O.init_params(0), O.init_params(1), ro.reshape(-1, 3), rd.reshape(-1, 3), 2.0, 6.0,
                                       64, 128, det=True, lindisp=True, white_bkgd=True)
    ok = (oc["raw_sigma"][:, -1].abs() > 4e-4) & (of["raw_sigma"][:, -1].abs() > 4e-4)
    assert ok.float().mean() > 0.9
    assert maxabs(full.reshape(-1, 3).cpu()[ok], of["rgb_map"][ok]) < 1e-3
    u8 = to_uint8(torch.tensor([[-0.2, 0.5, 0.999, 1.0, 1.7]], device=DEV))
    assert u8.cpu().tolist() == [[0, 127, 254, 255, 255]]        # truncation like the reference (0.5 -> 127)
    coarse_only = render_image(mc, None, pe, pv, H, W, focal, c2w, chunk=4096)
    assert coarse_only.shape == (H, W, 3)


def test_render_path_and_evaluate(tmp_path):
    """camera-path frame loop (render_nerf.py:254-314) and the PSNR evaluation loop
    (evaluate_nerf.py:201-270) on a tiny scene: frames dealt to two ranks are disjoint and complete, the
    PNG files decode to the returned frames, PSNR equals -10 log10(mse) of a render against the loader's
    target image."""
    import math
    import nerf_amd
    from nerf_amd.datasets import BlenderDataset
    from nerf_amd.render import evaluate, render_image, render_path, to_uint8
    from nerf_amd.utils.png import read_png
    from nerf_amd.utils.synthetic import camera_path
    from util import make_blender_scene
    mc, mf = amd_model(O.init_params(0), "fp16"), amd_model(O.init_params(1), "fp16")
    pe, pv = nerf_amd.PositionalEncoder(3, 10).to(DEV), nerf_amd.PositionalEncoder(3, 4).to(DEV)
    poses = camera_path(5).to(DEV)
    kw = dict(white_bkgd=True, lindisp=True, chunk=100)
    f0 = render_path(mc, mf, pe, pv, 12, 16, 20.0, poses, rank=0, world_size=2, output_dir=str(tmp_path / "f"), **kw)
    f1 = render_path(mc, mf, pe, pv, 12, 16, 20.0, poses, rank=1, world_size=2, output_dir=str(tmp_path / "f"), **kw)
    assert sorted(f0) == [0, 2, 4] and sorted(f1) == [1, 3]
    whole = render_path(mc, mf, pe, pv, 12, 16, 20.0, poses, **kw)
    for i, fr in {**f0, **f1}.items():
        assert fr.dtype == np.uint8 and fr.shape == (12, 16, 3)
        assert np.array_equal(fr, whole[i])
        assert np.array_equal(read_png(str(tmp_path / "f" / f"frame_{i:04d}.png")), fr)
    ref = to_uint8(render_image(mc, mf, pe, pv, 12, 16, 20.0, poses[3], **kw)).cpu().numpy()
    assert np.array_equal(ref, whole[3])

    root = str(tmp_path / "scene")
    make_blender_scene(root, n_images=3, H=8, W=8, channels=4, splits=("val",))
    ds = BlenderDataset(root, "val", data_white_bkgd=True)
    ev = evaluate(mc, mf, pe, pv, ds, white_bkgd=True)
    assert sorted(ev["psnr"]) == [0, 1, 2]
    img = render_image(mc, mf, pe, pv, 8, 8, ds.focal, ds.poses[1], white_bkgd=True)
    mse = torch.mean((img - ds.image_rgb(1)) ** 2).item()
    assert abs(ev["mse"][1] - mse) < 1e-7 and abs(ev["psnr"][1] + 10 * math.log10(mse)) < 1e-4
    part = evaluate(mc, mf, pe, pv, ds, rank=1, world_size=2, white_bkgd=True)
    assert sorted(part["psnr"]) == [1] and part["psnr"][1] == ev["psnr"][1]


def test_two_rank_training_step_on_one_gpu(tmp_path):
    """Data-parallel TrainStep with world_size 2 (two processes sharing this GPU, gloo): with the same rays,
    targets and jitter on both ranks the summed-and-halved gradient is the single-process gradient bit for
    bit, so the weights after three steps are identical to a single-process run; with different targets on
    rank 1 both ranks still end with identical weights, which differ from the single-process ones."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from dp_gpu_worker import run_steps
    single, single_losses = run_steps(1, 0, "same")
    import socket
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for mode in ("same", "split"):
        with socket.socket() as sk:                      # a port that is free right now
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(here, "dp_gpu_worker.py"), str(tmp_path), mode]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=100)
        assert r.returncode == 0, r.stderr[-2000:]
        a = torch.load(tmp_path / f"{mode}_rank0.pt", weights_only=True)
        b = torch.load(tmp_path / f"{mode}_rank1.pt", weights_only=True)
        assert torch.equal(a["flat"], b["flat"]), "ranks diverged"
        if mode == "same":
            assert torch.equal(a["flat"], single) and a["losses"] == single_losses
        else:
            assert not torch.equal(a["flat"], single)
            assert a["losses"][0] == single_losses[0] and b["losses"][0] != single_losses[0]


@pytest.mark.parametrize("hier", [False, True])
@pytest.mark.parametrize("views", [False, True])
def test_training_step_every_flag_combination(hier, views):
    """The four runs train_nerf.py's flags allow (--use_hierarchical x --use_viewdirs, :62, :79, :293-331): the step
    replayed from HIP graphs with the per-network launches equals the eagerly issued step bit for bit over six steps,
    the merged launches of the single-process step agree to rounding, and the loss falls."""
    import nerf_amd
    from nerf_amd.optim import FlatAdam
    from nerf_amd.train import TrainStep
    from nerf_amd.utils.synthetic import seed_parameters_, synthetic_rays
    n = 512
    ro, rd = synthetic_rays(n, DEV, seed=0)
    tgt = torch.rand(n, 3, generator=torch.Generator().manual_seed(0)).to(DEV)
    res = {}
    for mode in ("graph", "eager", "graph_per_network"):
        mc = seed_parameters_(nerf_amd.NeRF(use_viewdirs=views).to(DEV), 0)
        mf = seed_parameters_(nerf_amd.NeRF(use_viewdirs=views).to(DEV), 1) if hier else None
        pe = nerf_amd.PositionalEncoder(3, 10).to(DEV)
        pv = nerf_amd.PositionalEncoder(3, 4).to(DEV) if views else None
        opt = FlatAdam(list(mc.parameters()) + (list(mf.parameters()) if hier else []), lr=5e-4)
        step = TrainStep(mc, mf, pe, pv, opt, n, use_graph=mode != "eager", external_uniforms=True, joint=mode == "graph")
        step.uniforms.copy_(torch.rand(step.n_uniforms, generator=torch.Generator().manual_seed(5)).to(DEV))
        res[mode] = ([step(ro, rd, tgt, lr=5e-4).item() for _ in range(6)], opt.flat.clone())
    (lg, _), (le, we), (ln, wn) = res["graph"], res["eager"], res["graph_per_network"]
    assert ln == le and torch.equal(wn, we)
    assert all(abs(a - b) < 2e-3 * abs(b) for a, b in zip(lg, le)), (lg, le)
    assert all(np.isfinite(le)) and le[-1] < le[0]


def test_rccl_collectives_with_one_rank(tmp_path):
    """The data-parallel step and the sharded render on RCCL itself (backend "nccl"), with the one rank a one-GPU
    machine allows: process-group set-up, capture of the step's three graphs while the communicator's watchdog thread
    is alive, the two asynchronous all-reduces of the gradient halves between graph replays, Adam on the reduced
    bucket, and the all_gather of a frame's row blocks on device tensors.  A lone rank's sum is its own gradient, so
    the weights after three steps equal the single-process step's bit for bit, and the gathered frame the plain one."""
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from dp_gpu_worker import render_frame, run_steps
    single, single_losses = run_steps(1, 0, "same")
    frame = render_frame(0, 1).cpu()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(here, "dp_gpu_worker.py"), str(tmp_path), "solo", "nccl"], env=env,
                       capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-3000:]
    a = torch.load(tmp_path / "solo_rank0.pt", weights_only=True)
    assert torch.equal(a["flat"], single) and a["losses"] == single_losses
    assert torch.equal(a["frame"], frame)


def test_runs_are_bit_reproducible():
    """The kernels order their LDS rings and loads with counted waits; a missed wait would show up as run-to-run
    differences.  The same seeded 300-step training run twice, and the same frame at two chunk sizes, agree
    bit for bit (the long version, 3000 steps and 30 frames per precision, is scratch/soak.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dp_gpu_worker import run_steps
    a, la = run_steps(1, 0, "same", n=1024, steps=300)
    b, lb = run_steps(1, 0, "same", n=1024, steps=300)
    assert torch.equal(a, b) and la == lb and bool(torch.isfinite(a).all())
    from nerf_amd.render import render_image
    from nerf_amd.utils.synthetic import focal_from_fov, pose_spherical
    mc, mf, pe, pv, _ = _setup("fp16")
    c2w = pose_spherical(40.0, -30.0, 4.0).to(DEV)
    frames = [render_image(mc, mf, pe, pv, 200, 200, focal_from_fov(200), c2w, chunk=c) for c in (65536, 4096, 1000, 65536)]
    assert all(torch.equal(frames[0], f) for f in frames[1:])
