#!/usr/bin/env python3
"""Benchmark of the NeRF hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (the single-GPU training configuration of the benchmark set): one training step of scripts/train_nerf.py:288-336 with its
intended flat ray batch -- 1024 rays per GPU, 64 coarse + 128 fine samples, stratified jitter,
stochastic hierarchical sampling, both 8x256 networks, loss, backward, Adam.  With N > 1 every rank
runs that step on its own rays (weak scaling) and the gradients are summed with one RCCL
all-reduce of the flat 4.77 MB bucket before the optimizer step.  Inputs are synthetic (seeded
Blender-style camera, random-init weights, uniform targets) and resident in HBM.

Rank 0 prints one JSON line.  `value` = rays processed by all ranks / wall time (max over ranks) of the
median of --repeats timed regions of exactly --steps steps each.  The line also carries the same step at
4096 rays per GPU (the data-parallel configuration's batch) and, for N > 1, a row-sharded 800x800 frame render
with its all_gather and a frame-sharded slice of the 120-pose camera path.
The timed steps replay the step from HIP graphs (a 1024-ray step is 25 short launches; issued from
Python the host would be the bottleneck).  `roofline` describes the kernel with the largest share of
GPU time in the step, priced against the dense MFMA peak; its duration comes from one-lane time-stamp kernels
captured into a second copy of the step's graph around every C-ABI call and replayed right after the timed region.  `cpu_baseline` times the CPU oracle
(oracle/nerf_oracle.py, a restatement of the reference's PyTorch path) on the same workload on
this host's cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MLP_FWD_FLOP = 1186816          # 2 * 593 408 MAC per sample, unpadded dims
MLP_DGRAD_FLOP = 2 * 557696     # backward-data MACs per sample (view/point-encoding inputs need none)
MLP_WGRAD_FLOP = 2 * 593408
MLP_FLOP_PER_RAY_TRAIN = 256 * (MLP_FWD_FLOP + MLP_DGRAD_FLOP + MLP_WGRAD_FLOP)   # 64 + 192 evaluations: 893.2 MFLOP
MLP_FLOP_PER_RAY_RENDER = 256 * MLP_FWD_FLOP                                       # 303.8 MFLOP
MFMA_PEAK_TFLOPS = 2500.0       # dense bf16/fp16 peak, MI355X
HBM_PEAK_GBS = 8000.0
WGRAD_BYTES_PER_SAMPLE = 280 * 1024 // 32   # saved activations (142 KiB) + dZ (138 KiB) per 32 samples, each read once
# HBM bytes per launch measured with rocprofv3 PMC passes of this same command (FETCH_SIZE doubled as
# gfx950 requires for wide streams, plus WRITE_SIZE), written by scratch/summarize_prof.py
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")


def measured_traffic(kernel, rays_per_gpu, precision):
    """HBM bytes per launch of `kernel` from the committed PMC summary, or None when the summary was
    taken for another workload."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if t.get("rays_per_gpu") != rays_per_gpu or t.get("precision") != precision:
        return None
    e = t.get("kernels", {}).get(kernel)
    return None if e is None else e.get("hbm_bytes_per_launch")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays-per-gpu", type=int, default=1024)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--no-render", action="store_true", help="skip the 800x800 frame render measurement")
    ap.add_argument("--render-chunk", type=int, nargs="+", default=[65536, 4096],
                    help="rays per launch group of the frame render; the first value is the headline one, "
                         "4096 is the reference's eval_batch_size (render_nerf.py:55)")
    ap.add_argument("--no-one-eighth", action="store_true",
                    help="skip the 32-of-256-CU run (the metric's '1/8 GPU' reading), a child process under HSA_CU_MASK")
    ap.add_argument("--dist-backend", default=None,
                    help="torch.distributed backend (default: nccl = RCCL on GPUs); gloo lets several ranks share one GPU "
                         "to rehearse the N>1 code path")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch from Python instead of replaying HIP graphs")
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed region of --steps steps is run this many times; the median region is reported")
    ap.add_argument("--big-batch", type=int, default=4096,
                    help="rays per GPU of the second training measurement (the data-parallel configuration's batch); 0 = skip")
    ap.add_argument("--no-default-flags", action="store_true",
                    help="skip the extra leg with the reference script's default flags (coarse only, no view directions)")
    ap.add_argument("--no-joint", action="store_true",
                    help="keep the per-network pack and weight-gradient launches in the single-process step (A/B switch)")
    ap.add_argument("--kernel-steps", type=int, default=10, help="eager steps with per-kernel HIP events after the timed region")
    return ap.parse_args()


def build(device, precision):
    from nerf_amd.models.nerf import NeRF, PositionalEncoder
    from nerf_amd.optim import FlatAdam
    from nerf_amd.utils.synthetic import seed_parameters_
    mc = seed_parameters_(NeRF(precision=precision).to(device), 0)
    mf = seed_parameters_(NeRF(precision=precision).to(device), 1)
    pe, pv = PositionalEncoder(3, 10).to(device), PositionalEncoder(3, 4).to(device)
    opt = FlatAdam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4)
    return mc, mf, pe, pv, opt


def render_section(mc, mf, pe, pv, device, args):
    """The full-frame render configuration: one 800x800 frame (640 000 rays, 64 + 128 samples, deterministic, no_grad),
    rays generated on the GPU, rendered in chunks.  Reports rays/s and the achieved rates of the MLP
    forward (MFMA) and of the compositing / sampling / ray-generation kernels (HBM, algorithmic bytes)."""
    from nerf_amd import ops
    from nerf_amd.render import render_image
    from nerf_amd.utils.synthetic import focal_from_fov, pose_spherical
    H = W = 800
    c2w = pose_spherical(0.0, -30.0, 4.0).to(device)
    focal = focal_from_fov(W)
    out = []
    for chunk in args.render_chunk:
        render_image(mc, mf, pe, pv, H, W, focal, c2w, chunk=chunk)          # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames = 2
        for _ in range(frames):
            render_image(mc, mf, pe, pv, H, W, focal, c2w, chunk=chunk)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / frames
        ops.KERNEL_TIMES = {}
        render_image(mc, mf, pe, pv, H, W, focal, c2w, chunk=chunk)
        torch.cuda.synchronize()
        times, ops.KERNEL_TIMES = ops.KERNEL_TIMES, None
        sec = {"workload": "800x800 frame, 640000 rays, 64+128 samples, chunk %d, no_grad, deterministic" % chunk,
               "s_per_frame": round(dt, 4), "rays_per_s": round(H * W / dt, 1),
               # the whole render against the dense MFMA peak: 256 evaluations x 1 186 816 FLOP per ray
               "mlp_flop_frac_of_mfma_peak": round(H * W / dt * MLP_FLOP_PER_RAY_RENDER / (MFMA_PEAK_TFLOPS * 1e12), 4),
               "kernels": {}}
        for name, evs in sorted(times.items()):
            tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
            units = sum(u for _, _, u in evs)
            e = {"total_ms": round(tot_ms, 3), "launches": len(evs)}
            per_unit = None
            if name == "mlp_fwd":
                e["TFLOP/s"] = round(units * MLP_FWD_FLOP / (tot_ms * 1e-3) / 1e12, 1)
                e["frac_of_mfma_peak"] = round(e["TFLOP/s"] / MFMA_PEAK_TFLOPS, 4)
            elif name.startswith("composite_fwd"):
                per_unit = 28 * int(name.rsplit("S", 1)[1]) + 24       # raw 16S + z 4S in; weights, alpha 8S + maps out
            elif name == "get_rays":
                per_unit = 24
            elif name == "sample_merge_det":
                per_unit = 64 * 4 * 2 + 192 * 4                         # z, weights [64] in; sorted z [192] out
            if per_unit is not None:
                e["GB/s"] = round(units * per_unit / (tot_ms * 1e-3) / 1e9, 1)
                e["frac_of_hbm_peak"] = round(e["GB/s"] / HBM_PEAK_GBS, 4)
            sec["kernels"][name] = e
        out.append(sec)
    # frames of the 120-pose camera path with the white background and inverse-depth sampling
    # (render_nerf.py:145-149 with --white_bkgd --lindisp): render, clip, uint8 on the device, copy to the host
    from nerf_amd.render import render_path
    from nerf_amd.utils.synthetic import camera_path
    poses = camera_path(120).to(device)[::30]
    render_path(mc, mf, pe, pv, H, W, focal, poses[:1], chunk=args.render_chunk[0], lindisp=True, white_bkgd=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = render_path(mc, mf, pe, pv, H, W, focal, poses, chunk=args.render_chunk[0], lindisp=True, white_bkgd=True)
    dt = (time.perf_counter() - t0) / len(poses)
    out.append({"workload": "camera path: %d of the 120 poses, 800x800, 64+128 samples, lindisp, white background, chunk %d, "
                            "uint8 frames on the host" % (len(poses), args.render_chunk[0]),
                "s_per_frame": round(dt, 4), "frames_per_s": round(1.0 / dt, 2), "rays_per_s": round(H * W / dt, 1),
                "frames": len(frames), "kernels": {}})
    return out


def whole_frame_kernels(device):
    """The bandwidth-bound kernels of the render path launched once over a whole 800x800 frame (640 000 rays)
    instead of per chunk: with 1-3 GB per launch the launch ramp and the infinity cache no longer matter and the
    rate is the kernel's own HBM rate.  Algorithmic bytes as in render_section; HIP events on the launch stream."""
    from nerf_amd import ops
    n = 640000
    out = {}

    def rate(fn, nbytes, iters=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        gbs = nbytes / us / 1e3
        return {"avg_us": round(us, 1), "bytes_per_launch": nbytes, "GB/s": round(gbs, 1),
                "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}

    with torch.no_grad():
        for S in (64, 192):
            raw = torch.randn(n, S, 4, device=device)
            z = 2.0 + torch.cumsum(torch.rand(n, S, device=device), dim=-1) * (4.0 / S)     # increasing depths
            out["composite_fwd_S%d" % S] = rate(lambda: ops.composite(raw, z, False), n * (28 * S + 24))
            # the backward with the cotangents a training step has (d rgb_map; d weights is absent because the
            # fine sample positions are not differentiated): raw 16S + z 4S + g_rgb 12 in, d_raw 16S out
            g_rgb, d_raw = torch.randn(n, 3, device=device), torch.empty_like(raw)
            st = torch.cuda.current_stream().cuda_stream
            out["composite_bwd_S%d" % S] = rate(lambda: ops.check(ops.lib.nerf_composite_bwd(
                n, S, raw.data_ptr(), z.data_ptr(), 0, g_rgb.data_ptr(), None, None, None, None, None,
                d_raw.data_ptr(), st), "nerf_composite_bwd"), n * (36 * S + 12))
            del raw, z, g_rgb, d_raw
        # the fp32-operand validation kernel (v_mfma_f32_32x32x2_f32) on the encoded inputs of 1024 rays x 256 samples
        from nerf_amd.models.nerf import NeRF
        from nerf_amd.utils.synthetic import seed_parameters_
        m32 = seed_parameters_(NeRF(precision="fp32").to(device), 0)
        xp, xv = torch.randn(262144, 63, device=device), torch.randn(262144, 27, device=device)
        r32 = rate(lambda: m32(xp, xv), 0, iters=3)
        tf = 262144 * MLP_FWD_FLOP / r32["avg_us"] / 1e6
        out["mlp_fwd_fp32_262144_samples"] = {"avg_us": r32["avg_us"], "TFLOP/s": round(tf, 1), "peak_TFLOP/s": 157.3,
                                              "frac_of_fp32_mfma_peak": round(tf / 157.3, 4)}
        # the same network's fp32 training pass: forward that keeps activations + backward-data chain + weight gradients
        # (nerf_mlp_fp32_fwd_train, nerf_mlp_fp32_bwd), 3 489 024 FLOP per sample
        cot = torch.randn(262144, 4, device=device)

        def fwd_bwd():
            with torch.enable_grad():
                m32.zero_grad(set_to_none=True)
                torch.autograd.backward([m32(xp, xv)], [cot])
        r32b = rate(fwd_bwd, 0, iters=3)
        tfb = 262144 * (MLP_FWD_FLOP + MLP_DGRAD_FLOP + MLP_WGRAD_FLOP) / r32b["avg_us"] / 1e6
        out["mlp_fwd_bwd_fp32_262144_samples"] = {"avg_us": r32b["avg_us"], "TFLOP/s": round(tfb, 1), "peak_TFLOP/s": 157.3,
                                                  "frac_of_fp32_mfma_peak": round(tfb / 157.3, 4)}
        del m32
        # the 16-bit layer-program kernels (any architecture) on the same network and inputs, run through the program
        # path on purpose (the scripts' network itself goes to the fused chain kernels): durations between HIP events
        # around the C calls, without the Python-side weight packing
        from nerf_amd.models.nerf import _ProgramForward
        mp = seed_parameters_(NeRF(precision="bf16").to(device), 0)
        ops.KERNEL_TIMES = {}
        try:
            for _ in range(4):
                with torch.no_grad():
                    _ProgramForward.apply(mp, xp, xv, False, "bf16", *mp.ordered_params())
                with torch.enable_grad():
                    mp.zero_grad(set_to_none=True)
                    torch.autograd.backward([_ProgramForward.apply(mp, xp, xv, True, "bf16", *mp.ordered_params())], [cot])
            torch.cuda.synchronize()
            evs, ops.KERNEL_TIMES = ops.KERNEL_TIMES, None
        finally:
            ops.KERNEL_TIMES = None
        us = {k: sum(a.elapsed_time(b) * 1e3 for a, b, _ in v[1:]) / (len(v) - 1) for k, v in evs.items()}
        tfp = 262144 * MLP_FWD_FLOP / us["mlp_fwd_p16"] / 1e6
        out["mlp_fwd_p16_bf16_262144_samples"] = {"avg_us": round(us["mlp_fwd_p16"], 1), "TFLOP/s": round(tfp, 1),
                                                  "frac_of_mfma_peak": round(tfp / MFMA_PEAK_TFLOPS, 4)}
        tb = us["mlp_fwd_p16_train"] + us["mlp_bwd_p16"]
        tfpb = 262144 * (MLP_FWD_FLOP + MLP_DGRAD_FLOP + MLP_WGRAD_FLOP) / tb / 1e6
        out["mlp_fwd_bwd_p16_bf16_262144_samples"] = {"avg_us": round(tb, 1), "fwd_train_us": round(us["mlp_fwd_p16_train"], 1),
                                                      "bwd_us": round(us["mlp_bwd_p16"], 1), "TFLOP/s": round(tfpb, 1),
                                                      "frac_of_mfma_peak": round(tfpb / MFMA_PEAK_TFLOPS, 4)}
        del mp, xp, xv, cot
        c2w = torch.eye(4, device=device)[:3].contiguous()
        out["get_rays_800x800"] = rate(lambda: ops.get_rays(800, 800, 1111.0, c2w), n * 24, iters=20)
        # 983 MB written per launch: four times the 256 MB infinity cache, so this is the HBM rate
        out["get_rays_6400x6400"] = rate(lambda: ops.get_rays(6400, 6400, 8888.0, c2w), 64 * n * 24)
    return out


def one_eighth_gpu(args):
    """The same bench restricted to 32 of the 256 CUs (HSA_CU_MASK=0:0-31 in a child process, which must
    be set before the runtime starts): the headline metric is also quoted "at 1/8 GPU".  Returns a summary, or a
    note when the run failed."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-one-eighth", "--steps", "20",
           "--warmup", "5", "--rays-per-gpu", str(args.rays_per_gpu), "--precision", args.precision,
           "--render-chunk", str(args.render_chunk[0]), "--big-batch", "0", "--repeats", "1"]
    env = dict(os.environ, HSA_CU_MASK="0:0-31")
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001  (a failed side measurement must not fail the bench)
        return {"error": f"{type(e).__name__}: {e}"}
    out = {"cu_mask": "HSA_CU_MASK=0:0-31 (32 of 256 CUs)", "rays_per_s": d["value"], "ms_per_step": d["ms_per_step"],
           "peak_TFLOP/s": MFMA_PEAK_TFLOPS / 8}
    if d.get("render"):
        k = d["render"][0]["kernels"].get("mlp_fwd", {})
        out["render_s_per_frame"] = d["render"][0]["s_per_frame"]
        out["render_rays_per_s"] = d["render"][0]["rays_per_s"]
        if "TFLOP/s" in k:
            out["mlp_fwd_TFLOP/s"] = k["TFLOP/s"]
            out["mlp_fwd_frac_of_peak"] = round(k["TFLOP/s"] / (MFMA_PEAK_TFLOPS / 8), 4)
    return out


def cpu_baseline(n_rays, steps):
    """The CPU oracle's training step (fp32 torch ops, autograd, torch.optim.Adam) on this host."""
    from oracle import nerf_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = int(os.environ.get("OMP_NUM_THREADS", cores))
    torch.set_num_threads(threads)
    pc = {k: v.requires_grad_(True) for k, v in O.init_params(0).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.init_params(1).items()}
    opt = torch.optim.Adam(list(pc.values()) + list(pf.values()), lr=5e-4)
    ro, rd = O.synthetic_rays(n_rays, seed=0)
    gen = torch.Generator().manual_seed(0)
    tgt = torch.rand(n_rays, 3, generator=gen)

    def one():
        t1, uu, t2 = torch.rand(n_rays, 64), torch.rand(n_rays, 128), torch.rand(n_rays, 192)
        tot, *_ = O.train_step_loss(pc, pf, ro, rd, tgt, 2.0, 6.0, 64, 128, t1, uu, t2)
        opt.zero_grad()
        tot.backward()
        opt.step()
    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t0) / steps
    # the render path of the same oracle (no_grad, deterministic, both passes) on the same rays
    with torch.no_grad():
        pcd, pfd = {k: v.detach() for k, v in pc.items()}, {k: v.detach() for k, v in pf.items()}
        O.render_hierarchical(pcd, pfd, ro, rd, 2.0, 6.0, 64, 128, det=True)
        t1 = time.perf_counter()
        reps = max(1, steps // 2)
        for _ in range(reps):
            O.render_hierarchical(pcd, pfd, ro, rd, 2.0, 6.0, 64, 128, det=True)
        dt_render = (time.perf_counter() - t1) / reps
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": n_rays / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model, "torch_version": torch.__version__,
            "sample": f"{steps} training steps of {n_rays} rays (64+128 samples, fwd+bwd+Adam), fp32 torch CPU, "
                      f"{threads} threads, after 1 warm-up step",
            "render_rays_per_s": n_rays / dt_render,
            "render_sample": f"{reps} no_grad deterministic renders of the same {n_rays} rays (64+128 samples)"}


def stage(msg):
    """Progress marker on stderr when NERF_BENCH_TRACE is set (to localise a hang in a multi-process run)."""
    if os.environ.get("NERF_BENCH_TRACE"):
        print("[bench rank %s] %.2f %s" % (os.environ.get("RANK", "0"), time.time() % 1000, msg), file=sys.stderr, flush=True)


def train_leg(args, n, mc, mf, pe, pv, opt, device, rank, world, sync, max_over_ranks, kernel_steps, step0=0):
    """One training measurement at n rays per GPU: graphs captured, args.warmup untimed steps, then args.repeats
    timed regions of exactly args.steps steps, each bracketed by barrier + synchronize on both sides and taken as
    the maximum over ranks; the median region is the one reported.  With kernel_steps > 0 the same step is then
    issued eagerly with HIP events on the launch stream around every C-ABI call."""
    import torch.distributed as dist
    from nerf_amd import ops
    from nerf_amd.train import TrainStep
    from nerf_amd.utils.synthetic import synthetic_rays
    rays_o, rays_d = synthetic_rays(n, device, seed=rank)
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(1000 + rank)).to(device)
    mk = lambda graph: TrainStep(mc, mf, pe, pv, opt, n, near=2.0, far=6.0, N_samples_coarse=64, N_samples_fine=128,
                                 world_size=world, use_graph=graph, joint=not args.no_joint)
    step_fn = mk(not args.no_graph)
    # the batch sits in the step's own input buffers, where a loader's gather kernel writes it
    # (Trainer.train_step), so a step issues no input copies
    step_fn.rays_o.copy_(rays_o); step_fn.rays_d.copy_(rays_d); step_fn.target.copy_(target)
    rays_o, rays_d, target = step_fn.rays_o, step_fn.rays_d, step_fn.target

    # Capture the graphs before the first collective of a step.  If capture fails on any rank, every rank
    # falls back to issuing the launches from Python (slower, still the same work) and the line says so.
    graph_error = None
    if not args.no_graph:
        try:
            step_fn.prepare()
        except Exception as e:      # noqa: BLE001
            graph_error = "%s: %s" % (type(e).__name__, str(e)[:200])
        failed = torch.tensor([0.0 if graph_error is None else 1.0], device=device)
        if world > 1:
            from nerf_amd.parallel import all_reduce_
            all_reduce_(failed, op=dist.ReduceOp.MAX)
        if failed.item() > 0:
            graph_error = graph_error or "graph capture failed on another rank"
            print("bench.py: " + graph_error + "; continuing without graphs", file=sys.stderr, flush=True)
            step_fn = mk(False)
            step_fn.rays_o.copy_(rays_o); step_fn.rays_d.copy_(rays_d); step_fn.target.copy_(target)
            rays_o, rays_d, target = step_fn.rays_o, step_fn.rays_d, step_fn.target
    stage("graphs captured" if graph_error is None else "no graphs")

    def lr_at(i):      # scripts/train_nerf.py:288: lr0 * decay^(step / decay_steps)
        return 5e-4 * (0.1 ** (i / 250000.0))

    it = step0
    for _ in range(args.warmup):
        step_fn(rays_o, rays_d, target, lr=lr_at(it)); it += 1
    regions = []
    loss = None
    for _ in range(max(1, args.repeats)):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step_fn(rays_o, rays_d, target, lr=lr_at(it)); it += 1
        sync()
        regions.append(max_over_ranks(time.perf_counter() - t0))
        stage("timed region done")
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite loss")
    dt = sorted(regions)[len(regions) // 2]
    times, timing, eager = None, None, None
    if kernel_steps > 0:
        times, timing, eager = kernel_durations(mk, step_fn, kernel_steps, lr_at(it), sync, device)
        it += 3 * (kernel_steps + 1)
    return {"dt": dt, "regions_ms_per_step": [round(r / args.steps * 1e3, 4) for r in regions],
            "graph_error": graph_error, "times": times, "timing": timing, "eager_times": eager, "next_step": it}


class Stamper:
    """Slots of the device's 100 MHz counter written by one-lane kernels of libnerf_bench.so
    (nerf_amd/csrc/bench_stamp.hip); ``__call__(stream)`` enqueues a stamp and returns its slot."""

    def __init__(self, device, names=None, n_slots=4096):
        self.names = names          # only calls with these names are stamped (None: all)
        self.pair = None            # two stamps taken back to back: what a stamp itself adds to an interval
        import ctypes
        path = os.path.join(ROOT, "nerf_amd", "lib", "libnerf_bench.so")
        self.lib = ctypes.CDLL(path)
        self.lib.nerf_bench_stamp.restype = ctypes.c_int
        self.lib.nerf_bench_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.lib.nerf_bench_stamp_rate.restype = ctypes.c_int64
        self.rate = float(self.lib.nerf_bench_stamp_rate())
        self.slots = torch.zeros((n_slots,), dtype=torch.int64, device=device)
        self.used = 0

    def __call__(self, stream):
        if self.pair is None:
            self.pair = ()
            self.pair = (self(stream), self(stream))
            return self.pair[1]
        if self.used >= self.slots.numel():
            raise RuntimeError("out of time-stamp slots")
        i = self.used
        self.used += 1
        rc = self.lib.nerf_bench_stamp(self.slots.data_ptr() + 8 * i, stream.cuda_stream)
        if rc != 0:
            raise RuntimeError("nerf_bench_stamp failed with %d" % rc)
        return i


def kernel_durations(mk, step_fn, kernel_steps, lr, sync, device):
    """Per-kernel GPU time of the training step: {name: (avg ms per call, calls, avg units per call, total ms)}.

    The step is captured a second time with a one-lane time-stamp kernel in front of and behind every C-ABI call
    (on the capture stream, i.e. the stream the kernels are launched on), and that graph is replayed kernel_steps
    times: each replay stamps the slots, so the durations are those inside a graph-replayed step, launches back to
    back as in the timed region (HIP event records cannot be captured into a graph on ROCm).  The first stamp of a
    replay is taken twice, back to back: that distance -- one stamp kernel and one launch boundary, what the
    stamping itself adds -- is subtracted from every interval.  When the step runs without graphs it
    is issued eagerly with ordinary HIP events on the launch stream instead."""
    from nerf_amd import ops
    acc = {}
    if step_fn.use_graph:
        # Two copies of the graph, so that each carries few stamps (a stamp is a short idle gap; many of them let the
        # clock recover and the kernels of that replay run faster than in the timed region): one with stamps around
        # the three large MLP calls only, one with stamps around everything else.
        big = ("mlp_fwd_train", "mlp_bwd_data", "mlp_bwd_weights")
        rate = None
        replay_ms = {}
        for names in (big, None):
            stamper = Stamper(device, names=names)
            rate = stamper.rate
            if names is None:
                class _Rest:          # every call but the large ones
                    def __contains__(self, n):
                        return n not in big
                stamper.names = _Rest()
            try:
                ops.KERNEL_TIMES, ops.KERNEL_STAMP = {}, stamper
                st2 = mk(True)
                st2.rays_o.copy_(step_fn.rays_o); st2.rays_d.copy_(step_fn.rays_d); st2.target.copy_(step_fn.target)
                st2.prepare()
                pairs = ops.KERNEL_TIMES
            finally:
                ops.KERNEL_TIMES, ops.KERNEL_STAMP = None, None
            if not pairs:
                raise RuntimeError("no call of the step was stamped")
            st2(st2.rays_o, st2.rays_d, st2.target, lr=lr)
            for _ in range(kernel_steps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                st2(st2.rays_o, st2.rays_d, st2.target, lr=lr)
                e1.record()
                sync()
                replay_ms.setdefault("large calls stamped" if names is big else "other calls stamped", []).append(e0.elapsed_time(e1))
                t = stamper.slots.cpu()
                own = float(t[stamper.pair[1]] - t[stamper.pair[0]])
                for name, lst in pairs.items():
                    a = acc.setdefault(name, [0.0, 0, 0.0])
                    for i0, i1, u in lst:
                        a[0] += (float(t[i1] - t[i0]) - own) / stamper.rate * 1e3; a[1] += 1; a[2] += u
            del st2
        timing = ("one-lane time-stamp kernels (the device's %.0f MHz counter) captured into copies of the step's graph in "
                  "front of and behind each C-ABI call, on the stream the kernels are launched on, less the distance of two "
                  "back-to-back stamps; %d replays right after the timed region (one copy stamps the three large MLP calls, "
                  "another the rest); a replay of the stamped copies took %s ms"
                  % (rate / 1e6, kernel_steps, ", ".join("%s: %.4f" % (k, sorted(v)[len(v) // 2]) for k, v in replay_ms.items())))
    # the same calls issued one by one with ordinary HIP events on the launch stream (the host's launch gaps sit
    # between the kernels): the only figure when the step runs without graphs, a second opinion otherwise
    eager = {}
    was = step_fn.use_graph
    step_fn.use_graph = False
    try:
        step_fn(step_fn.rays_o, step_fn.rays_d, step_fn.target, lr=lr)
        ops.KERNEL_TIMES = {}
        for _ in range(kernel_steps):
            step_fn(step_fn.rays_o, step_fn.rays_d, step_fn.target, lr=lr)
        sync()
        evs, ops.KERNEL_TIMES = ops.KERNEL_TIMES, None
    finally:
        ops.KERNEL_TIMES = None
        step_fn.use_graph = was
    for name, lst in evs.items():
        a = eager.setdefault(name, [0.0, 0, 0.0])
        for e0, e1, u in lst:
            a[0] += e0.elapsed_time(e1); a[1] += 1; a[2] += u
    eager = {k: (a[0] / a[1], a[1], a[2] / a[1], a[0]) for k, a in eager.items() if a[1]}
    if not acc:
        timing = ("HIP events on the launch stream around each C-ABI call, %d eager steps of the same training step "
                  "run right after the timed region" % kernel_steps)
        return eager, timing, eager
    times = {k: (a[0] / a[1], a[1], a[2] / a[1], a[0]) for k, a in acc.items() if a[1]}
    return times, timing, eager


def render_sharded_section(mc, mf, pe, pv, device, args, rank, world, sync, max_over_ranks):
    """The two multi-GPU render measurements: (i) one 800x800 frame cut into row blocks, one per rank, assembled
    by one all_gather (render_nerf.py:260-307 per frame); (ii) frames of the 120-pose camera path with the white
    background and inverse-depth sampling dealt to the ranks, four per rank, no collective
    (render_nerf.py:145-149).  Times are the maximum over ranks between barriers."""
    from nerf_amd.parallel import gather_rows, shard_rows
    from nerf_amd.render import render_image, render_path
    from nerf_amd.utils.synthetic import camera_path, focal_from_fov, pose_spherical
    H = W = 800
    chunk = args.render_chunk[0]
    focal = focal_from_fov(W)
    c2w = pose_spherical(0.0, -30.0, 4.0).to(device)
    render_image(mc, mf, pe, pv, H, W, focal, c2w, chunk=chunk, rank=rank, world_size=world)      # warm-up
    frames = 3
    sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        render_image(mc, mf, pe, pv, H, W, focal, c2w, chunk=chunk, rank=rank, world_size=world)
    sync()
    dt = max_over_ranks(time.perf_counter() - t0) / frames
    block = torch.zeros((shard_rows(H, rank, world)[1], W, 3), device=device)
    gather_rows(block, H, rank, world)
    sync()
    t0 = time.perf_counter()
    for _ in range(10):
        gather_rows(block, H, rank, world)
    torch.cuda.synchronize()
    gather_us = max_over_ranks(time.perf_counter() - t0) / 10 * 1e6
    out = [{"workload": "800x800 frame, 640000 rays, 64+128 samples, chunk %d, no_grad, deterministic, image rows "
                        "sharded over %d ranks, one all_gather of the colour blocks" % (chunk, world),
            "s_per_frame": round(dt, 5), "rays_per_s": round(H * W / dt, 1), "all_gather_us": round(gather_us, 1),
            "all_gather_bytes": H * W * 12,
            "mlp_flop_frac_of_mfma_peak": round(H * W / dt * MLP_FLOP_PER_RAY_RENDER / (world * MFMA_PEAK_TFLOPS * 1e12), 4)}]
    per_rank = 4
    n_frames = per_rank * world
    idx = [int(i * 120 / n_frames) for i in range(n_frames)]
    poses = camera_path(120).to(device)[idx]
    render_path(mc, mf, pe, pv, H, W, focal, poses[:1], chunk=chunk, lindisp=True, white_bkgd=True)     # warm-up
    sync()
    t0 = time.perf_counter()
    got = render_path(mc, mf, pe, pv, H, W, focal, poses, chunk=chunk, lindisp=True, white_bkgd=True,
                      rank=rank, world_size=world)
    sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    out.append({"workload": "camera path: %d of the 120 poses (%d per rank, dealt round-robin, no collective), 800x800, "
                            "64+128 samples, lindisp, white background, chunk %d, uint8 frames on the host"
                            % (n_frames, per_rank, chunk),
                "frames_per_s": round(n_frames / dt, 3), "rays_per_s": round(n_frames * H * W / dt, 1),
                "s_per_frame_per_rank": round(dt / per_rank, 5), "frames_this_rank": len(got),
                "mlp_flop_frac_of_mfma_peak": round(n_frames * H * W / dt * MLP_FLOP_PER_RAY_RENDER
                                                    / (world * MFMA_PEAK_TFLOPS * 1e12), 4)})
    return out


def main():
    args = parse()
    if os.environ.get("NERF_BENCH_TRACE"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NERF_BENCH_TRACE"]), exit=False)     # where a hung rank is
    from nerf_amd.parallel import init_distributed
    rank, world, local_rank = init_distributed(args.dist_backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    device = torch.device("cuda", torch.cuda.current_device())
    import torch.distributed as dist
    from nerf_amd.parallel import all_reduce_, broadcast_parameters_

    n = args.rays_per_gpu
    mc, mf, pe, pv, opt = build(device, args.precision)
    broadcast_parameters_(opt.flat, world)
    stage("models built, parameters broadcast")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        return float(all_reduce_(torch.tensor([x], dtype=torch.float64, device=device), op=dist.ReduceOp.MAX).item())

    leg = train_leg(args, n, mc, mf, pe, pv, opt, device, rank, world, sync, max_over_ranks, args.kernel_steps)
    dt, times, graph_error = leg["dt"], leg["times"], leg["graph_error"]
    big = None
    if args.big_batch and args.big_batch != n:
        # the data-parallel configuration's batch (4096 rays per GPU), measured at every N so that its own
        # scaling curve exists; the headline value stays at the same per-GPU batch for every N
        big = train_leg(args, args.big_batch, mc, mf, pe, pv, opt, device, rank, world, sync, max_over_ranks, 0,
                        step0=leg["next_step"])
    default_flags = None
    if world == 1 and not args.no_render and not args.no_default_flags:
        # the reference script's default flags (train_nerf.py:62, :79): coarse network only, no view directions
        from nerf_amd.models.nerf import NeRF, PositionalEncoder
        from nerf_amd.optim import FlatAdam
        from nerf_amd.train import TrainStep
        from nerf_amd.utils.synthetic import seed_parameters_, synthetic_rays
        m0 = seed_parameters_(NeRF(input_ch_views=0, use_viewdirs=False, precision=args.precision).to(device), 0)
        o0 = FlatAdam(list(m0.parameters()), lr=5e-4)
        s0 = TrainStep(m0, None, PositionalEncoder(3, 10).to(device), None, o0, n, N_samples_coarse=64,
                       use_graph=not args.no_graph)
        ro0, rd0 = synthetic_rays(n, device, seed=0)
        s0.rays_o.copy_(ro0); s0.rays_d.copy_(rd0); s0.target.copy_(torch.rand(n, 3, device=device))
        for _ in range(args.warmup):
            s0(s0.rays_o, s0.rays_d, s0.target, lr=5e-4)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s0(s0.rays_o, s0.rays_d, s0.target, lr=5e-4)
        sync()
        d0 = (time.perf_counter() - t0) / args.steps
        default_flags = {"workload": "training step with the reference script's default flags: coarse network only (64 samples), "
                                     "no view directions, %d rays, fused 16-bit kernels" % n,
                         "ms_per_step": round(d0 * 1e3, 4), "rays_per_s": round(n / d0, 1)}
        del s0, o0, m0
    allreduce_us = None
    if world > 1:
        # the collective on its own: the whole gradient bucket, back to back, after the timed regions
        scratch_bucket = torch.zeros_like(opt.flat_grad)
        all_reduce_(scratch_bucket)
        sync()
        t1 = time.perf_counter()
        for _ in range(10):
            all_reduce_(scratch_bucket)
        torch.cuda.synchronize()
        allreduce_us = (time.perf_counter() - t1) / 10 * 1e6
    render_sharded = None
    if world > 1 and not args.no_render:
        render_sharded = render_sharded_section(mc, mf, pe, pv, device, args, rank, world, sync, max_over_ranks)

    if rank == 0:
        # per-kernel GPU time of the step (HIP events on the launch stream, inside the replayed graph)
        ktime = times
        flop = {"mlp_fwd_train": MLP_FWD_FLOP, "mlp_fwd": MLP_FWD_FLOP, "mlp_bwd_data": MLP_DGRAD_FLOP,
                "mlp_bwd_weights": MLP_WGRAD_FLOP}
        dom = max((k for k in ktime if k in flop), key=lambda k: ktime[k][3])
        avg_ms, launches, units, tot_ms = ktime[dom]
        mfma_kernel = units * flop[dom] / (avg_ms * 1e-3) / 1e12
        # The MLP kernels are priced against the dense MFMA peak with the reference algorithm's unpadded FLOP
        # counts: that is the bound a fused MLP has.  This design's weight-gradient pass also reads every saved
        # activation and every dZ unit once (280 KiB per 32 samples), so its HBM rate is given beside it.
        rl = {"bound": "mfma", "achieved": round(mfma_kernel, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
              "frac": round(mfma_kernel / MFMA_PEAK_TFLOPS, 4), "flop_per_sample": flop[dom]}
        hbm = None
        if dom == "mlp_bwd_weights":
            achieved = units * WGRAD_BYTES_PER_SAMPLE / (avg_ms * 1e-3) / 1e9
            hbm = {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(achieved / HBM_PEAK_GBS, 4), "bytes_per_sample": WGRAD_BYTES_PER_SAMPLE,
                   "algorithmic_bytes_per_launch": round(units * WGRAD_BYTES_PER_SAMPLE)}
        rays_per_s = n * world * args.steps / dt
        traffic = measured_traffic(dom, n, args.precision)
        roofline = {"kernel": dom, **rl, "traffic": traffic,
                    "traffic_note": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE), rocprofv3 PMC passes of this "
                                    "command, profiles/traffic.json" if traffic is not None else "not collected",
                    "hbm": hbm,
                    # the whole step against the dense MFMA peak: 893.2 MFLOP per ray for forward + backward of
                    # the 64 + 192 evaluations
                    "mfma_frac_kernel": round(mfma_kernel / MFMA_PEAK_TFLOPS, 4),
                    "step_mfma_frac": round(rays_per_s * MLP_FLOP_PER_RAY_TRAIN / (world * MFMA_PEAK_TFLOPS * 1e12), 4),
                    "mlp_flop_per_ray_fwd_bwd": MLP_FLOP_PER_RAY_TRAIN,
                    "avg_launch_us": round(avg_ms * 1e3, 2), "launches": launches,
                    # Second opinion: the same call between ordinary HIP events in eager steps.  The three large
                    # calls' stamped (and rocprofv3-traced) durations add up to more than the timed step itself, the
                    # eager ones do not: a kernel runs about a tenth slower when a stamp kernel or the profiler's
                    # packets separate it from its neighbours than back to back in the plain graph.
                    "avg_launch_us_eager_events": round(leg["eager_times"][dom][0] * 1e3, 2) if leg["eager_times"] and dom in leg["eager_times"] else None,
                    "samples_per_launch": units,
                    "timing": leg["timing"] + " (for mlp_bwd_weights: the contraction kernel plus its reduction "
                                              "and factored-gradient kernels, one C-ABI call)"}
        kernels = {}
        for k, (a_ms, cnt, u, tot) in sorted(ktime.items()):
            entry = {"avg_us": round(a_ms * 1e3, 2), "launches": cnt,
                     "share_of_step": round(tot / args.kernel_steps / (dt / args.steps * 1e3), 4)}
            if leg["eager_times"] and k in leg["eager_times"]:
                entry["avg_us_eager_events"] = round(leg["eager_times"][k][0] * 1e3, 2)
            if k in flop:
                entry["TFLOP/s"] = round(u * flop[k] / (a_ms * 1e-3) / 1e12, 1)
                entry["frac_of_mfma_peak"] = round(entry["TFLOP/s"] / MFMA_PEAK_TFLOPS, 4)
                if k == "mlp_bwd_weights":
                    entry["GB/s"] = round(u * WGRAD_BYTES_PER_SAMPLE / (a_ms * 1e-3) / 1e9, 1)
                    entry["frac_of_hbm_peak"] = round(entry["GB/s"] / HBM_PEAK_GBS, 4)
            elif k.startswith("composite") or k.startswith("sample_merge"):
                if k.startswith("composite"):
                    S = int(k.rsplit("S", 1)[1])
                    per_ray = (28 * S + 24) if "fwd" in k else (36 * S + 24)    # fwd also writes alpha [S]
                else:
                    per_ray = 64 * 4 * 2 + (128 * 4 if k.endswith("_u") else 0) + 192 * 4
                entry["GB/s"] = round(u * per_ray / (a_ms * 1e-3) / 1e9, 1)
                entry["frac_of_hbm_peak"] = round(entry["GB/s"] / HBM_PEAK_GBS, 4)
            kernels[k] = entry
        out = {
            "metric": "rays/sec (64+128 samples, fwd+bwd)", "value": round(rays_per_s, 1),
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "training step, %d rays/GPU, 64 coarse + 128 fine samples, coarse+fine 8x256 "
                                   "NeRF, fwd+bwd+Adam (benchmark configuration: single training step, batch 1024 rays)" % n,
                       "rays_per_gpu": n, "global_rays": n * world, "samples_per_ray": "64+192 evaluations",
                       "parallelism": "dp%d" % world if world > 1 else "single"},
            "timed_regions": {"repeats": len(leg["regions_ms_per_step"]), "ms_per_step_each": leg["regions_ms_per_step"],
                              "reported": "median region; each region is exactly --steps steps between barrier + "
                                          "synchronize, maximum over ranks"},
            "launch_mode": "eager" if args.no_graph else ("eager (%s)" % graph_error) if graph_error else ("hipgraph replay (one graph per step)" if world == 1 else
                                                         "hipgraph replay (3 graphs per step: the all-reduce of the fine network's gradients overlaps the "
                                                         "coarse backward, the coarse half follows, then Adam)"),
            "roofline": roofline, "kernels": kernels,
        }
        if big is not None:
            nb = args.big_batch
            out["train_%d_rays_per_gpu" % nb] = {
                "workload": "the same training step at %d rays/GPU (the data-parallel configuration's batch)" % nb,
                "rays_per_s": round(nb * world * args.steps / big["dt"], 1),
                "ms_per_step": round(big["dt"] / args.steps * 1e3, 4), "ms_per_step_each": big["regions_ms_per_step"],
                "step_mfma_frac": round(nb * args.steps / big["dt"] * MLP_FLOP_PER_RAY_TRAIN / (MFMA_PEAK_TFLOPS * 1e12), 4),
                "launch_mode": "eager (%s)" % big["graph_error"] if big["graph_error"] else "as the headline"}
        if default_flags is not None:
            out["train_default_flags"] = default_flags
        if allreduce_us is not None:
            out["allreduce"] = {"bytes": int(opt.flat_grad.numel()) * 4, "avg_us": round(allreduce_us, 1),
                                "note": "summing all-reduce of the whole fp32 gradient bucket of both networks, 10 back to back "
                                        "on rank 0's clock; in a step it is issued as two halves, the first hidden behind the "
                                        "coarse network's backward pass"}
        if render_sharded is not None:
            out["render_sharded"] = render_sharded
        if world == 1 and not args.no_render:
            out["render"] = render_section(mc, mf, pe, pv, device, args)
            out["whole_frame_kernels"] = whole_frame_kernels(device)
        if world == 1 and not args.no_one_eighth:
            out["one_eighth_gpu"] = one_eighth_gpu(args)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
