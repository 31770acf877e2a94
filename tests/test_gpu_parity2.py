"""GPU parity, second set: the 16-bit kernels on ALL rays of the well-conditioned end-to-end fixture, the
combined backward entry point, checkpoints written by the reference's own classes, and the cache of packed
weights across graph-replayed training."""
import ctypes
import os

import numpy as np
import pytest
import torch

from util import O, ROOT, T, amd_model, maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def amd():
    import nerf_amd
    return nerf_amd


def _wellcond_params():
    """the weight set of fixture F8b: seeds 0 / 1 with sigma_linear.bias += 1 (every sigma_last is far from 0)"""
    pc, pf = O.init_params(0), O.init_params(1)
    pc["sigma_linear.bias"] = pc["sigma_linear.bias"] + 1.0
    pf["sigma_linear.bias"] = pf["sigma_linear.bias"] + 1.0
    return pc, pf


def test_wellconditioned_weights_sha():
    import json
    meta = json.load(open(os.path.join(GOLD, "meta.json")))
    pc, pf = _wellcond_params()
    assert O.params_sha256(pc) == meta["sha_coarse_seed0_sigma_bias1"]
    assert O.params_sha256(pf) == meta["sha_fine_seed1_sigma_bias1"]


# Tolerances of the unmasked comparison: max abs over all 1024 rays and the four variants (rgb and acc are in
# [0, 1], depth in [2, 6]): 1e-3 on rgb / depth / acc with bf16 operands, 1e-4 with fp16, for the coarse pass, the
# fine pass end to end, and the fine pass on the reference's own sample positions.  Measured on MI355X:
# bf16 rgb 1.1e-4, depth 7.9e-5; fp16 rgb 1.4e-5, depth 1.2e-5 (acc is 1 to 2e-7: every ray ends opaque).
E2E_TOL = {"bf16": {"c": (1e-3, 1e-3, 1e-3), "f_refpos": (1e-3, 1e-3, 1e-3), "f": (1e-3, 1e-3, 1e-3)},
           "fp16": {"c": (1e-4, 1e-4, 1e-4), "f_refpos": (1e-4, 1e-4, 1e-4), "f": (1e-4, 1e-4, 1e-4)}}


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_render_e2e_all_rays_wellconditioned(amd, golden, precision, capsys):
    """Config 1 (1024 rays, 64+128, deterministic, all four lindisp / white_bkgd variants) through the fused
    16-bit kernels against the reference's maps on EVERY ray: no conditioning mask (masked rays: 0)."""
    g, e = golden("f8b_e2e_wellcond"), golden("f8_e2e_render")
    ro, rd = T(e["rays_o"], DEV), T(e["rays_d"], DEV)
    pc, pf = _wellcond_params()
    mc, mf = amd_model(pc, precision), amd_model(pf, precision)
    pe, pv = amd.PositionalEncoder(3, 10).to(DEV), amd.PositionalEncoder(3, 4).to(DEV)
    tol = E2E_TOL[precision]
    worst = {}
    for lind in (0, 1):
        for wb in (0, 1):
            tag = f"l{lind}w{wb}"
            assert float(g[f"c_sigma_last_min_{tag}"]) > 0.5 and float(g[f"f_sigma_last_min_{tag}"]) > 0.5
            with torch.no_grad():
                c, f = amd.render_hierarchical(mc, mf, pe, pv, ro, rd, 2.0, 6.0, 64, 128, rand=False, det=True,
                                               lindisp=bool(lind), white_bkgd=bool(wb))
            for name, out in (("c", c), ("f", f)):
                for k, t in zip(("rgb_map", "acc_map", "depth_map"), tol[name]):
                    err = maxabs(out[k], T(g[f"{name}_{k}_{tag}"]))
                    worst[(name, k)] = max(worst.get((name, k), 0.0), err)
                    assert err < t, (precision, name, k, tag, err)
                d = (out["disp_map"].cpu() - T(g[f"{name}_disp_map_{tag}"])).abs()
                assert d.max() < 1e-2 * T(g[f"{name}_disp_map_{tag}"]).abs().max()
            if tag == "l0w0":       # the fine network on the reference's own sample positions
                with torch.no_grad():
                    f2 = amd.render_rays(mf, pe, pv, ro, rd, 2.0, 6.0, 192, z_vals_override=T(g["z_comb_l0w0"], DEV))
                for k, t in zip(("rgb_map", "acc_map", "depth_map"), tol["f_refpos"]):
                    err = maxabs(f2[k], T(g[f"f_{k}_{tag}"]))
                    worst[("f_refpos", k)] = err
                    assert err < t, (precision, "fine at reference positions", k, err)
    with capsys.disabled():
        print(f"\n[{precision}] F8b worst max-abs errors over 4 variants, 1024 rays, masked rays: 0 -> "
              + ", ".join(f"{n}.{k}={v:.2e}" for (n, k), v in sorted(worst.items())))


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("M", [1, 300, 4096 + 77])
def test_mlp_bwd_combined_call_equals_split_calls(amd, precision, M):
    """nerf_mlp_bwd (one call, one workspace) against nerf_mlp_bwd_data + nerf_mlp_bwd_weights: the same
    kernels on the same buffers, so the gradients are bit-identical; called through ctypes."""
    from nerf_amd import _C, ops
    lib = _C.lib
    prec = _C.PREC[precision]
    gen = torch.Generator().manual_seed(M)
    flat = torch.cat([v.reshape(-1) for v in O.init_params(0).values()]).to(DEV)
    packed = ops.mlp_pack(flat, precision)
    ro = (torch.rand(M, 3, generator=gen) * 4 - 2).to(DEV)
    rd = torch.nn.functional.normalize(torch.randn(M, 3, generator=gen), dim=-1).to(DEV)
    z = torch.zeros(M, 1, device=DEV)
    raw = torch.empty(M, 4, device=DEV)
    saved = torch.empty(int(lib.nerf_mlp_saved_bytes(M)), dtype=torch.uint8, device=DEV)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())        # noqa: E731
    _C.check(lib.nerf_mlp_fwd(M, 1, P(ro), P(rd), P(z), P(packed), prec, P(raw), P(saved), st), "fwd")
    d_raw = (torch.randn(M, 4, generator=gen) * 1e-3).to(DEV)
    scale = torch.tensor([256.0], device=DEV) if precision == "fp16" else None
    ps = None if scale is None else P(scale)

    ws_bytes = int(lib.nerf_mlp_bwd_workspace_bytes(M))
    assert ws_bytes == int(lib.nerf_mlp_dz_bytes(M)) + int(lib.nerf_mlp_partial_bytes(M))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    g_comb = torch.full((_C.NERF_MLP_NPARAMS,), float("nan"), device=DEV)
    _C.check(lib.nerf_mlp_bwd(M, P(d_raw), P(packed), prec, P(saved), ps, P(ws), ws_bytes, P(g_comb), st), "bwd")

    dz = torch.empty(int(lib.nerf_mlp_dz_bytes(M)), dtype=torch.uint8, device=DEV)
    part = torch.empty(int(lib.nerf_mlp_partial_bytes(M)), dtype=torch.uint8, device=DEV)
    g_split = torch.full((_C.NERF_MLP_NPARAMS,), float("nan"), device=DEV)
    _C.check(lib.nerf_mlp_bwd_data(M, P(d_raw), P(packed), prec, P(saved), ps, P(dz), st), "bwd_data")
    _C.check(lib.nerf_mlp_bwd_weights(M, P(packed), P(saved), P(dz), prec, ps, P(part), part.numel(), P(g_split), st),
             "bwd_weights")
    torch.cuda.synchronize()
    assert torch.isfinite(g_comb).all() and g_comb.abs().max() > 0
    assert torch.equal(g_comb, g_split)
    # a workspace that is one byte short is refused before anything is launched
    assert lib.nerf_mlp_bwd(M, P(d_raw), P(packed), prec, P(saved), ps, P(ws), ws_bytes - 1, P(g_comb), st) == -5


def _load_ref_ckpt(name):
    from nerf_amd import checkpoint as C
    ck = C.load_checkpoint(os.path.join(GOLD, name + ".pth"))
    return ck, dict(np.load(os.path.join(GOLD, name + "_render.npz")))


@pytest.mark.parametrize("name", ["f10_ckpt_hier_views", "f10_ckpt_default_flags"])
def test_reference_written_checkpoint_renders_and_resumes(amd, name):
    """Checkpoint files assembled by the reference's own NeRF class + torch.optim.Adam exactly as
    scripts/train_nerf.py:118-131 does (oracle/gen_golden.py): they load with the safe loader, build_models
    rebuilds the architecture from args_saved_at_checkpoint (render_nerf.py:188-252), the render equals the
    reference's render of the same weights (fp32 mode: 2e-5), and one more optimizer step from the restored
    Adam state lands on the reference's next weights."""
    from nerf_amd import checkpoint as C
    from nerf_amd.optim import FlatAdam
    from nerf_amd.train import TrainStep
    ck, g = _load_ref_ckpt(name)
    assert ck["epoch"] == 3 and ck["global_step"] == 2 and (ck["H"], ck["W"]) == (20, 24)
    mc, mf, pe, pv, cfg = C.build_models(ck, device=DEV, precision="fp32")
    a = ck["args_saved_at_checkpoint"]
    assert (mf is not None) == a["use_hierarchical"] and (pv is not None) == a["use_viewdirs"]
    assert cfg["precision"] == "fp32" and mc.D == 4 and mc.W == 32 and pe.output_dims == 39
    # (the default operand type, bf16, is kept for this architecture too: it runs on the layer-program kernels)
    assert C.build_models(ck, device=DEV)[4]["precision"] == "bf16"
    ro, rd = T(g["rays_o"], DEV), T(g["rays_d"], DEV)
    nc, nf = a["N_samples_coarse"], a["N_samples_fine"]
    with torch.no_grad():
        if mf is not None:
            c, f = amd.render_hierarchical(mc, mf, pe, pv, ro, rd, 2.0, 6.0, nc, nf, rand=False, det=True)
        else:
            c, f = amd.render_rays(mc, pe, pv, ro, rd, 2.0, 6.0, nc, use_viewdirs=False), None
    for k in ("rgb_map", "depth_map", "acc_map"):
        assert maxabs(c[k], T(g[f"c_{k}"])) < 2e-5, (k, maxabs(c[k], T(g[f"c_{k}"])))
        if f is not None:
            assert maxabs(f[k], T(g[f"f_{k}"])) < 1e-4, (k, maxabs(f[k], T(g[f"f_{k}"])))
    # resume: Adam moments and step count from the file, then the step the reference took next
    for m in (mc, mf):
        if m is not None:
            m.train()
    params = list(mc.parameters()) + (list(mf.parameters()) if mf is not None else [])
    opt = FlatAdam(params, lr=a["lr"])
    opt.load_state_dict(ck["optimizer_state_dict"])
    assert opt.current_step() == 2
    step = TrainStep(mc, mf, pe, pv, opt, ro.shape[0], N_samples_coarse=nc, N_samples_fine=nf, use_graph=False,
                     external_uniforms=True)
    u = [T(g["t_rand_coarse"], DEV).reshape(-1)]
    if mf is not None:
        u += [T(g["u_fine"], DEV).reshape(-1), T(g["t_rand_fine"], DEV).reshape(-1)]
    step.uniforms.copy_(torch.cat(u))
    loss = step(ro, rd, T(g["target"], DEV), lr=a["lr"])
    assert abs(loss.item() - float(g["loss_step3"])) < 1e-5 * float(g["loss_step3"])
    assert maxabs(mc.pts_linears[0].bias, T(g["c_l0_bias_after_step3"])) < 2e-6
    last = mc.rgb_layers[2].weight if a["use_viewdirs"] else mc.output_linear.weight
    assert maxabs(last, T(g["c_last_weight_after_step3"])) < 2e-6
    assert opt.current_step() == 3


def test_checkpoint_without_viewdirs_renders_a_frame(amd):
    """render_image on a checkpoint trained with the script's default flags (no view directions, coarse only)."""
    from nerf_amd import checkpoint as C
    from nerf_amd.render import render_image
    ck, g = _load_ref_ckpt("f10_ckpt_default_flags")
    c2w = O.pose_spherical(20.0, -30.0, 4.0).to(DEV)
    ro, rd = O.get_rays(20, 24, 30.0, c2w.cpu())
    # fp32 validation mode to summation order; the default operand type (bf16, layer-program kernels) to its rounding
    for precision, tol in (("fp32", 2e-5), ("bf16", 3e-3)):
        mc, mf, pe, pv, cfg = C.build_models(ck, device=DEV, precision=precision)
        assert mf is None and pv is None and cfg["precision"] == precision
        img = render_image(mc, mf, pe, pv, ck["H"], ck["W"], ck["focal"], c2w, N_samples_coarse=16, chunk=100)
        assert img.shape == (20, 24, 3) and torch.isfinite(img).all()
        sd = {k: v.cpu() for k, v in mc.state_dict().items()}
        with torch.no_grad():
            ref = O.render_rays(sd, ro.reshape(-1, 3), rd.reshape(-1, 3), 2.0, 6.0, 16, L_pts=6, use_viewdirs=False, D=4,
                                skips=(2,))
        assert maxabs(img.reshape(-1, 3), ref["rgb_map"]) < tol


@pytest.mark.parametrize("use_graph", [True, False])
def test_packed_weight_cache_follows_graph_replayed_training(amd, use_graph):
    """train -> render -> train -> render: the second render must use the weights of the second training
    leg.  (A graph replay updates the weights on the device without touching the parameters' version
    counters, which the no_grad cache of packed weights is keyed on.)"""
    from nerf_amd.optim import FlatAdam
    from nerf_amd.train import TrainStep
    from nerf_amd.utils.synthetic import synthetic_rays
    n = 256
    mc, mf = amd_model(O.init_params(0), "bf16"), amd_model(O.init_params(1), "bf16")
    pe, pv = amd.PositionalEncoder(3, 10).to(DEV), amd.PositionalEncoder(3, 4).to(DEV)
    opt = FlatAdam(list(mc.parameters()) + list(mf.parameters()), lr=2e-3)
    ro, rd = synthetic_rays(n, DEV, seed=3)
    tgt = torch.rand(n, 3, generator=torch.Generator().manual_seed(3)).to(DEV)
    step = TrainStep(mc, mf, pe, pv, opt, n, use_graph=use_graph)

    def render():
        with torch.no_grad():
            return amd.render_hierarchical(mc, mf, pe, pv, ro, rd, 2.0, 6.0, 64, 128)[1]["rgb_map"].clone()

    for _ in range(5):
        step(ro, rd, tgt, lr=2e-3)
    r1 = render()
    for _ in range(5):
        step(ro, rd, tgt, lr=2e-3)
    r2 = render()
    assert not torch.equal(r1, r2)
    # fresh models loaded with the current weights render the same pixels
    mc2, mf2 = amd_model({k: v.detach().cpu() for k, v in mc.state_dict().items()}, "bf16"), \
        amd_model({k: v.detach().cpu() for k, v in mf.state_dict().items()}, "bf16")
    with torch.no_grad():
        r_fresh = amd.render_hierarchical(mc2, mf2, pe, pv, ro, rd, 2.0, 6.0, 64, 128)[1]["rgb_map"]
    assert torch.equal(r2, r_fresh)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_merged_weight_gradient_call_equals_separate_calls(precision):
    """nerf_mlp_bwd_weights_multi (both networks of a step in one launch sequence, straight into two slices of one
    bucket) against one nerf_mlp_bwd_weights call per network: the sample axis is cut differently, so the sums
    agree to fp32 rounding (1e-5 of the largest gradient), and the merged call is bit-reproducible."""
    from nerf_amd import ops, _C
    lib, _p = ops.lib, ops._p
    P = ops._prec(precision, True)
    st = torch.cuda.current_stream().cuda_stream
    entries, separate = [], []
    for seed, (N, S) in enumerate(((96, 64), (96, 192))):
        m = amd_model(O.init_params(seed), precision)
        packed = m.packed_weights()
        M = N * S
        ro, rd = O.synthetic_rays(N, seed=seed)
        z = torch.sort(torch.rand(N, S, generator=torch.Generator().manual_seed(seed)) * 4 + 2, dim=-1).values
        raw = torch.empty(N, S, 4, device=DEV)
        saved = torch.empty(int(lib.nerf_mlp_saved_bytes(M)), dtype=torch.uint8, device=DEV)
        dz = torch.empty(int(lib.nerf_mlp_dz_bytes(M)), dtype=torch.uint8, device=DEV)
        d_raw = (torch.randn(N, S, 4, generator=torch.Generator().manual_seed(10 + seed)) * 1e-3).to(DEV)
        ro, rd, z = ro.to(DEV).contiguous(), rd.to(DEV).contiguous(), z.to(DEV).contiguous()
        ops.check(lib.nerf_mlp_fwd(N, S, _p(ro), _p(rd), _p(z), _p(packed), P, _p(raw), _p(saved), st), "fwd")
        scale = torch.full((1,), 1024.0, device=DEV) if precision == "fp16" else None
        ops.check(lib.nerf_mlp_bwd_data(M, _p(d_raw), _p(packed), P, _p(saved), _p(scale), _p(dz), st), "dgrad")
        part = torch.empty(int(lib.nerf_mlp_partial_bytes(M)), dtype=torch.uint8, device=DEV)
        d = torch.empty(_C.NERF_MLP_NPARAMS, device=DEV)
        ops.check(lib.nerf_mlp_bwd_weights(M, _p(packed), _p(saved), _p(dz), P, _p(scale), _p(part), part.numel(), _p(d), st), "wgrad")
        separate.append(d)
        entries.append(dict(M=M, packed=packed, saved=saved, dz=dz, prec=P, scale=scale, n_params=_C.NERF_MLP_NPARAMS))
    bucket = torch.zeros(2 * _C.NERF_MLP_NPARAMS, device=DEV)
    sinks = [bucket[:_C.NERF_MLP_NPARAMS], bucket[_C.NERF_MLP_NPARAMS:]]
    ops.mlp_bwd_weights_multi(entries, sinks)
    first = bucket.clone()
    for a, b in zip(sinks, separate):
        assert maxabs(a, b) < 1e-5 * b.abs().max().item()
    bucket.zero_()
    ops.mlp_bwd_weights_multi(entries, sinks)
    assert torch.equal(bucket, first)
