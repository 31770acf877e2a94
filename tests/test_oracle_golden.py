"""CPU: the oracle restatement against the golden vectors produced by the reference itself
(oracle/gen_golden.py).  Bit-exact where the op sequence is identical on any x86 ATen build;
a tight tolerance where ATen's vectorised kernels may differ between CPUs."""
import json
import os

import numpy as np
import pytest
import torch

from util import O, ROOT, T

torch.set_num_threads(4)


def close(a, b, atol=0.0, rtol=0.0):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape
    assert torch.allclose(a, b, atol=atol, rtol=rtol, equal_nan=True), (a - b).abs().max().item()


def test_weights_sha(golden):
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "meta.json")))
    assert O.params_sha256(O.init_params(0)) == meta["sha_coarse_seed0"]
    assert O.params_sha256(O.init_params(1)) == meta["sha_fine_seed1"]
    assert O.params_sha256(O.init_params(2, D=4, W=64, skips=(2,))) == meta["sha_mini_seed2"]
    assert sum(v.numel() for v in O.init_params(0).values()) == meta["n_params"] == 595844


def test_get_rays(golden):
    g = golden("f1_get_rays")
    for nm in ("identity", "sph40"):
        c2w = T(g[nm + "_c2w"])
        ro, rd = O.get_rays(8, 8, 5.0, c2w)
        close(ro, g[nm + "_o8"]); close(rd, g[nm + "_d8"], atol=1e-7)
        ro, rd = O.get_rays(800, 800, float(g["focal800"]), c2w)
        close(rd[392:408, 792:800], g[nm + "_d800_crop"], atol=1e-7)
        close(rd[:4, :4], g[nm + "_d800_corner"], atol=1e-7)
        close(rd.double().sum(dim=(0, 1)), g[nm + "_d800_sum"], atol=1e-3)
    # the reference's own known-answer tests (tests/test_ray_utils.py:56-143)
    ro, rd = O.get_rays(4, 4, 5.0, torch.eye(4))
    assert torch.allclose(ro, torch.zeros_like(ro), atol=1e-7)
    assert torch.allclose(rd[2, 2], torch.tensor([0.0, 0.0, -1.0]), atol=1e-6)
    tl = torch.tensor([-0.4, -0.4, -1.0]); tl = tl / tl.norm()
    assert torch.allclose(rd[0, 0], tl, atol=1e-6)
    c = torch.eye(4); c[:3, 3] = torch.tensor([1.0, 2.0, 3.0])
    ro2, rd2 = O.get_rays(4, 4, 5.0, c)
    assert torch.allclose(ro2, torch.tensor([1.0, 2.0, 3.0]).expand_as(ro2)) and torch.allclose(rd2, rd, atol=1e-7)


def test_posenc(golden):
    g = golden("f2_posenc")
    x = T(g["x"])
    close(O.positional_encode(x, 10), g["e10"], atol=1e-6)
    close(O.positional_encode(x, 4), g["e4"], atol=1e-6)
    close(O.positional_encode(x, 5, False), g["elin5"], atol=1e-6)
    # reference tests/test_nerf_model.py:26-38, :62-92 (dims; L=0 identity; L=1 at pi/2 and 0)
    assert O.positional_encode(torch.zeros(1, 3), 10).shape[-1] == 63
    assert O.positional_encode(torch.zeros(1, 2), 5).shape[-1] == 22
    assert O.positional_encode(torch.zeros(1, 3), 0).shape[-1] == 3
    p = torch.tensor([[np.pi / 2]], dtype=torch.float32)
    close(O.positional_encode(p, 1), torch.tensor([[np.pi / 2, 1.0, 0.0]]), atol=1e-7)
    close(O.positional_encode(torch.zeros(1, 1), 1), torch.tensor([[0.0, 0.0, 1.0]]), atol=1e-7)


def test_mlp(golden):
    g = golden("f3_mlp")
    ep = O.positional_encode(T(g["pts"]), 10)
    ev = O.positional_encode(T(g["viewdirs"]), 4)
    close(O.mlp_forward(O.init_params(0), ep, ev), g["raw"], atol=2e-5)
    close(O.mlp_forward(O.init_params(2, D=4, W=64, skips=(2,)), ep[:512], ev[:512], D=4, skips=(2,)),
          g["raw_mini"], atol=2e-5)
    close(O.mlp_forward(O.init_params(3, use_viewdirs=False), ep[:512], None, use_viewdirs=False),
          g["raw_noview"], atol=2e-5)
    with pytest.raises(ValueError):
        O.mlp_forward(O.init_params(0), ep, None)
    # layer shapes (reference tests/test_nerf_model.py:126-165)
    shapes = dict(O.param_shapes())
    assert shapes["pts_linears.0.weight"] == (256, 63) and shapes["pts_linears.5.weight"] == (256, 319)
    assert all(shapes[f"pts_linears.{k}.weight"] == (256, 256) for k in (1, 2, 3, 4, 6, 7))


def test_composite(golden):
    g = golden("f4_composite")
    raw, z = T(g["raw"]), T(g["z"])
    for wb in (0, 1):
        rr = raw.clone().requires_grad_(True)
        o = O.composite(rr, z, bool(wb))
        for k in ("rgb_map", "depth_map", "disp_map", "acc_map", "weights", "alpha", "raw_sigma"):
            close(o[k].detach(), g[f"{k}_wb{wb}"], atol=1e-6, rtol=1e-6)
        loss = (o["rgb_map"] * T(g[f"g_rgb_wb{wb}"])).sum() + (o["depth_map"] * T(g[f"g_depth_wb{wb}"])).sum() \
            + (o["acc_map"] * T(g[f"g_acc_wb{wb}"])).sum() + (o["weights"] * T(g[f"g_w_wb{wb}"])).sum()
        (gr,) = torch.autograd.grad(loss, rr)
        ref = T(g[f"draw_withw_wb{wb}"])
        close(gr, ref, atol=1e-5 * ref.abs().max().item(), rtol=1e-4)


def test_sample_pdf(golden):
    g = golden("f6_sample_pdf")
    bins, w, u = T(g["bins"]), T(g["weights"]), T(g["u"])
    s, inds, cdf = O.sample_pdf(bins, w, 128, det=False, u=u, return_inds=True)
    close(cdf, g["cdf"], atol=2.4e-7)
    exact = (inds.numpy() == g["inds_rand"])
    assert exact.mean() > 0.9999
    close(s[torch.from_numpy(exact)], T(g["samples_rand"])[torch.from_numpy(exact)], atol=2e-5)
    s, inds, _ = O.sample_pdf(bins, w, 128, det=True, return_inds=True)
    exact = (inds.numpy() == g["inds_det"])
    assert exact.mean() > 0.9999
    close(s[torch.from_numpy(exact)], T(g["samples_det"])[torch.from_numpy(exact)], atol=2e-5)
    # properties the reference's tests state (tests/test_nerf_model.py:171-238), real calling convention
    assert s.shape == (256, 128)
    assert torch.all(s >= bins.min(-1, keepdim=True).values - 1e-6)
    assert torch.all(s <= bins.max(-1, keepdim=True).values + 1e-6)
    assert torch.all(torch.diff(s[2]) >= -1e-6)          # uniform weights, deterministic -> sorted


def test_merge(golden):
    g = golden("f7_merge")
    zs, _ = torch.sort(torch.cat([T(g["z_coarse"]), T(g["z_fine"])], -1), -1)
    close(zs, g["z_sorted"])


def test_e2e_render(golden):
    g = golden("f8_e2e_render")
    ro, rd = O.synthetic_rays(1024, seed=0)
    close(ro, g["rays_o"], atol=1e-7); close(rd, g["rays_d"], atol=1e-7)
    pc, pf = O.init_params(0), O.init_params(1)
    with torch.no_grad():
        c, f = O.render_hierarchical(pc, pf, T(g["rays_o"]), T(g["rays_d"]), 2.0, 6.0, 64, 128, det=True)
    # random-init sigma is within 1e-6 of zero on a handful of rays: compare where the sign is stable
    ok = torch.from_numpy(np.abs(g["f_sigma_last_l0w0"]) > 1e-5) & torch.from_numpy(np.abs(g["c_sigma_last_l0w0"]) > 1e-5)
    assert ok.float().mean() > 0.99
    for k in ("rgb_map", "depth_map", "acc_map"):
        close(c[k][ok], T(g[f"c_{k}_l0w0"])[ok], atol=1e-4)
        close(f[k][ok], T(g[f"f_{k}_l0w0"])[ok], atol=1e-4)


def test_train_step(golden):
    g = golden("f9_train_step")
    e = golden("f8_e2e_render")
    NT = g["target"].shape[0]
    ro, rd = T(e["rays_o"])[:NT], T(e["rays_d"])[:NT]
    pc = {k: v.requires_grad_(True) for k, v in O.init_params(0).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.init_params(1).items()}
    tot, lc, lf, _, fine = O.train_step_loss(pc, pf, ro, rd, T(g["target"]), 2.0, 6.0, 64, 128,
                                             T(g["t_rand_coarse"]), T(g["u_fine"]), T(g["t_rand_fine"]))
    tot.backward()
    close(lc.detach(), g["loss_c"], rtol=1e-5); close(lf.detach(), g["loss_f"], rtol=1e-5)
    for tag, p in (("c", pc), ("f", pf)):
        norms = np.array([v.grad.double().norm().item() for v in p.values()])
        np.testing.assert_allclose(norms, g[f"gnorm_{tag}"], rtol=2e-3, atol=1e-9)
        close(p["sigma_linear.weight"].grad, g[f"g_{tag}_sigma_w"], atol=1e-3 * np.abs(g[f"g_{tag}_sigma_w"]).max())


def test_train_step_lindisp_white(golden):
    """the oracle against the reference's training step with lindisp + white background (gen_golden_f9b.py)"""
    g = golden("f9b_train_step_lindisp_white")
    pc = {k: v.requires_grad_(True) for k, v in O.init_params(0).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.init_params(1).items()}
    tot, lc, lf, c, f = O.train_step_loss(pc, pf, T(g["rays_o"]), T(g["rays_d"]), T(g["target"]), 2.0, 6.0, 64, 128,
                                          T(g["t_rand_coarse"]), T(g["u_fine"]), T(g["t_rand_fine"]), lindisp=True,
                                          white_bkgd=True)
    tot.backward()
    close(lc.detach(), g["loss_c"], rtol=1e-5); close(lf.detach(), g["loss_f"], rtol=1e-5)
    close(c["rgb_map"].detach(), g["c_rgb_map"], atol=1e-5)
    for tag, p in (("c", pc), ("f", pf)):
        norms = np.array([v.grad.double().norm().item() for v in p.values()])
        np.testing.assert_allclose(norms, g[f"gnorm_{tag}"], rtol=2e-3, atol=1e-9)
