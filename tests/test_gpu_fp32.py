"""GPU parity of the fp32 validation mode (precision='fp32'): HIP encodings and compositing around
fp32 library GEMMs.  Tolerances: raw 2e-5 abs (GEMM summation order only), maps 1e-3 abs on all rays,
gradients 1e-3 relative."""
import numpy as np
import pytest
import torch

from util import O, T, amd_model, maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def amd():
    import nerf_amd
    return nerf_amd


def test_fp32_mlp_golden(amd, golden):
    g = golden("f3_mlp")
    pe, pv = amd.PositionalEncoder(3, 10).to(DEV), amd.PositionalEncoder(3, 4).to(DEV)
    ep, ev = pe(T(g["pts"], DEV)), pv(T(g["viewdirs"], DEV))
    with torch.no_grad():
        raw = amd_model(O.init_params(0), "fp32")(ep, ev)
        assert maxabs(raw, T(g["raw"])) < 2e-5
        # architectures outside the fused kernels (reference tests/test_nerf_model.py:105-109)
        mini = amd_model(O.init_params(2, D=4, W=64, skips=(2,)), "fp32", D=4, W=64, skips=[2])
        assert maxabs(mini(ep[:512], ev[:512]), T(g["raw_mini"])) < 2e-5
        nv = amd_model(O.init_params(3, use_viewdirs=False), "fp32", use_viewdirs=False)
        assert maxabs(nv(ep[:512]), T(g["raw_noview"])) < 2e-5
    with pytest.raises(ValueError):
        amd_model(O.init_params(0), "fp32")(ep)           # nerf.py:188-189
    with pytest.raises(ValueError):
        amd_model(O.init_params(0), "fp8")(ep, ev)


@pytest.mark.parametrize("lindisp,white", [(False, False), (True, True)])
def test_fp32_end_to_end_all_rays(amd, golden, lindisp, white):
    """config 1 in fp32 mode: every ray within 1e-3 of the reference's maps, no conditioning mask;
    depth is compared where the fine pass is well conditioned in fp32 as well (|sigma_last| > 1e-4)."""
    e = golden("f8_e2e_render")
    tag = f"l{int(lindisp)}w{int(white)}"
    ro, rd = T(e["rays_o"], DEV), T(e["rays_d"], DEV)
    mc, mf = amd_model(O.init_params(0), "fp32"), amd_model(O.init_params(1), "fp32")
    pe, pv = amd.PositionalEncoder(3, 10).to(DEV), amd.PositionalEncoder(3, 4).to(DEV)
    with torch.no_grad():
        c, f = amd.render_hierarchical(mc, mf, pe, pv, ro, rd, 2.0, 6.0, 64, 128, rand=False, det=True,
                                       lindisp=lindisp, white_bkgd=white)
    for name, out in (("c", c), ("f", f)):
        assert maxabs(out["rgb_map"], T(e[f"{name}_rgb_map_{tag}"])) < 1e-3
        assert maxabs(out["acc_map"], T(e[f"{name}_acc_map_{tag}"])) < 1e-3
        ok = torch.from_numpy(np.abs(e[f"{name}_sigma_last_{tag}"]) > 1e-4)
        assert ok.float().mean() > 0.99
        d = (out["depth_map"].cpu() - T(e[f"{name}_depth_map_{tag}"])).abs()
        assert d[ok].max() < (1e-3 if name == "c" else 5e-3)


def test_fp32_train_step_golden(amd, golden):
    """config 2 at fixture size in fp32 mode: losses to 1e-5 relative, gradients to 1e-3 relative."""
    g, e = golden("f9_train_step"), golden("f8_e2e_render")
    NT = g["target"].shape[0]
    ro, rd, tgt = T(e["rays_o"], DEV)[:NT], T(e["rays_d"], DEV)[:NT], T(g["target"], DEV)
    mc, mf = amd_model(O.init_params(0), "fp32"), amd_model(O.init_params(1), "fp32")
    pe, pv = amd.PositionalEncoder(3, 10).to(DEV), amd.PositionalEncoder(3, 4).to(DEV)
    c, f = amd.render_hierarchical(mc, mf, pe, pv, ro, rd, 2.0, 6.0, 64, 128, rand=True, det=False,
                                   t_rand_coarse=T(g["t_rand_coarse"], DEV), u_fine=T(g["u_fine"], DEV),
                                   t_rand_fine=T(g["t_rand_fine"], DEV))
    loss_c = torch.mean((c["rgb_map"] - tgt) ** 2)
    loss_f = torch.mean((f["rgb_map"] - tgt) ** 2)
    (loss_c + loss_f).backward()
    assert abs(loss_c.item() - float(g["loss_c"])) < 1e-5 * float(g["loss_c"])
    assert abs(loss_f.item() - float(g["loss_f"])) < 1e-4 * float(g["loss_f"])
    rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()   # noqa: E731
    for tag, m in (("c", mc), ("f", mf)):
        norms = np.array([p.grad.double().norm().item() for p in m.parameters()])
        assert np.all(np.abs(norms - g[f"gnorm_{tag}"]) < 2e-3 * g[f"gnorm_{tag}"] + 1e-12), norms / g[f"gnorm_{tag}"]
        assert rel(m.sigma_linear.weight.grad, T(g[f"g_{tag}_sigma_w"])) < 2e-3
        assert rel(m.rgb_layers[2].weight.grad, T(g[f"g_{tag}_rgb2_w"])) < 2e-3
        assert rel(m.pts_linears[0].bias.grad, T(g[f"g_{tag}_l0_b"])) < 2e-3
