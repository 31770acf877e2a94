"""GPU parity of the fp32 validation mode (precision='fp32'): HIP encodings and compositing around the fp32
MFMA kernels (forward, backward-data chain, weight gradients).  Tolerances: raw 2e-5 abs (summation order only),
maps 1e-3 abs on all rays, gradients 2e-3 relative against the reference and 2e-5 against library GEMMs."""
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


@pytest.mark.parametrize("arch", [
    dict(),                                                        # the scripts' network
    dict(D=4, W=64, skips=[2]),                                    # reference tests/test_nerf_model.py:105-109
    dict(use_viewdirs=False),                                      # the script default (train_nerf.py:79)
    dict(D=3, W=100, skips=[0], input_ch_pts=27, input_ch_views=15),   # widths that need padding
    dict(D=12, W=32, skips=[1, 5, 9]),                             # several skips, the deepest program (16 layers)
])
def test_fp32_kernel_runs_every_architecture(amd, arch):
    """The fp32 MFMA kernel is what evaluates the validation mode (not a library GEMM), for ragged sizes too,
    and agrees with layer-by-layer fp32 GEMMs to summation order (2e-5 abs on raw of order 0.1)."""
    from nerf_amd import ops
    torch.manual_seed(7)
    m = amd.NeRF(precision="fp32", **arch).to(DEV)
    assert m.fp32_kernel_supported() and not m.fused()
    for M in (1, 63, 1000):
        xp = torch.randn(M, m.input_ch_pts, device=DEV)
        xv = torch.randn(M, m.input_ch_views, device=DEV) if m.use_viewdirs else None
        ops.KERNEL_TIMES = {}
        with torch.no_grad():
            raw = m(xp, xv) if xv is not None else m(xp)
            times, ops.KERNEL_TIMES = ops.KERNEL_TIMES, None
            ref = m._forward_library(xp, xv, None)
        assert "mlp_fwd_fp32" in times
        assert raw.shape == (M, 4) and maxabs(raw, ref) < 2e-5
    # leading dimensions are kept; gradients come out as the library's
    xp = torch.randn(5, 7, m.input_ch_pts, device=DEV)
    xv = torch.randn(5, 7, m.input_ch_views, device=DEV) if m.use_viewdirs else None
    cot = torch.randn(5, 7, 4, device=DEV)
    out = m(xp, xv) if xv is not None else m(xp)
    assert out.shape == (5, 7, 4)
    ops.KERNEL_TIMES = {}
    g_kernel = torch.autograd.grad(out, list(m.parameters()), cot)
    times, ops.KERNEL_TIMES = ops.KERNEL_TIMES, None
    assert "mlp_bwd_fp32" in times                     # the gradients are this library's kernels, not library GEMMs
    g_lib = torch.autograd.grad(m._forward_library(xp, xv, None), list(m.parameters()), cot)
    for a, b in zip(g_kernel, g_lib):
        assert a.shape == b.shape and maxabs(a, b) <= 2e-5 * (1e-3 + b.abs().max().item())
    # ragged sample counts (tiles of 64, chunks of the weight-gradient contraction), every parameter tensor:
    # difference to the library's gradient relative to its norm.  fp32 products, summation order only: 2e-5 -- until,
    # among millions of hidden units, a pre-activation lies within rounding of zero and its ReLU mask differs between
    # the two evaluations (a flipped unit moves the gradients below it by its whole contribution): 2e-3 at 4097 samples
    for M in (1, 200, 4097):
        xp = torch.randn(M, m.input_ch_pts, device=DEV)
        xv = torch.randn(M, m.input_ch_views, device=DEV) if m.use_viewdirs else None
        cot = torch.randn(M, 4, device=DEV)
        g_kernel = torch.autograd.grad(m(xp, xv) if xv is not None else m(xp), list(m.parameters()), cot)
        g_lib = torch.autograd.grad(m._forward_library(xp, xv, None), list(m.parameters()), cot)
        for (k, _), a, b in zip(m.named_parameters(), g_kernel, g_lib):
            rel = ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
            assert rel < (2e-5 if M <= 200 else 5e-3), (arch, M, k, rel)
    # size-independent: the gradient over 4097 samples is the sum of the gradients over two parts of them (a sample's
    # masks do not depend on its batch), which checks tiles, chunks and tails without the mask-flip allowance
    parts = [torch.autograd.grad(m(xp[a:b], xv[a:b]) if xv is not None else m(xp[a:b]), list(m.parameters()), cot[a:b])
             for a, b in ((0, 2048), (2048, 4097))]
    for (k, _), whole, p0, p1 in zip(m.named_parameters(), g_kernel, *parts):
        rel = ((whole.double() - p0.double() - p1.double()).norm() / (whole.double().norm() + 1e-30)).item()
        assert rel < 2e-6, (arch, k, rel)
    # a gradient with respect to the encoded inputs still comes from the library path
    xp = torch.randn(9, m.input_ch_pts, device=DEV, requires_grad=True)
    xv = torch.randn(9, m.input_ch_views, device=DEV) if m.use_viewdirs else None
    (m(xp, xv) if xv is not None else m(xp)).sum().backward()
    assert xp.grad is not None and torch.isfinite(xp.grad).all() and xp.grad.abs().max() > 0
    m.zero_grad()
    # weights changed in place under no_grad: the cached program follows
    with torch.no_grad():
        before = m(xp, xv) if xv is not None else m(xp)
        for p in m.parameters():
            p.mul_(0.5)
        after = m(xp, xv) if xv is not None else m(xp)
        assert maxabs(after, m._forward_library(xp, xv, None)) < 2e-5 and maxabs(after, before) > 1e-4


def test_fp32_kernel_limits_fall_back(amd):
    """Beyond the kernel's widths the validation mode still runs (library GEMMs) and says so."""
    m = amd.NeRF(D=2, W=320, skips=[], precision="fp32").to(DEV)
    assert not m.fp32_kernel_supported()
    with torch.no_grad():
        assert m(torch.randn(10, 63, device=DEV), torch.randn(10, 27, device=DEV)).shape == (10, 4)
