"""GPU parity tests: every HIP kernel, called through the C ABI (ctypes), against the CPU
oracle on the same seeded inputs and against the committed golden vectors.

Tolerances (stated per test): integer/index/sort results are bit-exact; fp32 kernels agree with
the oracle to a few ulp (different exp/sin implementations and summation order); the bf16/fp16
MFMA MLP agrees to the operand precision.
"""
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


# ---------------------------------------------------------------------------
def test_library_loaded(amd):
    from nerf_amd import _C
    assert _C.lib.nerf_version() == 100
    assert torch.cuda.is_available()


def test_no_cpu_fallback(amd):
    from nerf_amd import ops
    with pytest.raises(RuntimeError):
        ops.get_rays(4, 4, 5.0, torch.eye(4))
    with pytest.raises(RuntimeError):
        amd.sample_pdf(torch.rand(2, 5), torch.rand(2, 4), 8, det=True)
    from nerf_amd.models.nerf import NeRF, PositionalEncoder, render_rays
    m, pe, pv = NeRF().to(DEV), PositionalEncoder(3, 10).to(DEV), PositionalEncoder(3, 4).to(DEV)
    with pytest.raises(RuntimeError, match="no CPU path"):
        render_rays(m, pe, pv, torch.rand(4, 3), torch.rand(4, 3), 2.0, 6.0, 8, False)
    with pytest.raises(TypeError, match="float32"):
        render_rays(m, pe, pv, torch.rand(4, 3, device=DEV).double(), torch.rand(4, 3, device=DEV).double(), 2.0, 6.0, 8, False)


# ---------------------------------------------------------------------------
def test_get_rays(amd, golden):
    g = golden("f1_get_rays")
    for nm in ("identity", "sph40"):
        c2w = T(g[nm + "_c2w"], DEV)
        ro, rd = amd.get_rays(8, 8, 5.0, c2w)
        assert ro.shape == (8, 8, 3) and rd.is_contiguous() and rd.device.type == "cuda"
        assert maxabs(ro, T(g[nm + "_o8"])) == 0.0
        assert maxabs(rd, T(g[nm + "_d8"])) < 2e-7          # fp32, one division + sqrt differ by <= 1 ulp
        ro, rd = amd.get_rays(800, 800, float(g["focal800"]), c2w)
        assert maxabs(rd[392:408, 792:800], T(g[nm + "_d800_crop"])) < 2e-7
        assert maxabs(rd[:4, :4], T(g[nm + "_d800_corner"])) < 2e-7
        assert maxabs(rd.double().sum(dim=(0, 1)), T(g[nm + "_d800_sum"])) < 1e-2
        assert torch.allclose(rd.norm(dim=-1), torch.ones(800, 800, device=DEV), atol=1e-6)
        # row-block form used for sharding: rows 100..149 equal the slice of the full frame
        ro_b, rd_b = amd.get_rays(800, 800, float(g["focal800"]), c2w, row0=100, nrows=50)
        assert torch.equal(rd_b, rd[100:150]) and torch.equal(ro_b, ro[100:150])
        # [3,4] pose and an odd, non-multiple-of-4 frame (scalar tail path)
        ro3, rd3 = amd.get_rays(7, 9, 5.0, c2w[:3].contiguous())
        oro, ord_ = O.get_rays(7, 9, 5.0, c2w.cpu())
        assert maxabs(rd3, ord_) < 2e-7 and maxabs(ro3, oro) == 0.0
    # reference known answers (tests/test_ray_utils.py:56-143)
    ro, rd = amd.get_rays(4, 4, 5.0, torch.eye(4, device=DEV))
    assert torch.all(ro == 0)
    assert torch.allclose(rd[2, 2].cpu(), torch.tensor([0.0, 0.0, -1.0]), atol=1e-6)
    tl = torch.tensor([-0.4, -0.4, -1.0]); tl = tl / tl.norm()
    assert torch.allclose(rd[0, 0].cpu(), tl, atol=1e-6)


def test_posenc(amd, golden):
    g = golden("f2_posenc")
    x = T(g["x"], DEV)
    e10 = amd.PositionalEncoder(3, 10).to(DEV)(x)
    e4 = amd.PositionalEncoder(3, 4).to(DEV)(x)
    el = amd.PositionalEncoder(3, 5, log_sampling=False).to(DEV)(x)
    # fp32 sin/cos of the same fp32 argument: <= 1e-6 abs (arguments reach 3e3 rad)
    assert maxabs(e10, T(g["e10"])) < 1e-6 and maxabs(e4, T(g["e4"])) < 1e-6 and maxabs(el, T(g["elin5"])) < 1e-6
    assert amd.PositionalEncoder(3, 10).output_dims == 63 and amd.PositionalEncoder(2, 5).output_dims == 22
    e0 = amd.PositionalEncoder(3, 0).to(DEV)(x)
    assert torch.equal(e0, x)
    xm = torch.randn(5, 7, 3, device=DEV)
    assert amd.PositionalEncoder(3, 10).to(DEV)(xm).shape == (5, 7, 63)
    p = torch.tensor([[np.pi / 2]], dtype=torch.float32, device=DEV)
    out = amd.PositionalEncoder(1, 1).to(DEV)(p).cpu()
    assert torch.allclose(out, torch.tensor([[np.pi / 2, 1.0, 0.0]]), atol=1e-7)


def test_place_samples(amd):
    from nerf_amd import ops
    N = 33
    for S in (64, 192, 5):
        for lind in (False, True):
            z = ops.place_samples(N, 2.0, 6.0, S, lind, torch.device(DEV))
            zo = O.place_samples(N, 2.0, 6.0, S, lind)
            assert maxabs(z, zo) < 1e-6           # fp32, <= 2 ulp at z ~ 6
            tr = torch.rand(N, S)
            z = ops.place_samples(N, 2.0, 6.0, S, lind, torch.device(DEV), t_rand=tr.to(DEV))
            assert maxabs(z, O.place_samples(N, 2.0, 6.0, S, lind, t_rand=tr)) < 1e-6
    # per-ray bounds and the override + jitter path (fine pass of training)
    near, far = torch.rand(N, 1) + 1.5, torch.rand(N, 1) + 5.0
    z = ops.place_samples(N, near.to(DEV), far.to(DEV), 64, False, torch.device(DEV))
    assert maxabs(z, O.place_samples(N, near, far, 64, False)) < 1e-6
    zov, _ = torch.sort(torch.rand(N, 192) * 4 + 2, -1)
    tr = torch.rand(N, 192)
    z = ops.place_samples(N, 0, 0, 0, False, torch.device(DEV), z_vals_override=zov.to(DEV), t_rand=tr.to(DEV))
    assert maxabs(z, O.place_samples(N, 0, 0, 0, False, z_vals_override=zov, t_rand=tr)) < 1e-6


# ---------------------------------------------------------------------------
def test_composite_golden(amd, golden):
    from nerf_amd import ops
    g = golden("f4_composite")
    raw, z = T(g["raw"], DEV), T(g["z"], DEV)
    for wb in (0, 1):
        rr = raw.clone().requires_grad_(True)
        rgb, depth, disp, acc, w, alpha = ops.composite(rr, z, bool(wb))
        # forward: 1e-5 abs on maps of O(1..6) magnitude, 1e-6 on per-sample quantities
        assert maxabs(rgb, T(g[f"rgb_map_wb{wb}"])) < 1e-5
        assert maxabs(depth, T(g[f"depth_map_wb{wb}"])) < 1e-5
        assert maxabs(acc, T(g[f"acc_map_wb{wb}"])) < 1e-5
        d_ref = T(g[f"disp_map_wb{wb}"])
        assert torch.allclose(disp.cpu(), d_ref, rtol=1e-5, atol=1e-6)
        assert maxabs(w, T(g[f"weights_wb{wb}"])) < 2e-6
        assert maxabs(alpha, T(g[f"alpha_wb{wb}"])) < 2e-6
        # backward: 1e-4 relative to the largest gradient entry
        loss = (rgb * T(g[f"g_rgb_wb{wb}"], DEV)).sum() + (depth * T(g[f"g_depth_wb{wb}"], DEV)).sum() \
            + (acc * T(g[f"g_acc_wb{wb}"], DEV)).sum()
        (gr,) = torch.autograd.grad(loss, rr, retain_graph=True)
        ref = T(g[f"draw_wb{wb}"])
        assert maxabs(gr, ref) < 1e-4 * ref.abs().max().item()
        loss2 = loss + (w * T(g[f"g_w_wb{wb}"], DEV)).sum()
        (gr2,) = torch.autograd.grad(loss2, rr)
        ref2 = T(g[f"draw_withw_wb{wb}"])
        assert maxabs(gr2, ref2) < 1e-4 * ref2.abs().max().item()


@pytest.mark.parametrize("S", [2, 7, 64, 65, 128, 192, 256, 300, 700, 1024, 1025, 1500, 2113])
def test_composite_sizes(amd, S):
    """Ragged and multi-chunk sample counts, all outputs and all cotangents incl. disp and alpha.  The reference
    has no upper limit on the samples per ray (nerf.py:377-409): beyond 256 the forward, and beyond 1024 the
    backward, run their run-time chunk loops."""
    from nerf_amd import ops
    gen = torch.Generator().manual_seed(S)
    N = 37
    raw = torch.randn(N, S, 4, generator=gen)
    raw[..., 3] = raw[..., 3] * 3 + 0.5
    z, _ = torch.sort(torch.rand(N, S, generator=gen) * 4 + 2, -1)
    gs = [torch.randn(N, 3, generator=gen), torch.randn(N, generator=gen), torch.randn(N, generator=gen) * 0.1,
          torch.randn(N, generator=gen), torch.randn(N, S, generator=gen), torch.randn(N, S, generator=gen)]
    rr = raw.clone().requires_grad_(True)
    o = O.composite(rr, z, True)
    outs_ref = [o["rgb_map"], o["depth_map"], o["disp_map"], o["acc_map"], o["weights"], o["alpha"]]
    (g_ref,) = torch.autograd.grad(sum((a * b).sum() for a, b in zip(outs_ref, gs)), rr)
    rg = raw.to(DEV).requires_grad_(True)
    outs = ops.composite(rg, z.to(DEV), True)
    for a, b in zip(outs, outs_ref):
        assert torch.allclose(a.cpu(), b.detach(), rtol=2e-5, atol=1e-5)
    (g_gpu,) = torch.autograd.grad(sum((a * b.to(DEV)).sum() for a, b in zip(outs, gs)), rg)
    assert maxabs(g_gpu, g_ref) < 2e-4 * g_ref.abs().max().item()


def test_composite_empty(amd):
    from nerf_amd import ops
    outs = ops.composite(torch.zeros(0, 64, 4, device=DEV), torch.zeros(0, 64, device=DEV), False)
    assert outs[0].shape == (0, 3) and outs[4].shape == (0, 64)


@pytest.mark.parametrize("precision", ["bf16", "fp16", "fp32"])
def test_empty_inputs_everywhere(amd, precision):
    """No rays, no rows: every public entry point returns empty results of the right shape, as the reference's
    torch code does (an empty tensor has no address, so the C ABI must not ask for one), and a backward pass
    over no rays leaves zero gradients.  One ray of two samples, the smallest ray batch the compositing takes."""
    from nerf_amd import ops
    from nerf_amd.models.nerf import NeRF, PositionalEncoder, render_rays, sample_pdf, render_hierarchical
    pe, pv = PositionalEncoder(3, 10).to(DEV), PositionalEncoder(3, 4).to(DEV)
    m = NeRF(precision=precision).to(DEV)
    none = torch.zeros(0, 3, device=DEV)
    out = render_rays(m, pe, pv, none, none, 2.0, 6.0, 64, False)
    assert out["rgb_map"].shape == (0, 3) and out["weights"].shape == (0, 64) and out["depth_map"].shape == (0,)
    assert m(torch.zeros(0, 63, device=DEV), torch.zeros(0, 27, device=DEV)).shape == (0, 4)
    out = render_rays(m, pe, pv, none, none, 2.0, 6.0, 64, True)
    out["rgb_map"].sum().backward()
    assert all(p.grad is not None and not p.grad.any() for p in m.parameters())
    if precision == "bf16":
        coarse, fine = render_hierarchical(m, NeRF(precision=precision).to(DEV), pe, pv, none, none, 2.0, 6.0, 64, 128,
                                           rand=False, det=True)
        assert coarse["rgb_map"].shape == (0, 3) and fine["rgb_map"].shape == (0, 3) and fine["z_vals"].shape == (0, 192)
    assert sample_pdf(torch.zeros(0, 63, device=DEV), torch.zeros(0, 62, device=DEV), 128, det=True).shape == (0, 128)
    ro, rd = ops.get_rays(8, 8, 10.0, torch.eye(4, device=DEV)[:3].contiguous(), row0=3, nrows=0)
    assert ro.shape == (0, 8, 3) and rd.shape == (0, 8, 3)
    assert pe(none).shape == (0, 63)
    one = render_rays(m, pe, pv, torch.zeros(1, 3, device=DEV), torch.ones(1, 3, device=DEV), 2.0, 6.0, 2, False)
    assert one["rgb_map"].shape == (1, 3) and torch.isfinite(one["rgb_map"]).all()


# ---------------------------------------------------------------------------
def _margin_mask(cdf, u, tol=1e-6):
    """uniforms farther than tol from every CDF knot (index is then insensitive to 1-ulp CDF noise)"""
    d = (u[:, :, None].double() - cdf[:, None, :].double()).abs().min(-1).values
    return d > tol


def _sample_tol(cdf, bins, inds):
    """Per-sample fp32 tolerance of the interpolation bins_lo + (u - cdf_lo)/denom * (bins_hi - bins_lo):
    one-ulp differences in the CDF knots (the normaliser's summation order is not pinned by the
    reference) are amplified by bin_width / denom."""
    inds = inds.long()
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    denom = cdf.gather(1, above) - cdf.gather(1, below)
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    width = (bins.gather(1, above) - bins.gather(1, below)).abs()
    return 2e-6 + width * 4e-7 / denom


def _check_flipped(cdf, bins, i_ref, i_gpu, s_ref, s_gpu):
    """Samples whose CDF index differs from the reference's (the uniform sits within an ulp of a knot):
    the index moves by one at most, and both samples lie next to that knot's bin edge -- within
    4e-6 * bin_width / delta_cdf of it on whichever side they fall.  (Where delta_cdf < 1e-5 the reference
    itself replaces the denominator by 1, nerf.py:310, and jumps by a bin width there.)  Returns the count."""
    mism = i_ref != i_gpu
    if not mism.any():
        return 0
    nb = cdf.shape[-1]
    i_ref, i_gpu = i_ref.long(), i_gpu.long()
    assert (i_ref - i_gpu).abs()[mism].max() <= 1
    k = (torch.maximum(i_ref, i_gpu) - 1).clamp(0, nb - 1)      # the knot one side counts as <= u and the other not
    lo, hi = (k - 1).clamp(min=0), (k + 1).clamp(max=nb - 1)

    def slack(a, b):
        denom = (cdf.gather(1, b) - cdf.gather(1, a)).double()
        width = (bins.gather(1, b) - bins.gather(1, a)).abs().double()
        return torch.where(denom < 1e-5, width, width * 4e-6 / denom.clamp_min(1e-5))
    bound = 4e-6 + torch.maximum(slack(lo, k), slack(k, hi))
    edge = bins.gather(1, k).double()
    assert ((s_gpu.double() - edge).abs() <= bound)[mism].all(), ((s_gpu.double() - edge).abs() - bound)[mism].max()
    assert ((s_ref.double() - edge).abs() <= bound)[mism].all()
    return int(mism.sum())


def test_sample_pdf_golden(amd, golden, capsys):
    from nerf_amd import ops
    g = golden("f6_sample_pdf")
    bins, w, u = T(g["bins"], DEV), T(g["weights"], DEV), T(g["u"], DEV)
    cdf = T(g["cdf"])
    s, inds = ops.sample_pdf(bins, w, 128, det=False, u=u, return_inds=True)
    ok = _margin_mask(cdf, T(g["u"]))
    assert ok.float().mean() > 0.999
    # indices: EXACT wherever the uniform is not within 1e-6 of a knot; the rest is counted
    mism = (inds.cpu() != T(g["inds_rand"]))
    assert not (mism & ok).any()
    assert mism.sum().item() <= 4
    same = ~mism
    tol = _sample_tol(cdf, T(g["bins"]), T(g["inds_rand"]))
    assert ((s.cpu() - T(g["samples_rand"])).abs() <= tol)[same].all()
    assert (s.cpu() == T(g["samples_rand"])).float().mean() > 0.3      # most samples are bit-identical
    n_flip_rand = _check_flipped(cdf, T(g["bins"]), T(g["inds_rand"]), inds.cpu(), T(g["samples_rand"]), s.cpu())
    s, inds = ops.sample_pdf(bins, w, 128, det=True, return_inds=True)
    udet = torch.linspace(0, 1, 128).expand(256, 128)
    ok = _margin_mask(cdf, udet)
    mism = (inds.cpu() != T(g["inds_det"]))
    # u = 1.0 sits exactly on the last knot of every row (cdf[-1] is 1 up to an ulp), so a 1-ulp difference
    # of that knot flips the index of the LAST deterministic sample; no other column may differ
    assert not (mism & ok).any() and mism.sum().item() <= 64
    assert not mism[:, :-1].any(), "only the u = 1.0 column may differ"
    tol = _sample_tol(cdf, T(g["bins"]), T(g["inds_det"]))
    assert ((s.cpu() - T(g["samples_det"])).abs() <= tol)[~mism].all()
    n_flip_det = _check_flipped(cdf, T(g["bins"]), T(g["inds_det"]), inds.cpu(), T(g["samples_det"]), s.cpu())
    with capsys.disabled():
        print(f"\nsample_pdf golden: indices differing from the reference: {n_flip_rand} of 32768 (injected u), "
              f"{n_flip_det} of 32768 (deterministic, all in the u = 1.0 column); each such sample checked against its "
              f"knot's bin edge")
    # drop-in signature; properties of the reference's tests with the real calling convention
    s2 = amd.sample_pdf(bins, w, 128, det=True)
    assert torch.equal(s2, s)
    assert s.shape == (256, 128)
    assert torch.all(s >= bins.min(-1, keepdim=True).values - 1e-6)
    assert torch.all(s <= bins.max(-1, keepdim=True).values + 1e-6)
    assert torch.all(torch.diff(s[2]) >= -1e-6)
    sr = amd.sample_pdf(bins, w, 128, det=False)
    assert sr.shape == (256, 128) and torch.isfinite(sr).all()
    with pytest.raises(ValueError):
        amd.sample_pdf(bins, T(g["bins"], DEV), 128)      # equal lengths: the reference's own tests' mistake


@pytest.mark.parametrize("nb,ns", [(2, 3), (5, 10), (63, 128), (64, 64), (130, 200), (1000, 77)])
def test_sample_pdf_sizes(amd, nb, ns):
    from nerf_amd import ops
    gen = torch.Generator().manual_seed(nb * 1000 + ns)
    N = 19
    bins, _ = torch.sort(torch.rand(N, nb, generator=gen) * 4 + 2, -1)
    w = torch.rand(N, nb - 1, generator=gen) ** 3
    u = torch.rand(N, ns, generator=gen)
    so, io, cdf = O.sample_pdf(bins, w, ns, u=u, return_inds=True)
    s, inds = ops.sample_pdf(bins.to(DEV), w.to(DEV), ns, u=u.to(DEV), return_inds=True)
    ok = _margin_mask(cdf, u)
    mism = inds.cpu() != io.to(torch.int32)
    assert not (mism & ok).any()
    assert ((s.cpu() - so).abs() <= _sample_tol(cdf, bins, io))[~mism].all()
    _check_flipped(cdf, bins, io.to(torch.int32), inds.cpu(), so, s.cpu())
    # strided views (weights[..., 1:-1] style) take the row-stride path
    wfull = torch.rand(N, nb + 1, generator=gen).to(DEV)
    s_view = ops.sample_pdf(bins.to(DEV), wfull[:, 1:-1], ns, u=u.to(DEV))
    s_cont = ops.sample_pdf(bins.to(DEV), wfull[:, 1:-1].contiguous(), ns, u=u.to(DEV))
    assert torch.equal(s_view, s_cont)


def test_sort_and_merge(amd, golden):
    from nerf_amd import ops
    g = golden("f7_merge")
    zc, zf = T(g["z_coarse"], DEV), T(g["z_fine"], DEV)
    out = ops.sort_rows(zc, zf)
    assert torch.equal(out.cpu(), T(g["z_sorted"]))                       # bit-exact
    for n in (1, 3, 64, 100, 192, 257, 500, 1024):
        x = torch.randn(11, n)
        x[0, : n // 2] = x[0, 0]                                          # duplicates
        assert torch.equal(ops.sort_rows(x.to(DEV)).cpu(), torch.sort(x, -1).values)
    # fused glue == separate kernels (bit-exact) and == oracle (fp32 tolerance)
    gen = torch.Generator().manual_seed(5)
    N = 200
    zc, _ = torch.sort(torch.rand(N, 64, generator=gen) * 4 + 2, -1)
    wc = torch.rand(N, 64, generator=gen) ** 4
    u = torch.rand(N, 128, generator=gen)
    for uu in (u, None):
        det = uu is None
        zm = 0.5 * (zc[:, 1:] + zc[:, :-1])
        sf = ops.sample_pdf(zm.to(DEV), (wc[:, 1:-1] + 1e-5).to(DEV), 128, det=det, u=None if det else uu.to(DEV))
        ref = ops.sort_rows(zc.to(DEV), sf)
        fused, zfine = ops.sample_merge(zc.to(DEV), wc.to(DEV), 128, det, u=None if det else uu.to(DEV),
                                        return_fine=True)
        assert torch.equal(fused, ref) and torch.equal(zfine, sf)
        orc = O.merge_samples(zc, wc, 128, det, uu)
        # a sample whose CDF index differs by one moves by a bin width; compare sorted rows loosely
        # in count, tightly in value
        diff = (fused.cpu() - orc).abs()
        assert (diff > 5e-5).float().mean() < 2e-3 and diff.max() < 0.1
    assert torch.all(torch.diff(fused, dim=-1) >= 0)


# ---------------------------------------------------------------------------
def _raw_tolerance(precision):
    # raw outputs are O(0.1); operand rounding 2^-8 (bf16) / 2^-11 (fp16) through 10 layers
    return {"bf16": 3e-3, "fp16": 4e-4}[precision]


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_mlp_forward_golden(amd, golden, precision):
    g = golden("f3_mlp")
    pts, vd, raw_ref = T(g["pts"]), T(g["viewdirs"]), T(g["raw"])
    model = amd_model(O.init_params(0), precision)
    tol = _raw_tolerance(precision)
    with torch.no_grad():
        # fused mode: a ray per point, z = 0 puts the sample on the origin; rays_d = unit view direction
        raw = model.forward_rays(pts.to(DEV), vd.to(DEV), torch.zeros(pts.shape[0], 1, device=DEV))
        assert raw.shape == (4096, 1, 4)
        err = maxabs(raw.view(-1, 4), raw_ref)
        assert err < tol, err
        # literal NeRF.forward(x_pts_encoded, x_views_encoded) signature
        ep, ev = O.positional_encode(pts, 10), O.positional_encode(vd, 4)
        raw2 = model(ep.to(DEV), ev.to(DEV))
        assert raw2.shape == (4096, 4)
        err2 = maxabs(raw2, raw_ref)
        assert err2 < tol, err2
        # ragged sizes (not a multiple of the 256-sample tile), and a tiny call
        for n in (1, 31, 257, 1000):
            r = model.forward_rays(pts[:n].to(DEV), vd[:n].to(DEV), torch.zeros(n, 1, device=DEV))
            assert maxabs(r.view(-1, 4), raw_ref[:n]) < tol
    with pytest.raises(ValueError):
        model(ep.to(DEV))
    sd = model.state_dict()
    assert list(sd.keys()) == [k for k, _ in O.param_shapes()]


def test_mlp_other_architectures_run(amd, golden):
    """Architectures outside the fused kernels (the reference's own test shape, tests/test_nerf_model.py:105-109:
    D=4, W=64, skips=[2]) run on the layer-program kernels in the requested operand type (test_gpu_p16.py,
    test_gpu_fp32.py); beyond those kernels' widths the layers are library GEMMs in fp32, and the model says so."""
    g = golden("f3_mlp")
    m = amd_model(O.init_params(2, D=4, W=64, skips=(2,)), "fp32", D=4, W=64, skips=[2])
    assert m.pts_linears[3].in_features == 64 + 63
    assert not m.fused() and m.effective_precision() == "fp32"
    ep = O.positional_encode(T(g["pts"])[:512], 10).to(DEV)
    ev = O.positional_encode(T(g["viewdirs"])[:512], 4).to(DEV)
    with torch.no_grad():
        assert maxabs(m(ep, ev), T(g["raw_mini"])) < 2e-5
    with pytest.raises(NotImplementedError):
        m.forward_rays(torch.zeros(8, 3, device=DEV), torch.zeros(8, 3, device=DEV), torch.zeros(8, 4, device=DEV))
    assert amd_model(O.init_params(2, D=4, W=64, skips=(2,)), "bf16", D=4, W=64, skips=[2]).effective_precision() == "bf16"
    wide = amd.NeRF(D=2, W=320, skips=[], precision="bf16").to(DEV)
    assert not wide.program_supported() and wide.effective_precision() == "fp32"
    with torch.no_grad():
        x, v = torch.randn(10, 63, device=DEV), torch.randn(10, 27, device=DEV)
        assert maxabs(wide(x, v), wide._forward_library(x, v, None)) == 0.0


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_render_e2e_golden(amd, golden, precision):
    """Config 1 (1024 rays, 64+128, deterministic) against the reference's maps.

    Two ill-conditioned spots of the pipeline at random init are handled explicitly:
    (1) the last interval is 1e10 long, so alpha_last is a step function of raw_sigma_last, and
        1-2 % of rays have |raw_sigma_last| below the operand rounding error: maps are compared
        on the rays whose reference |raw_sigma_last| exceeds 10x the mode's observed raw error,
        and the masked share is bounded;
    (2) the fine sample positions are an inverse CDF of the coarse weights, which amplifies their
        1e-5 differences to ~1e-2 in z and ~1e-2 in the fine DEPTH (an fp32 evaluation at the
        shifted positions moves by the same amount).  So the fine pass is checked at 1e-3 stage-wise:
        on the reference's own sample positions, and against the fp32 oracle evaluated at this
        implementation's positions; end to end rgb/acc hold 1e-3 and depth 2e-2."""
    g = golden("f8_e2e_render")
    ro, rd = T(g["rays_o"], DEV), T(g["rays_d"], DEV)
    mc, mf = amd_model(O.init_params(0), precision), amd_model(O.init_params(1), precision)
    pe, pv = amd.PositionalEncoder(3, 10).to(DEV), amd.PositionalEncoder(3, 4).to(DEV)
    delta = {"bf16": 3e-3, "fp16": 4e-4}[precision]   # 10x the observed raw-sigma error of the mode
    for lind in (0, 1):
        for wb in (0, 1):
            tag = f"l{lind}w{wb}"
            with torch.no_grad():
                c, f = amd.render_hierarchical(mc, mf, pe, pv, ro, rd, 2.0, 6.0, 64, 128, rand=False, det=True,
                                               lindisp=bool(lind), white_bkgd=bool(wb))
            assert set(c.keys()) == {"rgb_map", "depth_map", "disp_map", "acc_map", "weights", "z_vals", "alpha",
                                     "raw_sigma"}
            assert f["z_vals"].shape == (1024, 192) and f["raw_sigma"].shape == (1024, 192)
            okc = torch.from_numpy(np.abs(g[f"c_sigma_last_{tag}"]) > delta)
            okf = okc & torch.from_numpy(np.abs(g[f"f_sigma_last_{tag}"]) > delta)
            assert okf.float().mean() > (0.8 if precision == "bf16" else 0.95)
            for k in ("rgb_map", "depth_map", "acc_map"):
                ec = maxabs(c[k].cpu()[okc], T(g[f"c_{k}_{tag}"])[okc])
                ef = maxabs(f[k].cpu()[okf], T(g[f"f_{k}_{tag}"])[okf])
                assert ec < 1e-3, (k, tag, ec)
                assert ef < (2e-2 if k == "depth_map" else 1e-3), (k, tag, ef)
            # stage-wise: fp32 oracle at this implementation's sample positions
            of = O.render_rays(O.init_params(1), ro.cpu(), rd.cpu(), 2.0, 6.0, 192, white_bkgd=bool(wb),
                               z_vals_override=f["z_vals"].cpu())
            oks = torch.abs(of["raw_sigma"][:, -1]) > delta
            for k in ("rgb_map", "depth_map", "acc_map"):
                assert maxabs(f[k].cpu()[oks], of[k][oks]) < 1e-3, (k, tag)
            if tag == "l0w0":   # fine pass on the reference's own sample positions
                with torch.no_grad():
                    f2 = amd.render_rays(mf, pe, pv, ro, rd, 2.0, 6.0, 192, z_vals_override=T(g["z_comb_l0w0"], DEV))
                assert f2["z_vals"].data_ptr() != 0
                for k in ("rgb_map", "depth_map", "acc_map"):
                    assert maxabs(f2[k].cpu()[okf], T(g[f"f_{k}_{tag}"])[okf]) < 1e-3
                raw_tol = {"bf16": 1e-3, "fp16": 1.5e-4}[precision]
                assert maxabs(f2["raw_sigma"], T(g["f_raw_sigma_l0w0"])) < raw_tol
                assert maxabs(c["raw_sigma"], T(g["c_raw_sigma_l0w0"])) < raw_tol


def test_mse_pair_matches_torch(amd):
    """loss of train_nerf.py:299, :330-331 and its gradients against the ATen spelling"""
    from nerf_amd import ops
    gen = torch.Generator().manual_seed(0)
    for n in (1, 257, 1024, 5000):
        a0, b0, t = (torch.rand(n, 3, generator=gen).to(DEV) for _ in range(3))
        a1, b1 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        want = torch.mean((a1 - t) ** 2) + torch.mean((b1 - t) ** 2)
        (want * 3.0).backward()
        a2, b2 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        out = torch.zeros(3, device=DEV)
        total, la, lb = ops.mse_pair(a2, b2, t, out=out)
        (total * 3.0).backward()
        assert abs(total.item() - want.item()) < 1e-6 * want.item()
        assert abs(la.item() - torch.mean((a0 - t) ** 2).item()) < 1e-6 and abs(out[2].item() - lb.item()) == 0.0
        assert not la.requires_grad and total.requires_grad
        assert torch.allclose(a2.grad, a1.grad, rtol=1e-6, atol=1e-12)
        assert torch.allclose(b2.grad, b1.grad, rtol=1e-6, atol=1e-12)
    with torch.no_grad():
        total, _, _ = ops.mse_pair(a0, b0, t)
        assert not total.requires_grad
    with pytest.raises(ValueError):
        ops.mse_pair(a0, b0[:-1], t)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_mlp_inference_kernel_matches_training_forward(amd, precision):
    """The no_grad forward (16x16x32 kernel) and the forward that keeps activations (32x32x16 kernel) are
    two implementations of nerf.py:173-209 with the same rounding points; ragged sample counts exercise
    partial column tiles, partial wave tiles and partial workgroup tiles."""
    tol = 2e-3 if precision == "bf16" else 2.5e-4
    m = amd_model(O.init_params(0), precision)
    gen = torch.Generator().manual_seed(5)
    for n_rays, S in ((1, 1), (1, 15), (3, 17), (5, 51), (2, 129), (7, 192), (1000, 3)):
        ro = (torch.rand(n_rays, 3, generator=gen) * 4 - 2).to(DEV)
        rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1).to(DEV) * 1.3
        z = (torch.rand(n_rays, S, generator=gen) * 4 + 2).to(DEV)
        with torch.no_grad():
            inf = m.forward_rays(ro, rd, z)
        trn = m.forward_rays(ro, rd, z)
        assert trn.requires_grad and not inf.requires_grad and inf.shape == (n_rays, S, 4)
        assert maxabs(inf, trn) < tol, (n_rays, S, maxabs(inf, trn))
        # and both against the oracle's fp32 network on the same points
        pts = (ro[:, None, :] + rd[:, None, :] * z[..., None]).reshape(-1, 3).cpu()
        vd = (rd / rd.norm(dim=-1, keepdim=True))[:, None, :].expand(n_rays, S, 3).reshape(-1, 3).cpu()
        ref = O.mlp_forward(O.init_params(0), O.positional_encode(pts, 10), O.positional_encode(vd, 4))
        assert maxabs(inf.reshape(-1, 4), ref) < (3e-3 if precision == "bf16" else 4e-4)
    # the literal forward on encoded inputs, with and without autograd
    pts = torch.rand(777, 3, generator=gen) * 6 - 3
    vd = torch.nn.functional.normalize(torch.randn(777, 3, generator=gen), dim=-1)
    ep, ev = O.positional_encode(pts, 10).to(DEV), O.positional_encode(vd, 4).to(DEV)
    with torch.no_grad():
        a = m(ep, ev)
    b = m(ep, ev)
    fused = m.forward_rays(pts.to(DEV), vd.to(DEV), torch.zeros(777, 1, device=DEV))
    assert maxabs(a, b) == 0.0 and b.requires_grad
    assert maxabs(a, fused.reshape(-1, 4)) < tol
