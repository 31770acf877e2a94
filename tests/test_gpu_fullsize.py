"""GPU tests at the benchmark's full sizes (800x800 frames, 65 536-ray chunks, 4096-ray training batches),
where the CPU oracle would take minutes: size-independent properties instead of element-wise comparison.

* per-ray / per-sample kernels must give bit-identical results for a ray whether it is processed in one
  launch over everything or in a launch over a slice (the ray -> wave / workgroup mapping, persistent
  grids and tail handling differ between the two);
* compositing identities of src/nerf_project/models/nerf.py:430-472 (weights sum to acc, alpha in [0, 1],
  white background adds 1 - acc);
* sortedness and multiset preservation of the coarse + fine depths (scripts/train_nerf.py:320);
* linearity of the weight gradients in the batch: one 4096-ray backward equals the sum of four 1024-ray
  backwards.
"""
import pytest
import torch

from util import O, amd_model, maxabs

pytestmark = pytest.mark.gpu

DEV = "cuda"


def test_get_rays_full_frame_blocks_and_norms():
    from nerf_amd import ops
    H = W = 800
    c2w = O.pose_spherical(30.0, -30.0, 4.0).to(DEV)
    ro, rd = ops.get_rays(H, W, 1111.1, c2w)
    assert ro.shape == (H, W, 3) and rd.shape == (H, W, 3)
    assert (rd.norm(dim=-1) - 1.0).abs().max().item() < 2e-7 * 4
    assert torch.equal(ro.reshape(-1, 3), c2w[:3, 3].expand(H * W, 3))
    # row blocks (what a rank of a sharded render generates), including blocks that are not multiples of
    # the 256-pixel wave tile, reproduce the full frame bit for bit
    row = 0
    for nrows in (1, 7, 100, 333, 359):
        bo, bd = ops.get_rays(H, W, 1111.1, c2w, row0=row, nrows=nrows)
        assert torch.equal(bd.reshape(-1, 3), rd[row:row + nrows].reshape(-1, 3))
        assert torch.equal(bo.reshape(-1, 3), ro[row:row + nrows].reshape(-1, 3))
        row += nrows
    assert row == H
    # odd widths: pixel groups of four straddle rows
    ro2, rd2 = ops.get_rays(37, 803, 500.0, c2w)
    bo, bd = ops.get_rays(37, 803, 500.0, c2w, row0=5, nrows=11)
    assert torch.equal(bd.reshape(-1, 3), rd2[5:16].reshape(-1, 3))
    i, j = 21, 777                                    # one pixel by hand (ray_utils.py:53-62)
    d = torch.tensor([(j - 803 * 0.5) / 500.0, (i - 37 * 0.5) / 500.0, -1.0])
    w = (c2w[:3, :3].cpu() * d).sum(-1)
    assert maxabs(rd2[i, j], w / w.norm()) < 5e-7


@pytest.mark.parametrize("S", [64, 192])
def test_composite_full_frame_identities_and_slices(S):
    from nerf_amd import ops
    n = 640000 if S == 64 else 200000
    gen = torch.Generator(device=DEV).manual_seed(S)
    raw = torch.randn(n, S, 4, device=DEV, generator=gen)
    raw[..., 3] = raw[..., 3] * 4.0
    z = 2.0 + torch.cumsum(torch.rand(n, S, device=DEV, generator=gen), dim=-1) * (4.0 / S)
    with torch.no_grad():
        rgb, depth, disp, acc, w, alpha = ops.composite(raw, z, False)
        rgb_w = ops.composite(raw, z, True)[0]
    assert alpha.min().item() >= 0.0 and alpha.max().item() <= 1.0
    assert w.min().item() >= 0.0
    assert (w.sum(-1) - acc).abs().max().item() < 2e-6 * S ** 0.5 + 2e-6
    assert acc.max().item() <= 1.0 + 1e-5
    assert (rgb_w - (rgb + (1.0 - acc)[:, None])).abs().max().item() < 1e-6
    assert ((w * z).sum(-1) - depth).abs().max().item() < 1e-4
    assert rgb.min().item() >= 0.0 and rgb.max().item() <= 1.0 + 1e-5
    # the last sample's interval is 1e10 long: alpha is exactly 1 wherever its density is positive
    last = raw[:, -1, 3] > 1e-6
    assert torch.equal(alpha[:, -1][last], torch.ones_like(alpha[:, -1][last]))
    # a ray's results do not depend on which launch, wave or loop round handled it
    for lo, cnt in ((0, 1), (12345, 4096), (n - 70001, 70001)):
        with torch.no_grad():
            part = ops.composite(raw[lo:lo + cnt], z[lo:lo + cnt], False)
        for full, p in zip((rgb, depth, disp, acc, w, alpha), part):
            assert torch.equal(full[lo:lo + cnt], p)


def test_sample_merge_chunk_properties():
    from nerf_amd import ops
    n, Sc, Nf = 65536, 64, 128
    gen = torch.Generator(device=DEV).manual_seed(3)
    zc = 2.0 + torch.cumsum(torch.rand(n, Sc, device=DEV, generator=gen), dim=-1) * (4.0 / Sc)
    w = torch.rand(n, Sc, device=DEV, generator=gen) ** 4
    w[::7] = 0.0                                        # rays that hit nothing: uniform pdf
    for det in (True, False):
        u = None if det else torch.rand(n, Nf, device=DEV, generator=gen)
        z_all = ops.sample_merge(zc, w, Nf, det, u=u)
        assert z_all.shape == (n, Sc + Nf)
        assert (z_all[:, 1:] >= z_all[:, :-1]).all().item()                 # sorted
        mids = 0.5 * (zc[:, 1:] + zc[:, :-1])
        # multiset: removing the coarse depths leaves Nf values inside the span of the bin edges
        z_fine = ops.sample_pdf(mids, w[:, 1:-1] + 1e-5, Nf, det=det, u=u)
        ref = torch.sort(torch.cat([zc, z_fine], dim=-1), dim=-1).values
        assert torch.equal(z_all, ref)
        assert (z_fine >= mids[:, :1] - 1e-6).all().item() and (z_fine <= mids[:, -1:] + 1e-6).all().item()
        part = ops.sample_merge(zc[999:5000], w[999:5000], Nf, det, u=None if u is None else u[999:5000])
        assert torch.equal(part, z_all[999:5000])
    # deterministic mode merges two ascending lists instead of sorting; inputs that are not ascending, and
    # inputs full of ties between and inside the lists, must still come out as torch.sort orders them
    def reference(zc_, w_):
        mids_ = 0.5 * (zc_[:, 1:] + zc_[:, :-1])
        zf = ops.sample_pdf(mids_, w_[:, 1:-1] + 1e-5, Nf, det=True)
        return torch.sort(torch.cat([zc_, zf], dim=-1), dim=-1).values
    zq = torch.round(zc[:4096] * 4.0) / 4.0                     # 17 distinct depths per ray: ties everywhere
    assert torch.equal(ops.sample_merge(zq, w[:4096], Nf, True), reference(zq, w[:4096]))
    zr = zc[:4096].flip(-1).contiguous()                        # descending coarse depths
    assert torch.equal(ops.sample_merge(zr, w[:4096], Nf, True), reference(zr, w[:4096]))
    zs = zc[:4096].clone()
    zs[:, 40] = zs[:, 10]                                       # one element out of order
    assert torch.equal(ops.sample_merge(zs, w[:4096], Nf, True), reference(zs, w[:4096]))


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_mlp_chunk_65536_matches_slices(precision):
    m = amd_model(O.init_params(0), precision)
    n, S = 65536, 192
    gen = torch.Generator(device=DEV).manual_seed(11)
    ro = torch.rand(n, 3, device=DEV, generator=gen) * 2 - 1
    rd = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=gen), dim=-1)
    z = 2.0 + torch.cumsum(torch.rand(n, S, device=DEV, generator=gen), dim=-1) * (4.0 / S)
    with torch.no_grad():
        full = m.forward_rays(ro, rd, z)
        assert torch.isfinite(full).all().item()
        for lo, cnt in ((0, 1), (4096, 4096), (n - 1001, 1001)):
            part = m.forward_rays(ro[lo:lo + cnt], rd[lo:lo + cnt], z[lo:lo + cnt])
            assert torch.equal(part, full[lo:lo + cnt])
        # against the fp32 oracle on a sample of rays
        idx = torch.randint(0, n, (8,), generator=torch.Generator().manual_seed(1))
        pts = (ro[idx, None, :] + rd[idx, None, :] * z[idx][..., None]).reshape(-1, 3).cpu()
        vd = rd[idx, None, :].expand(8, S, 3).reshape(-1, 3).cpu()
        ref = O.mlp_forward(O.init_params(0), O.positional_encode(pts, 10), O.positional_encode(vd, 4))
    assert maxabs(full[idx].reshape(-1, 4), ref) < (3e-3 if precision == "bf16" else 4e-4)


def test_weight_gradients_are_additive_over_the_batch():
    """4096 rays x 64 samples in one backward against four backwards of 1024 rays accumulated into .grad."""
    n, S = 4096, 64
    gen = torch.Generator(device=DEV).manual_seed(2)
    ro = torch.rand(n, 3, device=DEV, generator=gen) * 2 - 1
    rd = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=gen), dim=-1)
    z = 2.0 + torch.cumsum(torch.rand(n, S, device=DEV, generator=gen), dim=-1) * (4.0 / S)
    cot = torch.randn(n, S, 4, device=DEV, generator=gen) * 1e-4
    a, b = amd_model(O.init_params(0), "bf16"), amd_model(O.init_params(0), "bf16")
    (a.forward_rays(ro, rd, z) * cot).sum().backward()
    for q in range(4):
        sl = slice(1024 * q, 1024 * (q + 1))
        (b.forward_rays(ro[sl], rd[sl], z[sl]) * cot[sl]).sum().backward()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        rel = ((pa.grad - pb.grad).norm() / (pa.grad.norm() + 1e-30)).item()
        assert rel < 1e-5, (k, rel)          # same products, fp32 sums in a different order
