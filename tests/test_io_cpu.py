"""CPU tests of the file formats either side of the hot path: PNG IO, the checkpoint wire format of
scripts/train_nerf.py:118-131, and the loader oracle's blend arithmetic."""
import os

import numpy as np
import pytest
import torch

from util import O  # noqa: F401  (puts the repo root on sys.path)
from oracle import blender_oracle as BO


def test_png_round_trip(tmp_path):
    from nerf_amd.utils import png
    rng = np.random.default_rng(0)
    for shape in ((7, 5, 4), (6, 9, 3), (4, 4, 1), (3, 8)):
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        p = str(tmp_path / f"a{len(shape)}_{shape[-1]}.png")
        png.write_png(p, a)
        want = a if a.ndim == 3 else a[..., None]
        with open(p, "rb") as f:
            assert np.array_equal(png._decode(f.read()), want)          # built-in decoder
        assert np.array_equal(png.read_png(p), want)                    # Pillow when present
    with pytest.raises(TypeError):
        png.write_png(str(tmp_path / "x.png"), np.zeros((2, 2, 3), dtype=np.float32))
    with pytest.raises(ValueError):
        png._decode(b"not a png file at all")


def test_png_decoder_filters(tmp_path):
    """Pillow writes adaptive row filters (Sub/Up/Average/Paeth): the built-in decoder must undo them."""
    Image = pytest.importorskip("PIL.Image")
    from nerf_amd.utils import png
    yy, xx = np.mgrid[0:33, 0:41]
    a = np.stack([(3 * xx + yy) % 256, (xx * yy) % 256, (5 * yy) % 256, (xx + 2 * yy) % 256], -1).astype(np.uint8)
    a[5:9] = np.random.default_rng(1).integers(0, 256, size=(4, 41, 4), dtype=np.uint8)
    for mode, arr in (("RGBA", a), ("RGB", a[..., :3])):
        p = str(tmp_path / f"{mode}.png")
        Image.fromarray(arr, mode).save(p, optimize=True)
        with open(p, "rb") as f:
            raw = f.read()
        assert np.array_equal(png._decode(raw), arr)


def test_blend_oracle_values():
    px = np.array([[[255, 0, 51, 255], [255, 0, 51, 0], [200, 100, 50, 128]]], dtype=np.uint8)
    b = BO.blend_images(px, white=False)
    w = BO.blend_images(px, white=True)
    f = np.float32
    assert np.array_equal(b[0, 0], np.array([1.0, 0.0, 0.2], dtype=f)) and np.array_equal(w[0, 0], b[0, 0])
    assert np.array_equal(b[0, 1], np.zeros(3, dtype=f)) and np.array_equal(w[0, 1], np.ones(3, dtype=f))
    a = f(128 / 255.0)
    assert np.array_equal(b[0, 2], np.array([f(200 / 255.0) * a, f(100 / 255.0) * a, f(50 / 255.0) * a], dtype=f))
    assert np.array_equal(w[0, 2], b[0, 2] + (f(1.0) - a))
    rgb = BO.blend_images(px[..., :3], white=True)           # no alpha channel: scaled only
    assert rgb.shape == (1, 3, 3) and rgb[0, 2, 0] == f(200 / 255.0)


def test_checkpoint_wire_format(tmp_path):
    """keys and types of train_nerf.py:118-131; readable with weights_only=True; loads into
    torch.optim.Adam and into fresh modules; build_models follows the saved args (render_nerf.py:188-252)."""
    from nerf_amd import checkpoint as C
    from nerf_amd.models.nerf import NeRF
    mc, mf = NeRF(precision="fp32"), NeRF(precision="fp32")
    mc.load_state_dict(O.init_params(0)); mf.load_state_dict(O.init_params(1))
    params = list(mc.parameters()) + list(mf.parameters())
    opt = torch.optim.Adam(params, lr=5e-4)
    for p in params:
        p.grad = torch.full_like(p, 1e-3)
    opt.step()
    args = {"expname": "t", "net_depth": 8, "net_width": 256, "input_ch_pts_L": 10, "input_ch_views_L": 4,
            "use_viewdirs": True, "use_hierarchical": True, "near": 2.0, "far": 6.0, "lr": 5e-4, "config": None}
    path = str(tmp_path / "ck" / C.checkpoint_name(3, 1234))
    assert path.endswith("checkpoint_epoch_0003_step_0001234.pth")
    assert C.checkpoint_name(9, 5, True) == "checkpoint_final_epoch_0009_step_0000005.pth"
    C.save_checkpoint(path, epoch=3, global_step=1234, args=args, H=800, W=800, focal=1111.1, model_coarse=mc,
                      model_fine=mf, optimizer=opt)
    ck = C.load_checkpoint(path)
    assert set(ck) == {"epoch", "global_step", "args_saved_at_checkpoint", "H", "W", "focal",
                       "model_coarse_state_dict", "optimizer_state_dict", "model_fine_state_dict"}
    assert (ck["epoch"], ck["global_step"], ck["H"], ck["W"]) == (3, 1234, 800, 800)
    assert list(ck["model_coarse_state_dict"]) == [k for k, _ in O.param_shapes()]
    opt2 = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in params], lr=1.0)
    opt2.load_state_dict(ck["optimizer_state_dict"])
    assert opt2.param_groups[0]["lr"] == 5e-4 and len(opt2.state) == len(params)
    m1, m2, pe, pv, cfg = C.build_models(ck, device="cpu", precision="fp32")
    assert m2 is not None and pe.output_dims == 63 and pv.output_dims == 27 and cfg["net_depth"] == 8
    assert all(torch.equal(a, b) for a, b in zip(m1.state_dict().values(), mc.state_dict().values()))
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), mf.state_dict().values()))
    # coarse-only, view-independent run (the script's defaults): no fine model, no view encoder
    nv = NeRF(input_ch_views=0, use_viewdirs=False, precision="fp32")
    p2 = str(tmp_path / "nv.pth")
    C.save_checkpoint(p2, epoch=0, global_step=1, args={"use_viewdirs": False, "use_hierarchical": False}, H=4, W=4,
                      focal=5.0, model_coarse=nv, optimizer=torch.optim.Adam(nv.parameters()))
    m1, m2, pe, pv, cfg = C.build_models(C.load_checkpoint(p2), device="cpu")
    # ... which the fused 16-bit kernels cover; other depths / widths keep the operand type (layer-program kernels)
    # up to width 256, and run as fp32 library GEMMs beyond
    assert m2 is None and pv is None and not m1.use_viewdirs and cfg["precision"] == "bf16"
    small = NeRF(D=4, W=64, skips=[2], input_ch_views=0, use_viewdirs=False, precision="fp32")
    p3 = str(tmp_path / "small.pth")
    C.save_checkpoint(p3, epoch=0, global_step=1, args={"use_viewdirs": False, "use_hierarchical": False, "net_depth": 4,
                                                        "net_width": 64}, H=4, W=4, focal=5.0, model_coarse=small,
                      optimizer=torch.optim.Adam(small.parameters()))
    assert C.build_models(C.load_checkpoint(p3), device="cpu")[4]["precision"] == "bf16"
    assert C.build_models(C.load_checkpoint(p3), device="cpu", precision="fp32")[4]["precision"] == "fp32"
    wide = NeRF(D=2, W=320, skips=[1], input_ch_views=0, use_viewdirs=False, precision="bf16")
    p4 = str(tmp_path / "wide.pth")
    C.save_checkpoint(p4, epoch=0, global_step=1, args={"use_viewdirs": False, "use_hierarchical": False, "net_depth": 2,
                                                        "net_width": 320}, H=4, W=4, focal=5.0, model_coarse=wide,
                      optimizer=torch.optim.Adam(wide.parameters()))
    assert C.build_models(C.load_checkpoint(p4), device="cpu")[4]["precision"] == "fp32"
    torch.save({"something": 1}, str(tmp_path / "other.pth"))
    with pytest.raises(ValueError):
        C.load_checkpoint(str(tmp_path / "other.pth"))


def test_learning_rate_schedule():
    from nerf_amd.trainer import TrainConfig, learning_rate, mse_to_psnr
    cfg = TrainConfig()
    assert learning_rate(cfg, 0) == 5e-4
    assert abs(learning_rate(cfg, 250000) - 5e-5) < 1e-12          # train_nerf.py:288
    assert abs(learning_rate(cfg, 125000) - 5e-4 * 0.1 ** 0.5) < 1e-12
    assert abs(mse_to_psnr(0.01) - 20.0) < 1e-9


def test_pose_spherical_known_answers():
    """scripts/render_nerf.py:72-124 cannot be imported here (configargparse is absent), so the product's and
    the oracle's pose_spherical are pinned by values worked out by hand from those lines:
    c2w = shift_y(h) @ R_y(theta) @ R_x(phi) @ T_z(r)."""
    import math
    from util import O
    from nerf_amd.utils.synthetic import camera_path, pose_spherical
    for fn in (pose_spherical, O.pose_spherical):
        # theta = phi = 0: the identity rotation, camera at (0, 0, r)
        p = fn(0.0, 0.0, 4.0)
        assert torch.allclose(p, torch.tensor([[1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0], [0, 0, 0, 1]]), atol=1e-7)
        # theta = 90: R_y maps z to x, so the camera sits at (r, 0, 0) and its x axis points along -z
        p = fn(90.0, 0.0, 4.0)
        want = torch.tensor([[0.0, 0, 1, 4.0], [0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 0, 1]])
        assert torch.allclose(p, want, atol=1e-6)
        # phi = -90: R_x(-90) maps z to -y... row 1 = (0, cos, -sin) = (0, 0, 1): camera at (0, r, 0)
        p = fn(0.0, -90.0, 4.0)
        want = torch.tensor([[1.0, 0, 0, 0], [0, 0, 1, 4.0], [0, -1, 0, 0], [0, 0, 0, 1]])
        assert torch.allclose(p, want, atol=1e-6)
        # the benchmark's elevation with a vertical shift: position r (0, sin 30, cos 30) + (0, h, 0)
        p = fn(0.0, -30.0, 4.0, 0.5)
        assert torch.allclose(p[:3, 3], torch.tensor([0.0, 4 * 0.5 + 0.5, 4 * math.cos(math.radians(30))]), atol=1e-6)
        assert p.dtype == torch.float32 and torch.allclose(p[:3, :3] @ p[:3, :3].T, torch.eye(3), atol=1e-6)
    path = camera_path(120)                                            # render_nerf.py:145-149
    assert path.shape == (120, 4, 4)
    assert torch.allclose(path[30], O.pose_spherical(90.0, -30.0, 4.0), atol=1e-6)
    assert torch.equal(path[0], pose_spherical(0.0, -30.0, 4.0))


def test_signatures_match_the_reference():
    """positional parameters (names, order, defaults) of the five boundary symbols against the strings
    inspect.signature gave for the reference's own objects (oracle/gen_golden.py -> meta.json); the product
    may add keyword-only parameters and constructor keywords with defaults after the reference's."""
    import inspect
    import json
    import re
    import nerf_amd
    from util import ROOT
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "meta.json")))["signatures"]

    def ref_params(sig):
        inner = sig[sig.index("(") + 1: sig.rindex(")")]
        out, depth, cur = [], 0, ""
        for ch in inner:                       # split on top-level commas
            depth += ch in "[(" ; depth -= ch in "])"
            if ch == "," and depth == 0:
                out.append(cur); cur = ""
            else:
                cur += ch
        out.append(cur)
        res = []
        for p in out:
            name = re.split(r"[:=]", p.strip())[0].strip()
            default = p.split("=")[-1].strip() if "=" in p else None
            res.append((name, default))
        return res

    objs = {"PositionalEncoder.__init__": nerf_amd.PositionalEncoder.__init__,
            "PositionalEncoder.forward": nerf_amd.PositionalEncoder.forward,
            "NeRF.__init__": nerf_amd.NeRF.__init__, "NeRF.forward": nerf_amd.NeRF.forward,
            "sample_pdf": nerf_amd.sample_pdf, "render_rays": nerf_amd.render_rays, "get_rays": nerf_amd.get_rays}
    for name, obj in objs.items():
        want = ref_params(meta[name])
        got = [(p.name, None if p.default is inspect.Parameter.empty else repr(p.default))
               for p in inspect.signature(obj).parameters.values()
               if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        assert [g[0] for g in got[:len(want)]] == [w[0] for w in want], (name, got, want)
        for (gn, gd), (wn, wd) in zip(got, want):
            assert (gd is None) == (wd is None) and (gd is None or gd == wd), (name, gn, gd, wd)
        for extra in got[len(want):]:
            assert extra[1] is not None, (name, "added positional parameter without a default", extra)


@pytest.mark.parametrize("name", ["f10_ckpt_hier_views", "f10_ckpt_default_flags"])
def test_reference_checkpoint_fixture_loads_and_oracle_renders_it(name):
    """The checkpoint fixtures were written by the reference's NeRF class and torch.optim.Adam in the layout of
    scripts/train_nerf.py:118-131.  Here (CPU): the safe loader accepts them, the dict has the reference's keys,
    and the oracle renders the stored weights to the maps the reference rendered."""
    import numpy as np
    from util import O, ROOT, T
    from nerf_amd import checkpoint as C
    gdir = os.path.join(ROOT, "tests", "golden")
    ck = C.load_checkpoint(os.path.join(gdir, name + ".pth"))
    g = dict(np.load(os.path.join(gdir, name + "_render.npz")))
    a = ck["args_saved_at_checkpoint"]
    hier, views = a["use_hierarchical"], a["use_viewdirs"]
    want = {"epoch", "global_step", "args_saved_at_checkpoint", "H", "W", "focal", "model_coarse_state_dict",
            "optimizer_state_dict"} | ({"model_fine_state_dict"} if hier else set())
    assert set(ck.keys()) == want
    assert set(a.keys()) >= {"expname", "basedir", "dataset_path", "near", "far", "net_depth", "use_viewdirs", "lr"}
    arch = dict(D=a["net_depth"], skips=(a["net_depth"] // 2,), use_viewdirs=views, L_pts=a["input_ch_pts_L"],
                L_views=a["input_ch_views_L"])
    shapes = O.param_shapes(D=a["net_depth"], W=a["net_width"], input_ch_pts=3 + 6 * a["input_ch_pts_L"],
                            input_ch_views=3 + 6 * a["input_ch_views_L"] if views else 0, skips=arch["skips"],
                            use_viewdirs=views)
    assert [(k, tuple(v.shape)) for k, v in ck["model_coarse_state_dict"].items()] == shapes
    n_par = len(shapes) * (2 if hier else 1)
    assert len(ck["optimizer_state_dict"]["state"]) == n_par
    assert float(ck["optimizer_state_dict"]["state"][0]["step"]) == 2.0
    ro, rd = T(g["rays_o"]), T(g["rays_d"])
    nc, nf = a["N_samples_coarse"], a["N_samples_fine"]
    with torch.no_grad():
        if hier:
            c, f = O.render_hierarchical(ck["model_coarse_state_dict"], ck["model_fine_state_dict"], ro, rd, 2.0, 6.0,
                                         nc, nf, det=True, **arch)
        else:
            c, f = O.render_rays(ck["model_coarse_state_dict"], ro, rd, 2.0, 6.0, nc, **arch), None
    for k in ("rgb_map", "depth_map", "acc_map"):
        assert (c[k] - T(g[f"c_{k}"])).abs().max() < 1e-5
        if f is not None:
            assert (f[k] - T(g[f"f_{k}"])).abs().max() < 1e-4
    # build_models on the CPU: the architecture of render_nerf.py:188-252
    mc, mf, pe, pv, cfg = C.build_models(ck, device="cpu")
    assert (mf is not None) == hier and (pv is not None) == views and cfg["precision"] == "bf16"
    assert C.build_models(ck, device="cpu", precision="fp32")[4]["precision"] == "fp32"


def test_wellconditioned_fixture_matches_oracle(golden):
    """F8b (sigma_linear.bias += 1): the oracle reproduces the reference's maps on every ray."""
    from util import O, T
    g, e = golden("f8b_e2e_wellcond"), golden("f8_e2e_render")
    pc, pf = O.init_params(0), O.init_params(1)
    pc["sigma_linear.bias"] = pc["sigma_linear.bias"] + 1.0
    pf["sigma_linear.bias"] = pf["sigma_linear.bias"] + 1.0
    with torch.no_grad():
        c, f = O.render_hierarchical(pc, pf, T(e["rays_o"]), T(e["rays_d"]), 2.0, 6.0, 64, 128, det=True, lindisp=True,
                                     white_bkgd=True)
    for k in ("rgb_map", "depth_map", "acc_map"):
        assert (c[k] - T(g[f"c_{k}_l1w1"])).abs().max() < 1e-4
        assert (f[k] - T(g[f"f_{k}_l1w1"])).abs().max() < 1e-3


def test_get_rays_host_branch_matches_reference_golden(golden):
    """A pose in host memory (the reference's DataLoader workers, datasets/blender.py:151) takes the package's
    torch-op branch: bit-identical to the rays the reference itself produced (golden f1; 1e-7 allowed on the
    directions for CPUs whose vectorised kernels round differently), same shapes, device and contiguity, the
    reference's own known answers (tests/test_ray_utils.py:56-143), and a row block equals the slice."""
    from util import O, T
    from nerf_amd.utils.ray_utils import get_rays
    g = golden("f1_get_rays")
    for nm in ("identity", "sph40"):
        c2w = T(g[nm + "_c2w"])
        ro, rd = get_rays(8, 8, 5.0, c2w)
        assert ro.device.type == "cpu" and ro.dtype == torch.float32 and ro.is_contiguous() and rd.is_contiguous()
        assert ro.shape == rd.shape == (8, 8, 3)
        assert torch.equal(ro, T(g[nm + "_o8"]))
        assert (rd - T(g[nm + "_d8"])).abs().max() <= 1e-7
        oro, ord_ = O.get_rays(8, 8, 5.0, c2w)
        assert torch.equal(rd, ord_) and torch.equal(ro, oro)        # same op order as the restatement, same machine
        ro, rd = get_rays(800, 800, float(g["focal800"]), c2w[:3])   # [3,4] pose
        assert (rd[392:408, 792:800] - T(g[nm + "_d800_crop"])).abs().max() <= 1e-7
        assert (rd[:4, :4] - T(g[nm + "_d800_corner"])).abs().max() <= 1e-7
        rb_o, rb_d = get_rays(800, 800, float(g["focal800"]), c2w, row0=300, nrows=17)
        assert torch.equal(rb_d, rd[300:317]) and torch.equal(rb_o, ro[300:317])
    ro, rd = get_rays(4, 4, 5.0, torch.eye(4))
    assert torch.equal(ro, torch.zeros(4, 4, 3)) and torch.allclose(rd[2, 2], torch.tensor([0.0, 0.0, -1.0]), atol=1e-6)
    tl = torch.tensor([-0.4, -0.4, -1.0])
    assert torch.allclose(rd[0, 0], tl / tl.norm(), atol=1e-6)
    assert torch.allclose(rd.norm(dim=-1), torch.ones(4, 4), atol=1e-6)
    with pytest.raises(ValueError):
        get_rays(4, 4, 5.0, torch.eye(3))
