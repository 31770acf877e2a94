"""CPU tests of the file formats either side of the hot path: PNG IO, the checkpoint wire format of
scripts/train_nerf.py:118-131, and the loader oracle's blend arithmetic."""
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
    assert m2 is None and pv is None and not m1.use_viewdirs and cfg["precision"] == "fp32"
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
