#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (development container only).

Usage (from the repo root; nothing is written into the reference tree):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/src python3 oracle/gen_golden.py

The script (1) imports ``nerf_project`` from the reference checkout, (2) checks
every function of ``oracle/nerf_oracle.py`` against it on the same inputs
(bit-exact on this machine, since the restatement issues the same ATen ops),
and (3) stores inputs + reference outputs as small fixtures.  Weights are not
stored: they are regenerated from ``nerf_oracle.init_params(seed)`` and pinned
by a SHA-256 in ``meta.json``.  Random draws inside the reference
(``torch.rand`` at nerf.py:255 and nerf.py:400) are replaced by a patched
``torch.rand`` that hands out pre-drawn tensors, so the reference code itself
runs unmodified.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import nerf_oracle as O  # noqa: E402

from nerf_project.models.nerf import NeRF, PositionalEncoder, render_rays, sample_pdf  # noqa: E402
from nerf_project.utils.ray_utils import get_rays  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


class PatchedRand:
    """Context manager: torch.rand returns the queued tensors in order."""

    def __init__(self, queue):
        self.queue = list(queue)

    def __enter__(self):
        self._orig = torch.rand

        def fake(*size, **kw):
            t = self.queue.pop(0)
            shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
            assert tuple(t.shape) == shape, (t.shape, shape)
            return t.clone()

        torch.rand = fake
        return self

    def __exit__(self, *a):
        torch.rand = self._orig
        assert not self.queue, "unused uniforms"


def ref_model(params, **kw):
    m = NeRF(**kw)
    m.load_state_dict(params, strict=True)
    return m.eval()


def eq(a, b, what):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if not torch.equal(a, b):
        d = (a.double() - b.double()).abs().max().item()
        raise SystemExit(f"oracle != reference for {what}: max abs diff {d}")


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        conv[k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **conv)


meta = {"torch": torch.__version__}
rng = np.random.default_rng(1234)


def U(*shape):
    return torch.from_numpy(rng.random(shape, dtype=np.float32))


# ---- weights ---------------------------------------------------------------
pc = O.init_params(0)
pf = O.init_params(1)
pmini = O.init_params(2, D=4, W=64, skips=(2,))
meta["sha_coarse_seed0"] = O.params_sha256(pc)
meta["sha_fine_seed1"] = O.params_sha256(pf)
meta["sha_mini_seed2"] = O.params_sha256(pmini)
assert [k for k, _ in O.param_shapes()] == list(NeRF().state_dict().keys())
assert [tuple(v.shape) for v in NeRF().state_dict().values()] == [s for _, s in O.param_shapes()]
assert [k for k, _ in O.param_shapes(use_viewdirs=False)] == list(NeRF(use_viewdirs=False).state_dict().keys())
meta["n_params"] = int(sum(v.numel() for v in pc.values()))

# ---- F1 get_rays -------------------------------------------------------------
focal800 = float(O.focal_from_fov(800, O.CAMERA_ANGLE_X))
poses = {"identity": torch.eye(4), "sph40": O.pose_spherical(40.0, -30.0, 4.0)}
f1 = {"focal800": np.float64(focal800)}
for nm, c2w in poses.items():
    ro, rd = get_rays(8, 8, 5.0, c2w)
    oro, ord_ = O.get_rays(8, 8, 5.0, c2w)
    eq(oro, ro, "get_rays o"); eq(ord_, rd, "get_rays d")
    f1[f"{nm}_c2w"] = c2w; f1[f"{nm}_o8"] = ro; f1[f"{nm}_d8"] = rd
    ro, rd = get_rays(800, 800, focal800, c2w)
    oro, ord_ = O.get_rays(800, 800, focal800, c2w)
    eq(ord_, rd, "get_rays d 800")
    f1[f"{nm}_d800_crop"] = rd[392:408, 792:800].clone()   # rows 392..407, last 8 columns
    f1[f"{nm}_d800_corner"] = rd[:4, :4].clone()
    f1[f"{nm}_d800_sum"] = rd.double().sum(dim=(0, 1))
# 3x4 pose accepted (ray_utils.py:26)
ro, rd = get_rays(8, 8, 5.0, poses["sph40"][:3])
eq(rd, f1["sph40_d8"], "3x4 pose")
save("f1_get_rays", **f1)

# ---- F2 positional encoding ---------------------------------------------------
x = (U(256, 3) * 12.0 - 6.0)
pe10, pe4 = PositionalEncoder(3, 10), PositionalEncoder(3, 4)
pelin = PositionalEncoder(3, 5, log_sampling=False)
e10, e4, el = pe10(x), pe4(x), pelin(x)
eq(O.positional_encode(x, 10), e10, "pe10"); eq(O.positional_encode(x, 4), e4, "pe4")
eq(O.positional_encode(x, 5, False), el, "pe lin")
eq(O.freq_bands(10), pe10.freq_bands, "freq")
save("f2_posenc", x=x, e10=e10, e4=e4, elin5=el)

# ---- F3 MLP --------------------------------------------------------------------
P = 4096
pts = U(P, 3) * 8.0 - 4.0
vd = U(P, 3) - 0.5
vd = vd / vd.norm(dim=-1, keepdim=True)
ep, ev = pe10(pts), pe4(vd)
with torch.no_grad():
    raw = ref_model(pc)(ep, ev)
    oraw = O.mlp_forward(pc, ep, ev)
    eq(oraw, raw, "mlp")
    raw_mini = ref_model(pmini, D=4, W=64, skips=[2])(ep[:512], ev[:512])
    eq(O.mlp_forward(pmini, ep[:512], ev[:512], D=4, skips=(2,)), raw_mini, "mlp mini")
    pnv = O.init_params(3, use_viewdirs=False)
    raw_nv = ref_model(pnv, use_viewdirs=False)(ep[:512])
    eq(O.mlp_forward(pnv, ep[:512], None, use_viewdirs=False), raw_nv, "mlp noview")
save("f3_mlp", pts=pts, viewdirs=vd, raw=raw, raw_mini=raw_mini, raw_noview=raw_nv)

# ---- F4/F5 compositing ----------------------------------------------------------
N, S = 96, 192
raw_c = torch.from_numpy(rng.standard_normal((N, S, 4), dtype=np.float32))
raw_c[..., 3] = raw_c[..., 3] * 2.0 + 0.3
z_c, _ = torch.sort(U(N, S) * 4.0 + 2.0, dim=-1)
# edge rows: all sigma<=0, huge sigma, duplicate z, alpha==1 in the middle, zero sigma last
raw_c[0, :, 3] = -1.0
raw_c[1, :, 3] = 1e4
z_c[2, 50:60] = z_c[2, 50]
raw_c[3, 10, 3] = 1e9
raw_c[4, -1, 3] = 0.0
raw_c[5, :, 3] = 0.0
raw_c[6, :, 3] = 1e-9


class _Const(torch.nn.Module):
    def __init__(self, r):
        super().__init__(); self.r = r

    def forward(self, a, b=None):
        return self.r.reshape(-1, 4)


ident = lambda t: t  # noqa: E731
rays_o0 = torch.zeros(N, 3); rays_d0 = torch.zeros(N, 3); rays_d0[:, 2] = -1.0
f4 = {"raw": raw_c, "z": z_c}
for wb in (False, True):
    rr = raw_c.clone().requires_grad_(True)
    res = render_rays(_Const(rr), ident, ident, rays_o0, rays_d0, 2.0, 6.0, S, white_bkgd=wb,
                      z_vals_override=z_c)
    oo = O.composite(raw_c, z_c, wb)
    for k in ("rgb_map", "depth_map", "disp_map", "acc_map", "weights", "alpha", "raw_sigma"):
        eq(oo[k], res[k], f"composite {k} wb={wb}")
        f4[f"{k}_wb{int(wb)}"] = res[k]
    # F5 gradients for a fixed cotangent on rgb/depth/acc (+weights)
    g_rgb, g_d, g_a, g_w = U(N, 3) - 0.5, U(N) - 0.5, U(N) - 0.5, U(N, S) - 0.5
    loss = (res["rgb_map"] * g_rgb).sum() + (res["depth_map"] * g_d).sum() + (res["acc_map"] * g_a).sum()
    (graw,) = torch.autograd.grad(loss, rr, retain_graph=True)
    loss2 = loss + (res["weights"] * g_w).sum()
    (graw2,) = torch.autograd.grad(loss2, rr)
    f4[f"g_rgb_wb{int(wb)}"] = g_rgb; f4[f"g_depth_wb{int(wb)}"] = g_d; f4[f"g_acc_wb{int(wb)}"] = g_a
    f4[f"g_w_wb{int(wb)}"] = g_w
    f4[f"draw_wb{int(wb)}"] = graw; f4[f"draw_withw_wb{int(wb)}"] = graw2
save("f4_composite", **f4)

# ---- F6 sample_pdf ----------------------------------------------------------------
Np = 256
bins, _ = torch.sort(U(Np, 63) * 4.0 + 2.0, dim=-1)
w = U(Np, 62) ** 4
w[0] = 0.0                     # all-zero weights
w[1] = 0.0; w[1, 30] = 1.0     # single spike
w[2] = 1.0                     # uniform
u = U(Np, 128)
u[3, :4] = torch.tensor([0.0, 1.0 - 2.0 ** -24, 0.5, 1e-8])
with PatchedRand([u]):
    s_rand = sample_pdf(bins, w, 128, det=False)
s_det = sample_pdf(bins, w, 128, det=True)
o_rand, inds_rand, cdf = O.sample_pdf(bins, w, 128, det=False, u=u, return_inds=True)
o_det, inds_det, _ = O.sample_pdf(bins, w, 128, det=True, return_inds=True)
eq(o_rand, s_rand, "sample_pdf rand"); eq(o_det, s_det, "sample_pdf det")
save("f6_sample_pdf", bins=bins, weights=w, u=u, samples_rand=s_rand, samples_det=s_det,
     inds_rand=inds_rand.to(torch.int32), inds_det=inds_det.to(torch.int32), cdf=cdf)

# ---- F7 merge sort --------------------------------------------------------------------
zc, _ = torch.sort(U(Np, 64) * 4.0 + 2.0, dim=-1)
zf = U(Np, 128) * 4.0 + 2.0
zf[0, :10] = zc[0, :10]   # ties between the lists
zs, _ = torch.sort(torch.cat([zc, zf], -1), -1)
save("f7_merge", z_coarse=zc, z_fine=zf, z_sorted=zs)

# ---- F8 end-to-end config 1 (render 1024 rays, det) --------------------------------------
NR = 1024
ro, rd = O.synthetic_rays(NR, seed=0)
mc, mf = ref_model(pc), ref_model(pf)
f8 = {"rays_o": ro, "rays_d": rd}
with torch.no_grad():
    for lind in (False, True):
        for wb in (False, True):
            c = render_rays(mc, pe10, pe4, ro, rd, 2.0, 6.0, 64, rand=False, lindisp=lind, white_bkgd=wb)
            zm = 0.5 * (c["z_vals"][..., 1:] + c["z_vals"][..., :-1])
            zs_f = sample_pdf(zm, c["weights"][..., 1:-1] + 1e-5, 128, det=True)
            zc2, _ = torch.sort(torch.cat([c["z_vals"], zs_f], -1), -1)
            f = render_rays(mf, pe10, pe4, ro, rd, 2.0, 6.0, 192, rand=False, lindisp=lind, white_bkgd=wb,
                            z_vals_override=zc2)
            oc, of = O.render_hierarchical(pc, pf, ro, rd, 2.0, 6.0, 64, 128, det=True, lindisp=lind,
                                           white_bkgd=wb)
            tag = f"l{int(lind)}w{int(wb)}"
            for k in ("rgb_map", "depth_map", "acc_map", "disp_map"):
                eq(oc[k], c[k], f"e2e coarse {k} {tag}"); eq(of[k], f[k], f"e2e fine {k} {tag}")
                f8[f"c_{k}_{tag}"] = c[k]; f8[f"f_{k}_{tag}"] = f[k]
            f8[f"c_sigma_last_{tag}"] = c["raw_sigma"][:, -1]
            f8[f"f_sigma_last_{tag}"] = f["raw_sigma"][:, -1]
            if not wb and not lind:
                f8[f"z_comb_{tag}"] = zc2
                f8[f"c_weights_{tag}"] = c["weights"]
                f8[f"c_raw_sigma_{tag}"] = c["raw_sigma"]
                f8[f"f_raw_sigma_{tag}"] = f["raw_sigma"]
save("f8_e2e_render", **f8)

# ---- F9 config 2 training step (injected uniforms) ------------------------------------------
NT = 256   # fixture-sized batch; the 1024-ray step is checked against the oracle at test time
ro_t, rd_t = ro[:NT], rd[:NT]
t1, uu, t2, tgt = U(NT, 64), U(NT, 128), U(NT, 192), U(NT, 3)
mc_t, mf_t = NeRF(), NeRF()
mc_t.load_state_dict(pc); mf_t.load_state_dict(pf)
with PatchedRand([t1, uu, t2]):
    c = render_rays(mc_t, pe10, pe4, ro_t, rd_t, 2.0, 6.0, 64, rand=True)
    loss_c = torch.mean((c["rgb_map"] - tgt) ** 2)
    with torch.no_grad():
        zm = 0.5 * (c["z_vals"][..., 1:] + c["z_vals"][..., :-1])
        zs_f = sample_pdf(zm, c["weights"].detach()[..., 1:-1] + 1e-5, 128, det=False)
    zc2, _ = torch.sort(torch.cat([c["z_vals"].detach(), zs_f], -1), -1)
    f = render_rays(mf_t, pe10, pe4, ro_t, rd_t, 2.0, 6.0, 192, rand=True, z_vals_override=zc2)
    loss_f = torch.mean((f["rgb_map"] - tgt) ** 2)
    total = loss_c + loss_f
total.backward()
pco = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
pfo = {k: v.clone().requires_grad_(True) for k, v in pf.items()}
ot, olc, olf, _, _ = O.train_step_loss(pco, pfo, ro_t, rd_t, tgt, 2.0, 6.0, 64, 128, t1, uu, t2)
ot.backward()
eq(olc, loss_c, "loss_c"); eq(olf, loss_f, "loss_f")
f9 = {"t_rand_coarse": t1, "u_fine": uu, "t_rand_fine": t2, "target": tgt,
      "loss_c": loss_c.detach(), "loss_f": loss_f.detach(), "z_comb": zc2}
for tag, m, po in (("c", mc_t, pco), ("f", mf_t, pfo)):
    norms = []
    for k, p in m.named_parameters():
        eq(po[k].grad, p.grad, f"grad {tag} {k}")
        norms.append(p.grad.double().norm().item())
    f9[f"gnorm_{tag}"] = np.array(norms)
    f9[f"g_{tag}_sigma_w"] = m.sigma_linear.weight.grad
    f9[f"g_{tag}_rgb2_w"] = m.rgb_layers[2].weight.grad
    f9[f"g_{tag}_rgb0_b"] = m.rgb_layers[0].bias.grad
    f9[f"g_{tag}_l0_b"] = m.pts_linears[0].bias.grad
    f9[f"g_{tag}_l5_w_head"] = m.pts_linears[5].weight.grad[:8, :80].clone()
save("f9_train_step", **f9)


# ---- F8b end-to-end config 1 with a well-conditioned weight set ---------------------------------------
# With random-init weights 1-2 % of rays have |raw_sigma_last| below the 16-bit operand-rounding error,
# and the 1e10-long last interval turns that sign into a 0/1 opacity.  sigma_linear.bias += 1 (both
# networks) moves every sigma_last away from zero, so that ALL rays can be compared through the 16-bit
# kernels without a mask.  Same rays as F8 (only the maps are stored).
pcb = {k: v.clone() for k, v in pc.items()}
pfb = {k: v.clone() for k, v in pf.items()}
pcb["sigma_linear.bias"] += 1.0
pfb["sigma_linear.bias"] += 1.0
meta["sha_coarse_seed0_sigma_bias1"] = O.params_sha256(pcb)
meta["sha_fine_seed1_sigma_bias1"] = O.params_sha256(pfb)
mcb, mfb = ref_model(pcb), ref_model(pfb)
f8b = {}
with torch.no_grad():
    for lind in (False, True):
        for wb in (False, True):
            c = render_rays(mcb, pe10, pe4, ro, rd, 2.0, 6.0, 64, rand=False, lindisp=lind, white_bkgd=wb)
            zm = 0.5 * (c["z_vals"][..., 1:] + c["z_vals"][..., :-1])
            zs_f = sample_pdf(zm, c["weights"][..., 1:-1] + 1e-5, 128, det=True)
            zc2, _ = torch.sort(torch.cat([c["z_vals"], zs_f], -1), -1)
            f = render_rays(mfb, pe10, pe4, ro, rd, 2.0, 6.0, 192, rand=False, lindisp=lind, white_bkgd=wb,
                            z_vals_override=zc2)
            oc, of = O.render_hierarchical(pcb, pfb, ro, rd, 2.0, 6.0, 64, 128, det=True, lindisp=lind,
                                           white_bkgd=wb)
            tag = f"l{int(lind)}w{int(wb)}"
            for k in ("rgb_map", "depth_map", "acc_map", "disp_map"):
                eq(oc[k], c[k], f"e2e-b coarse {k} {tag}"); eq(of[k], f[k], f"e2e-b fine {k} {tag}")
                f8b[f"c_{k}_{tag}"] = c[k]; f8b[f"f_{k}_{tag}"] = f[k]
            f8b[f"c_sigma_last_min_{tag}"] = c["raw_sigma"][:, -1].abs().min()
            f8b[f"f_sigma_last_min_{tag}"] = f["raw_sigma"][:, -1].abs().min()
            if not lind and not wb:
                f8b["z_comb_l0w0"] = zc2
save("f8b_e2e_wellcond", **f8b)

# ---- F10 checkpoints written by the reference's own classes -------------------------------------------
# scripts/train_nerf.py:118-131 stores {epoch, global_step, args_saved_at_checkpoint = vars(args), H, W,
# focal, model_coarse_state_dict, optimizer_state_dict[, model_fine_state_dict]} with torch.save.  The
# script itself cannot be imported here (configargparse is absent), so the dict is assembled exactly as
# those lines do, from the reference's NeRF class, torch.optim.Adam over coarse+fine parameters
# (train_nerf.py:203-206) after two real optimizer steps, and a plain dict with the keys of
# parse_args() (train_nerf.py:48-85).  Two small architectures: hierarchical with view directions, and
# the script's DEFAULT flags (no view directions, coarse only).
def ref_args(**over):
    a = {"config": None, "expname": "fixture", "basedir": "./logs/", "dataset_path": "data/none",
         "data_white_bkgd": False, "white_bkgd": False, "near": 2.0, "far": 6.0, "N_samples_coarse": 16,
         "N_samples_fine": 24, "use_hierarchical": False, "lindisp": False, "num_epochs": 200,
         "batch_size": 1024, "lr": 5e-4, "lr_decay_factor": 0.1, "lr_decay_steps": 250000, "seed": 42,
         "num_workers": 4, "net_depth": 8, "net_width": 256, "input_ch_pts_L": 10, "input_ch_views_L": 4,
         "use_viewdirs": False, "log_freq": 100, "save_freq": 10, "val_img_freq": 5000,
         "resume_from_checkpoint": None}
    a.update(over)
    return a


def make_ckpt(name, args, seed):
    torch.manual_seed(seed)
    use_v, hier = args["use_viewdirs"], args["use_hierarchical"]
    pe_p = PositionalEncoder(3, args["input_ch_pts_L"])
    pe_v = PositionalEncoder(3, args["input_ch_views_L"]) if use_v else None
    kw = dict(D=args["net_depth"], W=args["net_width"], input_ch_pts=pe_p.output_dims,
              input_ch_views=pe_v.output_dims if use_v else 0, skips=[args["net_depth"] // 2], use_viewdirs=use_v)
    m_c = NeRF(**kw)
    m_f = NeRF(**kw) if hier else None
    params = list(m_c.parameters()) + (list(m_f.parameters()) if hier else [])
    opt = torch.optim.Adam(params, lr=args["lr"])
    n = 64
    r_o, r_d = ro[:n], rd[:n]
    tgt_ = U(n, 3)
    nc, nf = args["N_samples_coarse"], args["N_samples_fine"]

    def two_pass(rand, queue):
        with PatchedRand(queue):
            c = render_rays(m_c, pe_p, pe_v, r_o, r_d, 2.0, 6.0, nc, rand=rand, use_viewdirs=use_v)
            f = None
            if hier:
                with torch.no_grad():
                    zm = 0.5 * (c["z_vals"][..., 1:] + c["z_vals"][..., :-1])
                    zs_ = sample_pdf(zm, c["weights"].detach()[..., 1:-1] + 1e-5, nf, det=not rand)
                zc_, _ = torch.sort(torch.cat([c["z_vals"].detach(), zs_], -1), -1)
                f = render_rays(m_f, pe_p, pe_v, r_o, r_d, 2.0, 6.0, nc + nf, rand=rand, use_viewdirs=use_v,
                                z_vals_override=zc_)
        return c, f

    for _ in range(2):                                    # train_nerf.py:293-336
        q = [U(n, nc)] + ([U(n, nf), U(n, nc + nf)] if hier else [])
        c, f = two_pass(True, q)
        loss = torch.mean((c["rgb_map"] - tgt_) ** 2)
        if hier:
            loss = loss + torch.mean((f["rgb_map"] - tgt_) ** 2)
        opt.zero_grad(); loss.backward(); opt.step()
    save_dict = {"epoch": 3, "global_step": 2, "args_saved_at_checkpoint": args, "H": 20, "W": 24, "focal": 30.0,
                 "model_coarse_state_dict": m_c.state_dict(), "optimizer_state_dict": opt.state_dict()}
    if hier:
        save_dict["model_fine_state_dict"] = m_f.state_dict()
    torch.save(save_dict, os.path.join(OUT, name + ".pth"))
    # what the reference renders from these weights (render_nerf.py:275-303: rand=False, det=True)
    with torch.no_grad():
        c, f = two_pass(False, [])
    out = {"rays_o": r_o, "rays_d": r_d, "c_rgb_map": c["rgb_map"], "c_depth_map": c["depth_map"],
           "c_acc_map": c["acc_map"]}
    if hier:
        out.update({"f_rgb_map": f["rgb_map"], "f_depth_map": f["depth_map"], "f_acc_map": f["acc_map"]})
    # one more reference optimizer step from the saved state: what a resumed run must reproduce
    q = [U(n, nc)] + ([U(n, nf), U(n, nc + nf)] if hier else [])
    out["t_rand_coarse"] = q[0]
    if hier:
        out["u_fine"], out["t_rand_fine"] = q[1], q[2]
    out["target"] = tgt_
    c, f = two_pass(True, q)
    loss = torch.mean((c["rgb_map"] - tgt_) ** 2)
    if hier:
        loss = loss + torch.mean((f["rgb_map"] - tgt_) ** 2)
    opt.zero_grad(); loss.backward(); opt.step()
    out["loss_step3"] = loss.detach()
    out["c_l0_bias_after_step3"] = m_c.pts_linears[0].bias.detach().clone()
    out["c_last_weight_after_step3"] = (m_c.rgb_layers[2].weight if use_v else m_c.output_linear.weight).detach().clone()
    save(name + "_render", **out)


make_ckpt("f10_ckpt_hier_views", ref_args(net_depth=4, net_width=32, input_ch_pts_L=6, input_ch_views_L=2,
                                          use_viewdirs=True, use_hierarchical=True), 7)
make_ckpt("f10_ckpt_default_flags", ref_args(net_depth=4, net_width=32, input_ch_pts_L=6), 8)

# ---- call signatures of the drop-in boundary ---------------------------------------------------------
import inspect  # noqa: E402
meta["signatures"] = {
    "PositionalEncoder.__init__": str(inspect.signature(PositionalEncoder.__init__)),
    "PositionalEncoder.forward": str(inspect.signature(PositionalEncoder.forward)),
    "NeRF.__init__": str(inspect.signature(NeRF.__init__)),
    "NeRF.forward": str(inspect.signature(NeRF.forward)),
    "sample_pdf": str(inspect.signature(sample_pdf)),
    "render_rays": str(inspect.signature(render_rays)),
    "get_rays": str(inspect.signature(get_rays)),
}

with open(os.path.join(OUT, "meta.json"), "w") as fh:
    json.dump(meta, fh, indent=1, sort_keys=True)
tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
print("golden fixtures written to", OUT, f"({tot / 1e6:.2f} MB); oracle == reference on every check")
