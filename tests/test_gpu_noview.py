"""GPU parity of the fused 16-bit kernels for the network without view directions
(NeRF(use_viewdirs=False), src/nerf_project/models/nerf.py:151-153 and :202-206 -- the default of
scripts/train_nerf.py:79): forward against the reference's golden raw values, both forward kernels against
each other, gradients against the oracle with the kernels' operand rounding (tight) and in fp32 (loose, for
the reason test_gpu_train.py gives), and the coarse-only training step of train_nerf.py:293-303."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import O, T, amd_model, maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"
NV = dict(use_viewdirs=False)


@pytest.fixture(scope="module")
def amd():
    import nerf_amd
    return nerf_amd


def _round_ste(x, dtype):
    return x + (x.to(dtype).float() - x).detach()


def mlp_noview_emulated(params, x_pts, dtype):
    """oracle MLP without view directions with the kernels' operand rounding (weights and layer inputs in
    `dtype`, fp32 accumulation and bias)."""
    r = lambda t: _round_ste(t, dtype)   # noqa: E731
    e = r(x_pts)
    h = e
    for i in range(8):
        h = r(F.relu(F.linear(h, r(params[f"pts_linears.{i}.weight"]), params[f"pts_linears.{i}.bias"])))
        if i == 4:
            h = torch.cat([e, h], -1)
    return F.linear(h, r(params["output_linear.weight"]), params["output_linear.bias"])


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-300)).item()


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_noview_forward_golden(amd, golden, precision):
    g = golden("f3_mlp")
    m = amd_model(O.init_params(3, **NV), precision, **NV)
    assert m.fused() and m.effective_precision() == precision
    pts = T(g["pts"], DEV)
    tol = 3e-3 if precision == "bf16" else 4e-4          # operand rounding through nine layers
    with torch.no_grad():
        z = torch.zeros(pts.shape[0], 1, device=DEV)
        raw_rays = m.forward_rays(pts, torch.ones_like(pts), z).view(-1, 4)       # inference kernel (16x16x32)
        ep = amd.PositionalEncoder(3, 10).to(DEV)(pts)
        raw_enc = m(ep)                                                          # encoded-input forward (32x32x16)
        raw_enc_v = m(ep, torch.zeros(pts.shape[0], 27, device=DEV))             # view encodings are ignored
    assert maxabs(raw_rays[:512], T(g["raw_noview"])) < tol
    assert maxabs(raw_enc[:512], T(g["raw_noview"])) < tol
    assert torch.equal(raw_enc, raw_enc_v)
    # the two kernels round the same operands and differ in summation order and in-kernel sin/cos only
    assert maxabs(raw_rays, raw_enc) < (1e-3 if precision == "bf16" else 1.5e-4)
    # the training forward (activations kept) is the same arithmetic as the encoded forward's chain
    raw_train = m.forward_rays(pts, torch.ones_like(pts), z).view(-1, 4)
    assert raw_train.requires_grad and maxabs(raw_train, raw_rays) < (1e-3 if precision == "bf16" else 1.5e-4)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("P", [1, 300, 4173])
def test_noview_gradients(amd, golden, precision, P):
    g = golden("f3_mlp")
    dtype = torch.bfloat16 if precision == "bf16" else torch.float16
    pts = T(g["pts"])
    pts = pts.repeat((P + pts.shape[0] - 1) // pts.shape[0], 1)[:P] * torch.linspace(0.5, 1.0, P)[:, None]
    cot = torch.randn(P, 4, generator=torch.Generator().manual_seed(P)) * 1e-4
    ep = O.positional_encode(pts, 10)
    pe = {k: v.clone().requires_grad_(True) for k, v in O.init_params(3, **NV).items()}
    raw_e = mlp_noview_emulated(pe, ep, dtype)
    (raw_e * cot).sum().backward()
    pf = {k: v.clone().requires_grad_(True) for k, v in O.init_params(3, **NV).items()}
    (O.mlp_forward(pf, ep, None, use_viewdirs=False) * cot).sum().backward()

    m = amd_model(O.init_params(3, **NV), precision, **NV)
    raw = m.forward_rays(pts.to(DEV), torch.ones(P, 3, device=DEV), torch.zeros(P, 1, device=DEV))
    (raw.view(-1, 4) * cot.to(DEV)).sum().backward()
    assert maxabs(raw.view(-1, 4), raw_e) < (1e-3 if precision == "bf16" else 1.5e-4)
    tight = 2e-2 if precision == "bf16" else 1.5e-2
    loose = 0.25 if precision == "bf16" else 0.1
    names = [k for k, _ in m.named_parameters()]
    assert names == list(pe.keys()) and len(names) == 18
    for k, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape
        got = p.grad.cpu()
        if P >= 300:            # with one sample most units are off and single flips dominate the norm
            assert _rel(got, pe[k].grad) < tight, (k, "vs emulated-precision oracle", _rel(got, pe[k].grad))
            assert _rel(got, pf[k].grad) < loose, (k, "vs fp32 oracle", _rel(got, pf[k].grad))
        else:
            assert maxabs(got, pe[k].grad) < 5e-2 * pe[k].grad.abs().max().item() + 1e-12, k
    # the combined C entry point gives the split calls' gradients bit for bit
    from nerf_amd import ops, _C
    lib, _p = ops.lib, ops._p
    M = P
    prec = ops._prec(precision, False)
    packed = m.packed_weights()
    st = torch.cuda.current_stream().cuda_stream
    saved = torch.empty(int(lib.nerf_mlp_saved_bytes(M)), dtype=torch.uint8, device=DEV)
    raw2 = torch.empty(M, 4, device=DEV)
    ro, rd, z = pts.to(DEV).contiguous(), torch.ones(P, 3, device=DEV), torch.zeros(P, 1, device=DEV)
    ops.check(lib.nerf_mlp_fwd(M, 1, _p(ro), _p(rd), _p(z), _p(packed), prec, _p(raw2), _p(saved), st), "fwd")
    ws = torch.empty(int(lib.nerf_mlp_bwd_workspace_bytes(M)), dtype=torch.uint8, device=DEV)
    d_flat = torch.empty(_C.NERF_MLP_NPARAMS_NOVIEW, device=DEV)
    d_raw = cot.to(DEV).contiguous()
    scale = None
    if precision == "fp16":
        scale = torch.exp2(torch.floor(torch.log2(1024.0 / d_raw.abs().amax()))).reshape(1)
    ops.check(lib.nerf_mlp_bwd(M, _p(d_raw), _p(packed), prec, _p(saved), _p(scale), _p(ws), ws.numel(), _p(d_flat), st), "bwd")
    flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.equal(flat, d_flat)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_noview_render_and_coarse_only_step(amd, golden, precision):
    """render_rays with the fused network (view directions ignored whatever use_viewdirs says) against the
    oracle's render, and one coarse-only training step (train_nerf.py:293-303, the script's default flags)
    through TrainStep against the oracle's loss and gradient norms."""
    from nerf_amd.optim import FlatAdam
    from nerf_amd.train import TrainStep
    e = golden("f8_e2e_render")
    n = 256
    ro, rd = T(e["rays_o"])[:n], T(e["rays_d"])[:n]
    params = O.init_params(3, **NV)
    with torch.no_grad():                        # sigma bias + 1: every ray well conditioned (no mask needed)
        params["output_linear.bias"][3] += 1.0
    m = amd_model(params, precision, **NV)
    pe = amd.PositionalEncoder(3, 10).to(DEV)
    with torch.no_grad():
        out = amd.render_rays(m, pe, None, ro.to(DEV), rd.to(DEV), 2.0, 6.0, 64, use_viewdirs=False)
        out_v = amd.render_rays(m, pe, amd.PositionalEncoder(3, 4).to(DEV), ro.to(DEV), rd.to(DEV), 2.0, 6.0, 64)
        ref = O.render_rays(params, ro, rd, 2.0, 6.0, 64, use_viewdirs=False)
    tol = 1e-3
    for k in ("rgb_map", "acc_map", "depth_map"):
        assert maxabs(out[k], ref[k]) < tol, (k, maxabs(out[k], ref[k]))
        assert torch.equal(out[k], out_v[k])
    # one coarse-only step with injected jitter
    gen = torch.Generator().manual_seed(5)
    t_rand, tgt = torch.rand(n, 64, generator=gen), torch.rand(n, 3, generator=gen)
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.render_rays(po, ro, rd, 2.0, 6.0, 64, t_rand=t_rand, use_viewdirs=False)
    loss_ref = torch.mean((ref["rgb_map"] - tgt) ** 2)
    loss_ref.backward()
    opt = FlatAdam(list(m.parameters()), lr=5e-4)
    step = TrainStep(m, None, pe, None, opt, n, N_samples_coarse=64, use_graph=False, external_uniforms=True)
    step.uniforms.copy_(t_rand.reshape(-1).to(DEV))
    w_before = opt.flat.clone()
    loss = step(ro.to(DEV), rd.to(DEV), tgt.to(DEV), lr=5e-4)
    assert abs(loss.item() - loss_ref.item()) < 2e-3 * loss_ref.item()
    off = 0
    for (k, p), n_el in zip(m.named_parameters(), opt.sizes):
        got = opt.flat_grad[off:off + n_el].cpu().view(p.shape)
        off += n_el
        assert _rel(got, po[k].grad) < (0.25 if precision == "bf16" else 0.1), (k, _rel(got, po[k].grad))
    assert not torch.equal(opt.flat, w_before)            # Adam stepped
