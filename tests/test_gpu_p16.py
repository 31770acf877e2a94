"""GPU parity of the 16-bit layer-program kernels (nerf_mlp_p16_*): NeRF.forward and its parameter gradients
(nerf.py:173-209, train_nerf.py:334-335) for architectures outside the fused kernels, e.g. the reference's own test
network D=4, W=64, skips=[2] (tests/test_nerf_model.py:105-109).

References: the layer-by-layer torch evaluation with the kernels' operand rounding (weights and every layer input in
the operand type, fp32 accumulation and bias; straight-through gradient), which has the same ReLU masks as the kernel
-- tight tolerances -- and the plain fp32 evaluation / golden vectors -- loose ones (a unit whose fp32 pre-activation
is below the 16-bit rounding error may flip, see test_gpu_train.py).  For the scripts' 8 x 256 network the program
kernels are also checked against the fused chain kernels, which round the same operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import O, T, amd_model, maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"

ARCHS = [
    dict(D=4, W=64, skips=[2]),                                    # reference tests/test_nerf_model.py:105-109
    dict(use_viewdirs=False, D=6, W=128, skips=[3]),               # four row tiles: two column tiles per wave
    dict(D=3, W=100, skips=[0], input_ch_pts=27, input_ch_views=15),   # widths that need padding
    dict(D=12, W=32, skips=[1, 5, 9]),                             # several skips, the deepest program (16 layers)
    dict(D=5, W=256, skips=[1]),                                   # full-width layers, not the scripts' depth
]


@pytest.fixture(scope="module")
def amd():
    import nerf_amd
    return nerf_amd


def _dtype(precision):
    return torch.bfloat16 if precision == "bf16" else torch.float16


def _round_ste(x, dtype):
    return x + (x.to(dtype).float() - x).detach()


def forward_emulated(m, params, x_pts, x_views, dtype):
    """nerf.py:173-209 for the architecture of `m` on the parameter list `params` (state_dict order) with the
    kernels' operand rounding."""
    r = lambda t: _round_ste(t, dtype)   # noqa: E731
    it = iter(params)
    lin = lambda x: F.linear(x, r(next(it)), next(it))   # noqa: E731
    e = r(x_pts)
    h = e
    for i in range(m.D):
        h = r(torch.relu(lin(h)))
        if i in m.skips:
            h = torch.cat((e, h), dim=-1)
    if not m.use_viewdirs:
        return lin(h)
    sigma = lin(h)
    feat = r(lin(h))
    g = r(torch.relu(lin(torch.cat((feat, r(x_views)), dim=-1))))
    rgb = lin(g)
    return torch.cat((rgb, sigma), dim=-1)


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-300)).item()


def _inputs(m, M, seed):
    gen = torch.Generator(device=DEV).manual_seed(seed)
    xp = torch.randn(M, m.input_ch_pts, device=DEV, generator=gen)
    xv = torch.randn(M, m.input_ch_views, device=DEV, generator=gen) if m.use_viewdirs else None
    cot = torch.randn(M, 4, device=DEV, generator=gen)
    return xp, xv, cot


def _call(m, xp, xv):
    return m(xp, xv) if xv is not None else m(xp)


def test_p16_reference_test_network_golden(amd, golden):
    """The reference's test shape in bf16 / fp16 against the reference's own fp32 output (golden f3, raw_mini): the
    requested operand type is what runs (no silent switch to fp32), on this library's kernel."""
    from nerf_amd import ops
    g = golden("f3_mlp")
    ep = O.positional_encode(T(g["pts"])[:512], 10).to(DEV)
    ev = O.positional_encode(T(g["viewdirs"])[:512], 4).to(DEV)
    for precision, tol in (("bf16", 3e-3), ("fp16", 4e-4)):
        m = amd_model(O.init_params(2, D=4, W=64, skips=(2,)), precision, D=4, W=64, skips=[2])
        assert not m.fused() and m.program_supported() and m.effective_precision() == precision
        ops.KERNEL_TIMES = {}
        with torch.no_grad():
            raw = m(ep, ev)
        times, ops.KERNEL_TIMES = ops.KERNEL_TIMES, None
        assert "mlp_fwd_p16" in times and "mlp_fwd_fp32" not in times
        assert maxabs(raw, T(g["raw_mini"])) < tol
        with pytest.raises(NotImplementedError):
            m.forward_rays(torch.zeros(8, 3, device=DEV), torch.zeros(8, 3, device=DEV), torch.zeros(8, 4, device=DEV))


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("arch", ARCHS)
def test_p16_forward_and_gradients_every_architecture(amd, arch, precision):
    from nerf_amd import ops
    torch.manual_seed(11)
    dtype = _dtype(precision)
    m = amd.NeRF(precision=precision, **arch).to(DEV)
    assert m.program_supported() and not m.fused()
    params = list(m.parameters())
    # forward, ragged sizes (tiles of 128 samples): summation order only against the emulated evaluation
    fwd_tol = 2e-3 if precision == "bf16" else 3e-4
    for M in (1, 127, 1000):
        xp, xv, _ = _inputs(m, M, M)
        with torch.no_grad():
            raw = _call(m, xp, xv)
            ref = forward_emulated(m, params, xp, xv, dtype)
            ref32 = m._forward_library(xp, xv, None)
        assert raw.shape == (M, 4)
        assert maxabs(raw, ref) < fwd_tol * (1.0 + ref.abs().max().item()), (arch, M)
        assert maxabs(raw, ref32) < 25 * fwd_tol * (1.0 + ref32.abs().max().item())
    # leading dimensions are kept
    xp = torch.randn(5, 7, m.input_ch_pts, device=DEV)
    xv = torch.randn(5, 7, m.input_ch_views, device=DEV) if m.use_viewdirs else None
    assert _call(m, xp, xv).shape == (5, 7, 4)
    # gradients of every parameter tensor: tight against the emulated evaluation (same masks; dZ is rounded to the
    # operand type in the kernel and not in the reference), loose against fp32
    # (a bias gradient is a plain sum over samples of 16-bit-rounded values: where it cancels, as the sum of a random
    # cotangent does, the rounding of the terms -- 2^-9 / 2^-12 relative -- is what is left, so it gets that allowance)
    tight = 2e-2 if precision == "bf16" else 1.5e-2
    eps = 2.0 ** -9 if precision == "bf16" else 2.0 ** -12
    for M, scale in ((1, 1.0), (200, 1.0), (4097, 1e-4)):          # 1e-4: the loss-gradient scale of a 1024-ray batch
        xp, xv, cot = _inputs(m, M, 100 + M)
        cot = cot * scale
        ops.KERNEL_TIMES = {}
        g_kernel = torch.autograd.grad(_call(m, xp, xv), params, cot)
        times, ops.KERNEL_TIMES = ops.KERNEL_TIMES, None
        assert "mlp_fwd_p16_train" in times and "mlp_bwd_p16" in times and "mlp_bwd_fp32" not in times
        g_emul = torch.autograd.grad(forward_emulated(m, params, xp, xv, dtype), params, cot)
        g_fp32 = torch.autograd.grad(m._forward_library(xp, xv, None), params, cot)
        for (k, _), a, b, c in zip(m.named_parameters(), g_kernel, g_emul, g_fp32):
            assert a.shape == b.shape and torch.isfinite(a).all()
            if b.norm() == 0:
                assert a.norm() == 0, (arch, M, k)
                continue
            allow = eps * cot.norm().item() if k.endswith("bias") else 0.0
            err = (a.double() - b.double()).norm().item()
            assert err < tight * b.double().norm().item() + allow, (arch, precision, M, k, _rel(a, b))
            if M >= 200 and not k.endswith("bias"):
                assert _rel(a, c) < 0.35, (arch, precision, M, k, _rel(a, c))
    # size-independent: the gradient over 4097 samples is the sum of the gradients over two parts of them (a sample's
    # masks do not depend on its batch; in fp16 the parts may get another gradient scale, a power of two, so the
    # 16-bit rounding of dZ is the same)
    parts = [torch.autograd.grad(_call(m, xp[a:b], None if xv is None else xv[a:b]), params, cot[a:b])
             for a, b in ((0, 2048), (2048, 4097))]
    for (k, _), whole, p0, p1 in zip(m.named_parameters(), g_kernel, *parts):
        assert _rel(whole, p0 + p1) < 1e-5, (arch, k)
    # empty batch: zero gradients of the right shapes
    xp, xv, cot = _inputs(m, 0, 5)
    g0 = torch.autograd.grad(_call(m, xp, xv), params, cot)
    assert all(a.shape == p.shape and a.abs().max() == 0 for a, p in zip(g0, params))
    # a gradient with respect to the encoded inputs still comes from the library path
    xp = torch.randn(9, m.input_ch_pts, device=DEV, requires_grad=True)
    xv = torch.randn(9, m.input_ch_views, device=DEV) if m.use_viewdirs else None
    _call(m, xp, xv).sum().backward()
    assert xp.grad is not None and torch.isfinite(xp.grad).all() and xp.grad.abs().max() > 0
    m.zero_grad()
    # weights changed in place under no_grad: the cached program follows
    with torch.no_grad():
        xp = xp.detach()
        before = _call(m, xp, xv)
        for p in m.parameters():
            p.mul_(0.5)
        after = _call(m, xp, xv)
        assert maxabs(after, forward_emulated(m, list(m.parameters()), xp, xv, dtype)) < fwd_tol and maxabs(after, before) > 1e-4


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_p16_program_agrees_with_the_fused_kernels(amd, golden, precision):
    """The scripts' 8 x 256 network run through the layer-program kernels: raw and gradients agree with the fused
    chain kernels (same operand rounding; summation order, and units whose pre-activation rounds across a 16-bit
    boundary, differ)."""
    from nerf_amd.models.nerf import _ProgramForward
    g = golden("f3_mlp")
    P = 2000
    ep = O.positional_encode(T(g["pts"])[:P], 10).to(DEV)
    ev = O.positional_encode(T(g["viewdirs"])[:P], 4).to(DEV)
    cot = torch.randn(P, 4, generator=torch.Generator().manual_seed(3)).to(DEV) * 1e-4
    for views in (True, False):
        params = O.init_params(0) if views else O.init_params(3, use_viewdirs=False)
        kw = {} if views else dict(use_viewdirs=False)
        mf, mp = amd_model(params, precision, **kw), amd_model(params, precision, **kw)
        assert mf.fused()
        raw_f = mf(ep, ev) if views else mf(ep)
        raw_p = _ProgramForward.apply(mp, ep, ev if views else None, True, precision, *mp.ordered_params())
        assert maxabs(raw_p, raw_f) < (1.5e-3 if precision == "bf16" else 2e-4)
        (raw_f * cot).sum().backward()
        (raw_p * cot).sum().backward()
        for (k, a), b in zip(mp.named_parameters(), mf.parameters()):
            assert _rel(a.grad, b.grad) < (3e-2 if precision == "bf16" else 2e-2), (views, k, _rel(a.grad, b.grad))


def test_p16_nan_weight_gives_nan_output(amd):
    """A non-finite parameter reaches raw (and so the loss) as NaN, as in the reference (torch.relu keeps a NaN)."""
    m = amd.NeRF(D=4, W=64, skips=[2], precision="bf16").to(DEV)
    xp, xv, _ = _inputs(m, 300, 1)
    with torch.no_grad():
        m.pts_linears[1].weight[5, 7] = float("nan")
        raw = m(xp, xv)
    assert torch.isnan(raw).all()


def test_p16_training_step_small_network(amd, golden):
    """One hierarchical training step (train_nerf.py:293-336) on the reference's test-sized networks in bf16 against
    the same step in the fp32 validation mode: losses to 1 %, updated weights within Adam's step size."""
    from nerf_amd.optim import FlatAdam
    from nerf_amd.train import TrainStep
    e, g = golden("f8_e2e_render"), golden("f9_train_step")
    NT = g["target"].shape[0]
    ro, rd, tgt = T(e["rays_o"], DEV)[:NT], T(e["rays_d"], DEV)[:NT], T(g["target"], DEV)
    pe, pv = amd.PositionalEncoder(3, 10).to(DEV), amd.PositionalEncoder(3, 4).to(DEV)
    arch = dict(D=4, W=64, skips=[2])
    out = {}
    for precision, use_graph in (("fp32", False), ("bf16", False), ("bf16", True)):
        mc = amd_model(O.init_params(5, D=4, W=64, skips=(2,)), precision, **arch)
        mf = amd_model(O.init_params(6, D=4, W=64, skips=(2,)), precision, **arch)
        opt = FlatAdam(list(mc.parameters()) + list(mf.parameters()), lr=5e-4)
        step = TrainStep(mc, mf, pe, pv, opt, NT, N_samples_coarse=64, N_samples_fine=128, use_graph=use_graph,
                         external_uniforms=True)
        step.uniforms.copy_(torch.cat([T(g[k], DEV).reshape(-1) for k in ("t_rand_coarse", "u_fine", "t_rand_fine")]))
        losses = [step(ro, rd, tgt, lr=5e-4).item() for _ in range(3)]
        out[(precision, use_graph)] = (losses, torch.cat([p.detach().reshape(-1) for p in mc.parameters()]).clone())
    l32, w32 = out[("fp32", False)]
    l16, w16 = out[("bf16", False)]
    assert all(np.isfinite(l16))
    assert all(abs(a - b) < 1e-2 * b for a, b in zip(l16, l32)), (l16, l32)
    # Adam's first steps move every weight by about +-lr: a weight differs by 2 lr per step where the two gradients
    # disagree in sign, which only tiny gradients do
    assert (w16 - w32).abs().max() < 3 * 2.5 * 5e-4 and (w16 - w32).abs().mean() < 0.5 * 5e-4
    # the same step captured into a HIP graph (weight packing, both program kernels and their torch glue replayed)
    lg, wg = out[("bf16", True)]
    assert lg == l16 and torch.equal(wg, w16)


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_program_kernels_under_graph_replay_follow_the_weights(amd, precision):
    """A forward + backward of a layer-program network captured into a HIP graph packs its weights inside the graph:
    a replay after the weights changed in place (as an optimizer step inside the graph changes them) equals the eager
    evaluation of the new weights, bit for bit."""
    torch.manual_seed(0)
    m = amd.NeRF(D=4, W=64, skips=[2], precision=precision).to(DEV)
    xp, xv, cot = _inputs(m, 500, 9)
    params = list(m.parameters())

    def fwd_bwd():
        for p in params:
            p.grad = None
        out = m(xp, xv)
        torch.autograd.backward([out], [cot])
        return out, [p.grad for p in params]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd_bwd()
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        out_g, grads_g = fwd_bwd()
    for it in range(3):
        graph.replay()
        torch.cuda.synchronize()
        og, gg = out_g.clone(), [x.clone() for x in grads_g]
        oe, ge = fwd_bwd()
        assert torch.equal(og, oe) and all(torch.equal(a, b) for a, b in zip(gg, ge)), it
        with torch.no_grad():
            for p in params:
                p.mul_(0.7)
