"""GPU: nerf_mlp_fwd_encoded_static -- the forward on encoded inputs whose staging buffer is a static device
array (an entry point that allocates nothing) -- against nerf_mlp_fwd_encoded with a caller-owned workspace,
through the C ABI: bit-identical outputs on two calls in a row, the second with a smaller input."""
import ctypes

import pytest
import torch

from util import O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_static_scratch_forward_equals_workspace_forward(precision):
    from nerf_amd import _C, ops
    lib = _C.lib
    prec = _C.PREC[precision]
    P = lambda t: ctypes.c_void_p(t.data_ptr())        # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    flat = torch.cat([v.reshape(-1) for v in O.init_params(0).values()]).to(DEV)
    packed = ops.mlp_pack(flat, precision)
    max_m = int(lib.nerf_mlp_fwd_encoded_static_max_samples())
    assert max_m == 65536
    gen = torch.Generator().manual_seed(17)
    for M in (5000 + 37, 300):                 # the second call is smaller: stale units of the first stay behind it
        pts = torch.rand(M, 3, generator=gen) * 6 - 3
        vd = torch.nn.functional.normalize(torch.randn(M, 3, generator=gen), dim=-1)
        ep, ev = O.positional_encode(pts, 10).to(DEV), O.positional_encode(vd, 4).to(DEV)
        ws = torch.empty(int(lib.nerf_mlp_encoded_workspace_bytes(M)), dtype=torch.uint8, device=DEV)
        want = torch.full((M, 4), float("nan"), device=DEV)
        got = torch.full((M, 4), float("nan"), device=DEV)
        _C.check(lib.nerf_mlp_fwd_encoded(M, P(ep), P(ev), P(packed), prec, P(want), None, P(ws), st), "fwd_encoded")
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        _C.check(lib.nerf_mlp_fwd_encoded_static(M, P(ep), P(ev), P(packed), prec, P(got), st), "fwd_encoded_static")
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        assert torch.isfinite(got).all() and torch.equal(got, want), (precision, M)
        # and against the oracle's fp32 network, as for every MLP kernel
        ref = O.mlp_forward(O.init_params(0), ep.cpu(), ev.cpu())
        assert (got.cpu() - ref).abs().max() < (3e-3 if precision == "bf16" else 4e-4)
    # the full static size works, one sample more is refused with a shape error and launches nothing
    ep, ev = torch.zeros(max_m, 63, device=DEV), torch.zeros(max_m, 27, device=DEV)
    out = torch.empty(max_m, 4, device=DEV)
    _C.check(lib.nerf_mlp_fwd_encoded_static(max_m, P(ep), P(ev), P(packed), prec, P(out), st), "full size")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert lib.nerf_mlp_fwd_encoded_static(max_m + 1, P(ep), P(ev), P(packed), prec, P(out), st) == -2
    assert lib.nerf_mlp_fwd_encoded_static(0, P(ep), P(ev), P(packed), prec, P(out), st) == 0
