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
        # free device memory as the driver reports it (hipMemGetInfo): a hipMalloc inside the call would show
        # here, which torch's own allocator statistics would not
        before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        _C.check(lib.nerf_mlp_fwd_encoded_static(M, P(ep), P(ev), P(packed), prec, P(got), st), "fwd_encoded_static")
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        if M != 5000 + 37 or precision != "bf16":      # (the very first call on the device may load the code object)
            assert torch.cuda.mem_get_info()[0] == free_before
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


def test_static_scratch_forward_inside_a_captured_graph():
    """The callers the entry point is for: after one call outside capture (which looks up the device's copy of
    the static array), the call is captured into a HIP graph -- where an allocation would fail the capture --
    and replayed on new inputs: results equal the workspace form's bit for bit."""
    from nerf_amd import _C, ops
    lib = _C.lib
    P = lambda t: ctypes.c_void_p(t.data_ptr())        # noqa: E731
    flat = torch.cat([v.reshape(-1) for v in O.init_params(1).values()]).to(DEV)
    packed = ops.mlp_pack(flat, "fp16")
    M = 2048 + 5
    gen = torch.Generator().manual_seed(23)
    ep, ev = torch.zeros(M, 63, device=DEV), torch.zeros(M, 27, device=DEV)
    got = torch.zeros(M, 4, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):      # warm-up outside capture, off the default stream
        _C.check(lib.nerf_mlp_fwd_encoded_static(M, P(ep), P(ev), P(packed), 1, P(got), ctypes.c_void_p(side.cuda_stream)), "warm-up")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _C.check(lib.nerf_mlp_fwd_encoded_static(M, P(ep), P(ev), P(packed), 1, P(got), st), "captured")
    for _ in range(2):
        pts = torch.rand(M, 3, generator=gen) * 6 - 3
        vd = torch.nn.functional.normalize(torch.randn(M, 3, generator=gen), dim=-1)
        ep.copy_(O.positional_encode(pts, 10)); ev.copy_(O.positional_encode(vd, 4))
        free_before = torch.cuda.mem_get_info()[0]
        graph.replay()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
        ws = torch.empty(int(lib.nerf_mlp_encoded_workspace_bytes(M)), dtype=torch.uint8, device=DEV)
        want = torch.empty(M, 4, device=DEV)
        _C.check(lib.nerf_mlp_fwd_encoded(M, P(ep), P(ev), P(packed), 1, P(want), None, P(ws),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "fwd_encoded")
        torch.cuda.synchronize()
        assert torch.equal(got, want)
