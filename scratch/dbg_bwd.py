import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from util import O, T, amd_model, maxabs
import nerf_amd as amd
DEV='cuda'
g = dict(np.load('tests/golden/f3_mlp.npz'))
P = 1024
pts, vd = T(g['pts'])[:P], T(g['viewdirs'])[:P]
gen = torch.Generator().manual_seed(0)
cot = torch.randn(P, 4, generator=gen) * 1e-3
params = {k: v.clone().requires_grad_(True) for k, v in O.init_params(0).items()}
raw = O.mlp_forward(params, O.positional_encode(pts, 10), O.positional_encode(vd, 4))
(raw * cot).sum().backward()
for prec in ('bf16',):
    m = amd_model(O.init_params(0), prec)
    r = m.forward_rays(pts.to(DEV), vd.to(DEV), torch.zeros(P, 1, device=DEV))
    (r.view(-1, 4) * cot.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    print(prec, 'fwd err', maxabs(r.view(-1,4), raw))
    for (k, p) in m.named_parameters():
        ref = params[k].grad
        got = p.grad.cpu()
        rel = (got - ref).norm() / (ref.norm() + 1e-30)
        print(f"  {k:28s} ref_norm {ref.norm():.3e} got_norm {got.norm():.3e} rel_err {rel:.3e} maxabs {maxabs(got, ref):.3e}")
