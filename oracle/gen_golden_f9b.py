#!/usr/bin/env python3
"""Generate tests/golden/f9b_train_step_lindisp_white.npz by running the REFERENCE itself (development
container only): the training step of scripts/train_nerf.py:293-331 with the flags of the camera-path
benchmark configuration, ``lindisp=True`` and ``white_bkgd=True``, on injected uniforms.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/src python3 oracle/gen_golden_f9b.py

Same method as oracle/gen_golden.py (this file only adds one fixture and leaves the others untouched): the
reference's ``render_rays`` / ``sample_pdf`` run unmodified with ``torch.rand`` replaced by a queue of
pre-drawn tensors, the oracle is checked against them bit for bit, and inputs plus reference outputs are stored.
Weights are the seeded sets ``init_params(0)`` / ``init_params(1)`` pinned by SHA-256 in meta.json.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import nerf_oracle as O  # noqa: E402

from nerf_project.models.nerf import NeRF, PositionalEncoder, render_rays, sample_pdf  # noqa: E402

torch.set_num_threads(8)
rng = np.random.default_rng(4321)
U = lambda *shape: torch.from_numpy(rng.random(shape, dtype=np.float32))  # noqa: E731


class QueuedRand:
    """torch.rand hands out the queued tensors in order while the context is active"""

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


def same(a, b, what):
    if not torch.equal(torch.as_tensor(a), torch.as_tensor(b)):
        raise SystemExit(f"oracle != reference for {what}")


NT = 256
ro, rd = O.synthetic_rays(1024, seed=0)
ro, rd = ro[:NT], rd[:NT]
pc, pf = O.init_params(0), O.init_params(1)
pe10, pe4 = PositionalEncoder(3, 10), PositionalEncoder(3, 4)
t1, uu, t2, tgt = U(NT, 64), U(NT, 128), U(NT, 192), U(NT, 3)
mc, mf = NeRF(), NeRF()
mc.load_state_dict(pc); mf.load_state_dict(pf)
with QueuedRand([t1, uu, t2]):
    c = render_rays(mc, pe10, pe4, ro, rd, 2.0, 6.0, 64, rand=True, lindisp=True, white_bkgd=True)
    loss_c = torch.mean((c["rgb_map"] - tgt) ** 2)
    with torch.no_grad():
        zm = 0.5 * (c["z_vals"][..., 1:] + c["z_vals"][..., :-1])
        zs = sample_pdf(zm, c["weights"].detach()[..., 1:-1] + 1e-5, 128, det=False)
    zc2, _ = torch.sort(torch.cat([c["z_vals"].detach(), zs], -1), -1)
    f = render_rays(mf, pe10, pe4, ro, rd, 2.0, 6.0, 192, rand=True, lindisp=True, white_bkgd=True, z_vals_override=zc2)
    loss_f = torch.mean((f["rgb_map"] - tgt) ** 2)
(loss_c + loss_f).backward()
pco = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
pfo = {k: v.clone().requires_grad_(True) for k, v in pf.items()}
ot, olc, olf, oc, of = O.train_step_loss(pco, pfo, ro, rd, tgt, 2.0, 6.0, 64, 128, t1, uu, t2, lindisp=True, white_bkgd=True)
ot.backward()
same(olc, loss_c, "loss_c"); same(olf, loss_f, "loss_f")
same(oc["rgb_map"], c["rgb_map"], "coarse rgb"); same(of["rgb_map"], f["rgb_map"], "fine rgb")
out = {"t_rand_coarse": t1, "u_fine": uu, "t_rand_fine": t2, "target": tgt, "rays_o": ro, "rays_d": rd,
       "loss_c": loss_c.detach(), "loss_f": loss_f.detach(), "c_rgb_map": c["rgb_map"].detach(),
       "f_rgb_map": f["rgb_map"].detach(), "c_z_vals": c["z_vals"].detach(),
       "c_sigma_last": c["raw_sigma"][:, -1].detach(), "f_sigma_last": f["raw_sigma"][:, -1].detach()}
for tag, m, po in (("c", mc, pco), ("f", mf, pfo)):
    norms = []
    for k, p in m.named_parameters():
        same(po[k].grad, p.grad, f"grad {tag} {k}")
        norms.append(p.grad.double().norm().item())
    out[f"gnorm_{tag}"] = np.array(norms)
    out[f"g_{tag}_sigma_w"] = m.sigma_linear.weight.grad
    out[f"g_{tag}_rgb2_w"] = m.rgb_layers[2].weight.grad
    out[f"g_{tag}_l0_b"] = m.pts_linears[0].bias.grad
    out[f"g_{tag}_l5_w_head"] = m.pts_linears[5].weight.grad[:8, :80].clone()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "f9b_train_step_lindisp_white.npz"),
                    **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
print("f9b written: loss_c %.6f loss_f %.6f" % (loss_c.item(), loss_f.item()))
