"""Adam over one flat fp32 buffer (the optimizer.step() of scripts/train_nerf.py:203-206, :336).

The reference builds ``torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()))`` and
rewrites ``param_group['lr']`` every step (train_nerf.py:288-290).  ``FlatAdam`` keeps that surface
(``param_groups``, ``zero_grad``, ``step``) but stores all parameters, both moment buffers and the
gradients in single flat tensors, so that an optimizer step is one kernel launch and a data-parallel
gradient exchange is one all-reduce of one bucket.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import ops


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam got an empty parameter list")
        dev = self.params[0].device
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("FlatAdam needs fp32 parameters on one GPU")
        self.sizes = [p.numel() for p in self.params]
        with torch.no_grad():
            self.flat = torch.cat([p.detach().reshape(-1) for p in self.params])
            off = 0
            for p, n in zip(self.params, self.sizes):
                p.data = self.flat[off:off + n].view(p.shape)     # parameters become views of the bucket
                off += n
        self.flat_grad = torch.zeros_like(self.flat)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = 0
        self.betas, self.eps = betas, eps
        self.param_groups = [{"params": self.params, "lr": lr}]

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def gather_grads(self) -> torch.Tensor:
        """All gradients as one contiguous bucket (one concatenation kernel)."""
        gs = []
        for p in self.params:
            if p.grad is None:
                raise RuntimeError("a parameter has no gradient; call backward() first")
            gs.append(p.grad.reshape(-1))
        torch.cat(gs, out=self.flat_grad)
        return self.flat_grad

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0, grads_gathered: bool = False) -> None:
        """Adam update.  ``grad_scale`` multiplies the gradients first (1/world_size after a summing
        all-reduce); ``grads_gathered`` says ``flat_grad`` already holds them."""
        if not grads_gathered:
            self.gather_grads()
        self.steps += 1
        b1, b2 = self.betas
        ops.adam_step(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.param_groups[0]["lr"], b1, b2,
                      self.eps, self.steps, grad_scale)
        for p in self.params:                       # in-place update behind autograd's back: bump versions
            torch._C._increment_version(p)

    # -- device-resident hyper-parameters (HIP-graph friendly) -----------------------------------------
    def device_state(self) -> torch.Tensor:
        """[lr, beta1, beta2, eps, grad_scale, step, -, -] on the GPU, created on first use."""
        if getattr(self, "_dev_state", None) is None:
            b1, b2 = self.betas
            self._dev_state = torch.tensor([self.param_groups[0]["lr"], b1, b2, self.eps, 1.0, float(self.steps), 0.0,
                                            0.0], dtype=torch.float32, device=self.flat.device)
        return self._dev_state

    @torch.no_grad()
    def step_dev(self, grads_gathered: bool = False) -> None:
        """Adam update reading lr / grad_scale / step from ``device_state()``; safe to capture in a graph.
        The step counter lives on the device in this mode (``steps`` on the host is not advanced)."""
        if not grads_gathered:
            self.gather_grads()
        ops.adam_step_dev(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.device_state())
        for p in self.params:
            torch._C._increment_version(p)

    def state_dict(self):
        return {"steps": self.steps, "lr": self.param_groups[0]["lr"], "betas": self.betas, "eps": self.eps,
                "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone()}

    def load_state_dict(self, sd) -> None:
        self.steps = int(sd["steps"])
        self.param_groups[0]["lr"] = float(sd["lr"])
        self.betas, self.eps = tuple(sd["betas"]), float(sd["eps"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
