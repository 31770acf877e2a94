"""Adam over one flat fp32 buffer (the optimizer.step() of scripts/train_nerf.py:203-206, :336).

The reference builds ``torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()))`` and
rewrites ``param_group['lr']`` every step (train_nerf.py:288-290).  ``FlatAdam`` keeps that surface
(``param_groups``, ``zero_grad``, ``step``) but stores all parameters, both moment buffers and the
gradients in single flat tensors, so that an optimizer step is one kernel launch and a data-parallel
gradient exchange is one all-reduce of one bucket.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

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

    def gather_grads(self, first: int = 0, last: Optional[int] = None) -> torch.Tensor:
        """Gradients of parameters [first, last) concatenated into their part of the flat bucket (one
        concatenation kernel); by default all of them.  Returns that part (a view of ``flat_grad``)."""
        last = len(self.params) if last is None else last
        gs = []
        for p in self.params[first:last]:
            if p.grad is None:
                raise RuntimeError("a parameter has no gradient; call backward() first")
            gs.append(p.grad.reshape(-1))
        lo, hi = sum(self.sizes[:first]), sum(self.sizes[:last])
        part = self.flat_grad[lo:hi]
        torch.cat(gs, out=part)
        return part

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
        """[lr, beta1, beta2, eps, grad_scale, step, ticket, -] on the GPU, created on first use (slot 6 is the
        update kernel's workgroup counter: it must be 0 before the first step and every step leaves it 0)."""
        if getattr(self, "_dev_state", None) is None:
            b1, b2 = self.betas
            self._dev_state = torch.tensor([self.param_groups[0]["lr"], b1, b2, self.eps, 1.0, float(self.steps), 0.0,
                                            0.0], dtype=torch.float32, device=self.flat.device)
            self._dev_lr = self.param_groups[0]["lr"]
        return self._dev_state

    def set_lr(self, lr: float) -> None:
        """The learning rate of the next step, on the host and in ``device_state()`` (the per-step decay of
        train_nerf.py:288-290 survives graph replay this way); an unchanged rate costs no launch."""
        self.param_groups[0]["lr"] = lr
        state = self.device_state()
        if lr != self._dev_lr:
            state[0:1].fill_(lr)
            self._dev_lr = lr

    @torch.no_grad()
    def step_dev(self, grads_gathered: bool = False) -> None:
        """Adam update reading lr / grad_scale / step from ``device_state()``; safe to capture in a graph.
        The step counter lives on the device in this mode (``steps`` on the host is not advanced)."""
        if not grads_gathered:
            self.gather_grads()
        ops.adam_step_dev(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.device_state())
        for p in self.params:
            torch._C._increment_version(p)

    # -- torch.optim.Adam's state_dict layout, so that checkpoints move both ways -----------------------
    def current_step(self) -> int:
        """Optimizer steps taken so far (from the device counter when the graph path is in use)."""
        dev = getattr(self, "_dev_state", None)
        return max(self.steps, int(dev[5].item())) if dev is not None else self.steps

    def state_dict(self):
        """Same layout as ``torch.optim.Adam.state_dict()`` (what train_nerf.py:126 stores): per-parameter
        ``step`` / ``exp_avg`` / ``exp_avg_sq`` under ``state`` and one entry in ``param_groups``."""
        step = float(self.current_step())
        state, off = {}, 0
        if step > 0:
            for i, (p, n) in enumerate(zip(self.params, self.sizes)):
                state[i] = {"step": torch.tensor(step), "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
                off += n
        group = {"lr": self.param_groups[0]["lr"], "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd) -> None:
        """Accepts the dict written by ``torch.optim.Adam`` (a reference checkpoint) or by this class."""
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("optimizer state does not match: expected one parameter group with "
                             f"{len(self.params)} parameters")
        g = groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("FlatAdam implements plain Adam (no weight decay, amsgrad or maximize)")
        self.param_groups[0]["lr"] = float(g["lr"])
        self.betas, self.eps = (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"])
        steps, off = set(), 0
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for i, (p, n) in enumerate(zip(self.params, self.sizes)):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is not None:
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state of parameter {i} has shape {tuple(st['exp_avg'].shape)}, "
                                     f"expected {tuple(p.shape)}")
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            off += n
        if len(steps) > 1:
            raise ValueError(f"parameters were stepped a different number of times: {sorted(steps)}")
        self.steps = steps.pop() if steps else 0
        if getattr(self, "_dev_state", None) is not None:
            b1, b2 = self.betas
            self._dev_state[:4].copy_(torch.tensor([self.param_groups[0]["lr"], b1, b2, self.eps]))
            self._dev_lr = self.param_groups[0]["lr"]
            self._dev_state[5] = float(self.steps)
