"""The training step of scripts/train_nerf.py:288-336 as one callable, optionally replayed from a HIP graph.

A 1024-ray step is 25 short kernel launches; issued one by one from Python the host is
the bottleneck.  ``TrainStep`` runs the same call sequence eagerly or captures it once into HIP
graphs and replays them, with the inputs in static buffers and the optimizer's hyper-parameters in
device memory so that the learning rate can change between replays.  A single process captures the
whole step (forward, backward, gradient bucket, optimizer, weight re-pack) as one graph (a graph
boundary idles the GPU for about 9 us).  Data-parallel training needs the summing all-reduce of the flat gradient
bucket, which cannot sit inside a graph, and splits the step where the collectives go: the backward
pass of the fine network comes first and does not depend on the coarse one (the fine sample positions
are not differentiated, train_nerf.py:318), so its half of the bucket is reduced while the coarse
network's backward pass runs:

    graph 1: forward (coarse, fine), losses, fine backward      -> all-reduce(fine half), asynchronous
    graph 2: coarse backward                                    -> all-reduce(coarse half)
    graph 3: Adam on the whole bucket (scaled by 1/world), weight re-pack

The static input buffers are ``TrainStep.rays_o / rays_d / target``: a producer that writes a batch
straight into them (``BlenderDataset.sample_rays(out=...)``) and passes those same tensors to the
call skips the three device copies.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .models.nerf import render_hierarchical, render_rays
from .optim import FlatAdam
from .parallel import all_reduce_async_, allreduce_gradients_


class TrainStep:
    def __init__(self, model_coarse, model_fine, embed_fn_pts, embed_fn_views, optimizer: FlatAdam, n_rays: int,
                 near: float = 2.0, far: float = 6.0, N_samples_coarse: int = 64, N_samples_fine: int = 128,
                 lindisp: bool = False, white_bkgd: bool = False, world_size: int = 1, use_graph: bool = True,
                 external_uniforms: bool = False, joint: bool = True):
        """``model_fine=None`` is the coarse-only run of train_nerf.py:303 (no ``--use_hierarchical``);
        ``embed_fn_views=None`` (or a model built with ``use_viewdirs=False``) the run without
        ``--use_viewdirs`` (train_nerf.py:79, :181-184).  With ``external_uniforms`` the step reads its
        uniform draws from the static buffer ``self.uniforms`` instead of generating them (parity tests
        inject the reference's draws there): [n, Nc] jitter of the coarse pass, then [n, Nf] inverse-CDF
        uniforms and [n, Nc+Nf] jitter of the fine pass, flattened and concatenated.  ``joint=False`` keeps
        the per-network launches even where the merged ones apply (``_joint_layout``); the two orders of
        summation give weight gradients that differ in the last bits."""
        self.mc, self.mf, self.pe, self.pv, self.opt = model_coarse, model_fine, embed_fn_pts, embed_fn_views, optimizer
        self.near, self.far, self.nc, self.nf = near, far, N_samples_coarse, N_samples_fine
        self.lindisp, self.white = lindisp, white_bkgd
        self.world = world_size
        self.use_graph = use_graph
        self.hier = model_fine is not None
        self.use_viewdirs = embed_fn_views is not None and bool(getattr(model_coarse, "use_viewdirs", True))
        dev = optimizer.flat.device
        self.n_uniforms = n_rays * ((2 * N_samples_coarse + 2 * N_samples_fine) if self.hier else N_samples_coarse)
        self.uniforms = torch.zeros((self.n_uniforms,), device=dev) if external_uniforms else None
        self.rays_o = torch.zeros((n_rays, 3), device=dev)
        self.rays_d = torch.zeros((n_rays, 3), device=dev)
        self.target = torch.zeros((n_rays, 3), device=dev)
        self.losses = torch.zeros((3,), device=dev)           # total, coarse, fine: written by the loss kernel
        self.loss, self.loss_coarse, self.loss_fine = self.losses[0], self.losses[1], self.losses[2]
        self._g_fb: Optional[torch.cuda.CUDAGraph] = None
        self._g_coarse: Optional[torch.cuda.CUDAGraph] = None
        self._g_opt: Optional[torch.cuda.CUDAGraph] = None
        # parameters of the fine network as a range [first, last) of the optimizer's list, if they form one
        fine_ids = {id(p) for p in model_fine.parameters()} if self.hier else set()
        idx = [i for i, p in enumerate(optimizer.params) if id(p) in fine_ids]
        n_par = len(optimizer.params)
        self._fine = (idx[0], idx[-1] + 1) if 0 < len(idx) < n_par and idx == list(range(idx[0], idx[-1] + 1)) else None
        if self._fine is not None and not (self._fine[0] == 0 or self._fine[1] == n_par):
            self._fine = None          # (the coarse parameters must be one range too)
        self.opt.device_state()[4] = 1.0 / world_size        # gradients are summed over ranks, then averaged
        self._joint = self._joint_layout() if joint else None
        if self._joint is not None:
            nb = int(ops.lib.nerf_mlp_packed_bytes())
            self._packed_bufs = [torch.empty((nb,), dtype=torch.uint8, device=dev) for _ in self._joint]

    def _joint_layout(self):
        """[(model, first float, last float)] when the step can pack the networks in one launch and compute
        their weight gradients in one launch sequence straight into the optimizer's bucket: a single process
        (data-parallel training keeps the two backward passes apart to overlap the first all-reduce), fused
        16-bit networks of one precision, each network's parameters one range of the bucket in state_dict
        order.  Else None: the networks are packed and differentiated one by one."""
        from .models.nerf import NeRF
        models = [self.mc] + ([self.mf] if self.hier else [])
        if self.world != 1 or not all(isinstance(m, NeRF) and m.fused() for m in models):
            return None
        if len({m.precision for m in models}) != 1:
            return None
        ids = [id(p) for p in self.opt.params]
        layout = []
        for m in models:
            ps = m.ordered_params()
            try:
                first = ids.index(id(ps[0]))
            except ValueError:
                return None
            if ids[first:first + len(ps)] != [id(p) for p in ps]:
                return None
            lo = sum(self.opt.sizes[:first])
            layout.append((m, lo, lo + sum(self.opt.sizes[first:first + len(ps)])))
        if sum(hi - lo for _, lo, hi in layout) != self.opt.flat.numel():
            return None                 # the optimizer holds parameters of something else as well
        return layout

    # the step body: losses of train_nerf.py:299, :330-331; gradients into the optimizer's flat bucket
    def _forward(self):
        # the three uniform draws of a step (train_nerf.py:296, :317, :327) come from one generator call
        n, nc, nf = self.rays_o.shape[0], self.nc, self.nf
        r = self.uniforms
        if r is None:
            r = torch.rand((self.n_uniforms,), dtype=torch.float32, device=self.rays_o.device)
        if not self.hier:           # train_nerf.py:293-303: total_loss = loss_coarse
            coarse = render_rays(self.mc, self.pe, self.pv, self.rays_o, self.rays_d, self.near, self.far, nc,
                                 rand=True, lindisp=self.lindisp, use_viewdirs=self.use_viewdirs,
                                 white_bkgd=self.white, t_rand=r.view(n, nc))
            _, d_coarse, _ = ops.mse_pair_grads(coarse["rgb_map"], None, self.target, out=self.losses)
            self.opt.zero_grad(set_to_none=True)
            return coarse["rgb_map"], None, d_coarse, None
        t_c, u_f, t_f = (r[:n * nc].view(n, nc), r[n * nc:n * (nc + nf)].view(n, nf),
                         r[n * (nc + nf):].view(n, nc + nf))
        coarse, fine = render_hierarchical(self.mc, self.mf, self.pe, self.pv, self.rays_o, self.rays_d, self.near,
                                           self.far, self.nc, self.nf, rand=True, det=False, lindisp=self.lindisp,
                                           use_viewdirs=self.use_viewdirs, white_bkgd=self.white, t_rand_coarse=t_c,
                                           u_fine=u_f, t_rand_fine=t_f)
        # loss.backward() of train_nerf.py:334 with the unit root gradient folded away: the loss kernel
        # hands back d loss / d rgb of both maps, which seed the backward pass directly
        _, d_coarse, d_fine = ops.mse_pair_grads(coarse["rgb_map"], fine["rgb_map"], self.target, out=self.losses)
        self.opt.zero_grad(set_to_none=True)
        return coarse["rgb_map"], fine["rgb_map"], d_coarse, d_fine

    def _forward_backward(self):
        if self._joint is not None:
            return self._forward_backward_joint()
        rgb_c, rgb_f, d_c, d_f = self._forward()
        if rgb_f is None:
            torch.autograd.backward([rgb_c], [d_c])
        else:
            torch.autograd.backward([rgb_c, rgb_f], [d_c, d_f])
        self.opt.gather_grads()

    def _forward_backward_joint(self):
        """The same step with one pack launch for all weight streams and the weight gradients written straight
        into the optimizer's gradient bucket (no concatenation).  The weight-gradient launches stay one per
        network: a merged launch (ops.mlp_bwd_weights_multi) halves the partial sums, but each network's
        backward-data output is still partly in the last-level cache when its own weight-gradient launch
        follows at once, and on one box the two orders measured equal within noise at 1024 and 4096 rays."""
        models = [m for m, _, _ in self._joint]
        sinks = [self.opt.flat_grad[lo:hi] for _, lo, hi in self._joint]
        ops.mlp_pack_multi([self.opt.flat[lo:hi] for _, lo, hi in self._joint], models[0].precision, ops.PACK_TRAIN,
                           [m.use_viewdirs for m in models], out=self._packed_bufs)
        try:
            for m, buf in zip(models, self._packed_bufs):
                m._external_pack = buf
            rgb_c, rgb_f, d_c, d_f = self._forward()
            ops.GRAD_SINKS = {buf.data_ptr(): g for buf, g in zip(self._packed_bufs, sinks)}
            if rgb_f is None:
                torch.autograd.backward([rgb_c], [d_c])
            else:
                torch.autograd.backward([rgb_c, rgb_f], [d_c, d_f])
        finally:
            ops.GRAD_SINKS = None
            for m in models:
                m._external_pack = None

    # the same, split where the data-parallel collectives go
    def _forward_backward_fine(self):
        rgb_c, rgb_f, d_c, d_f = self._forward()
        torch.autograd.backward([rgb_f], [d_f])
        self._coarse_root = (rgb_c, d_c)
        return self.opt.gather_grads(*self._fine)

    def _backward_coarse(self):
        rgb_c, d_c = self._coarse_root
        self._coarse_root = None
        torch.autograd.backward([rgb_c], [d_c])
        first, last = self._fine
        return self.opt.gather_grads(0, first) if first > 0 else self.opt.gather_grads(last, None)

    def _optimizer(self):
        self.opt.step_dev(grads_gathered=True)

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up off the capture stream, as graph capture requires;
            for _ in range(2):                 # the optimizer is not stepped here
                self._forward_backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # other threads of the process (the collective library's watchdog polls events) must not be able to
        # invalidate the capture: only this thread's calls are checked
        MODE = "thread_local"
        self._g_fb = torch.cuda.CUDAGraph()
        if self.world == 1:
            with torch.cuda.graph(self._g_fb, capture_error_mode=MODE):
                self._forward_backward()
                self._optimizer()
            return
        pool = torch.cuda.graph_pool_handle()      # graph 2 reads what graph 1 saved: one memory pool
        if self._fine is not None:
            with torch.cuda.graph(self._g_fb, pool=pool, capture_error_mode=MODE):
                self._part_fine = self._forward_backward_fine()
            self._g_coarse = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_coarse, pool=pool, capture_error_mode=MODE):
                self._part_coarse = self._backward_coarse()
        else:
            with torch.cuda.graph(self._g_fb, pool=pool, capture_error_mode=MODE):
                self._forward_backward()
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt, pool=pool, capture_error_mode=MODE):
            self._optimizer()
        # capture only records: nothing ran, so neither the weights nor the step counter moved

    def prepare(self) -> None:
        """Capture the graphs now instead of on the first call.  Capturing runs the forward and backward
        a few times on the current buffers (consuming random numbers) but does not step the optimizer."""
        if self.use_graph and self._g_fb is None:
            self._capture()

    def __call__(self, rays_o: torch.Tensor, rays_d: torch.Tensor, target_rgb: torch.Tensor,
                 lr: Optional[float] = None) -> torch.Tensor:
        """One optimisation step on this rank's rays; returns the (device) total loss of this rank."""
        for dst, src in ((self.rays_o, rays_o), (self.rays_d, rays_d), (self.target, target_rgb)):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        if lr is not None:
            self.opt.set_lr(lr)
        if self.use_graph:
            if self._g_fb is None:
                self._capture()
            self._g_fb.replay()
            if self._g_coarse is not None:
                fine_reduced = all_reduce_async_(self._part_fine)       # overlaps with the coarse backward
                self._g_coarse.replay()
                coarse_reduced = all_reduce_async_(self._part_coarse)
                fine_reduced.wait()
                coarse_reduced.wait()
                self._g_opt.replay()
            elif self._g_opt is not None:
                allreduce_gradients_(self.opt.flat_grad, self.world)
                self._g_opt.replay()
        else:
            self._forward_backward()
            allreduce_gradients_(self.opt.flat_grad, self.world)
            self._optimizer()
        # A graph replay rewrites the weights on the device without touching the parameters' version
        # counters, which the models' no_grad cache of packed weights is keyed on: drop the caches.
        self.mc.invalidate_packed()
        if self.mf is not None:
            self.mf.invalidate_packed()
        return self.loss
