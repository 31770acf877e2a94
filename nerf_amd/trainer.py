"""Training loop with the semantics scripts/train_nerf.py:272-383 intends.

The reference's DataLoader hands the step ``batch_size`` *images* worth of rays although
``--batch_size`` is documented as "number of rays per batch" (train_nerf.py:67, :163, :280-283).  This
trainer does what the option says: every step draws ``batch_rays`` pixels uniformly over all training
images (on the GPU, ``BlenderDataset.sample_rays``), runs the coarse + fine step
(train_nerf.py:293-336) from HIP graphs, decays the learning rate per global step (:288-290), logs
every ``log_freq`` steps (:342-357) and writes checkpoints in the reference's format (:109-137).
An "epoch" keeps the reference's meaning of one pass over the data: ceil(total rays / batch_rays) steps.
"""
from __future__ import annotations

import dataclasses
import math
import os
from typing import Any, Callable, Dict, Optional

import torch

from . import checkpoint as ckpt_io
from .models.nerf import NeRF, PositionalEncoder
from .optim import FlatAdam
from .parallel import broadcast_parameters_
from .train import TrainStep


@dataclasses.dataclass
class TrainConfig:
    """Options of train_nerf.py:49-85 that the loop uses, with the script's defaults (``use_hierarchical``
    and ``use_viewdirs`` are store_true flags there: off unless given)."""
    expname: str = "nerf"
    basedir: str = "./logs/"
    white_bkgd: bool = False
    near: float = 2.0
    far: float = 6.0
    N_samples_coarse: int = 64
    N_samples_fine: int = 128
    use_hierarchical: bool = False
    lindisp: bool = False
    num_epochs: int = 200
    batch_size: int = 1024               # rays per step and per rank
    lr: float = 5e-4
    lr_decay_factor: float = 0.1
    lr_decay_steps: int = 250000
    seed: int = 42
    net_depth: int = 8
    net_width: int = 256
    input_ch_pts_L: int = 10
    input_ch_views_L: int = 4
    use_viewdirs: bool = False
    log_freq: int = 100
    save_freq: int = 10                  # epochs
    precision: str = "bf16"


def mse_to_psnr(mse: float) -> float:
    """PSNR of an MSE on [0,1] colours, as the reference's logging reports it."""
    return -10.0 * math.log10(max(mse, 1e-10))


def learning_rate(cfg: TrainConfig, global_step: int) -> float:
    """train_nerf.py:288."""
    return cfg.lr * (cfg.lr_decay_factor ** (global_step / cfg.lr_decay_steps))


class Trainer:
    def __init__(self, dataset, cfg: TrainConfig, rank: int = 0, world_size: int = 1, device="cuda",
                 use_graph: bool = True, log: Optional[Callable[[str], None]] = print):
        self.cfg, self.data, self.rank, self.world = cfg, dataset, rank, world_size
        self.device = torch.device(device)
        self.log = log if rank == 0 else None
        torch.manual_seed(cfg.seed)                                       # train_nerf.py:149-152
        # encoders and networks as train_nerf.py:177-200 builds them
        self.pe = PositionalEncoder(3, cfg.input_ch_pts_L).to(self.device)
        self.pv = PositionalEncoder(3, cfg.input_ch_views_L).to(self.device) if cfg.use_viewdirs else None
        kw = dict(D=cfg.net_depth, W=cfg.net_width, input_ch_pts=self.pe.output_dims,
                  input_ch_views=self.pv.output_dims if cfg.use_viewdirs else 0, skips=[cfg.net_depth // 2],
                  use_viewdirs=cfg.use_viewdirs, precision=cfg.precision)
        self.model_coarse = NeRF(**kw).to(self.device)
        self.model_fine = NeRF(**kw).to(self.device) if cfg.use_hierarchical else None
        params = list(self.model_coarse.parameters())                     # train_nerf.py:203-206
        if self.model_fine is not None:
            params += list(self.model_fine.parameters())
        self.optimizer = FlatAdam(params, lr=cfg.lr)
        broadcast_parameters_(self.optimizer.flat, world_size)
        # every rank starts from rank 0's weights but draws its own jitter and inverse-CDF uniforms, so that an
        # N-rank step is one step over N * batch_size rays with independent draws
        torch.manual_seed(cfg.seed + rank)
        self.step_fn = TrainStep(self.model_coarse, self.model_fine, self.pe, self.pv, self.optimizer, cfg.batch_size,
                                 near=cfg.near, far=cfg.far, N_samples_coarse=cfg.N_samples_coarse,
                                 N_samples_fine=cfg.N_samples_fine, lindisp=cfg.lindisp, white_bkgd=cfg.white_bkgd,
                                 world_size=world_size, use_graph=use_graph)
        # every rank draws the same pixel ids and keeps its slice: one generator state per job
        self.sampler = torch.Generator(device=self.device)
        self.sampler.manual_seed(cfg.seed)
        self.global_step = 0
        self.epoch = 0
        self.steps_per_epoch = max(1, math.ceil(dataset.num_rays / (cfg.batch_size * world_size)))
        self.history: list = []

    # -- one optimisation step (train_nerf.py:280-339) -------------------------------------------------
    def train_step(self) -> torch.Tensor:
        step = self.step_fn       # the batch is gathered straight into the step's input buffers
        rays_o, rays_d, target = self.data.sample_rays(self.cfg.batch_size, self.sampler, self.rank, self.world,
                                                       out=(step.rays_o, step.rays_d, step.target))
        lr = learning_rate(self.cfg, self.global_step)
        loss = self.step_fn(rays_o, rays_d, target, lr=lr)
        self.global_step += 1
        if self.global_step % self.cfg.log_freq == 0:
            lc, lf = float(self.step_fn.loss_coarse), float(self.step_fn.loss_fine)     # the only host syncs
            rec = {"global_step": self.global_step, "lr": lr, "loss": lc + lf, "loss_coarse": lc,
                   "psnr_coarse": mse_to_psnr(lc)}
            msg = f"GS: {rec['global_step']} LR: {lr:.7f} Loss_T: {rec['loss']:.4f} PSNR_c: {rec['psnr_coarse']:.2f}"
            if self.model_fine is not None:                                # train_nerf.py:353-355
                rec.update(loss_fine=lf, psnr_fine=mse_to_psnr(lf))
                msg += f" Loss_f: {lf:.4f} PSNR_f: {rec['psnr_fine']:.2f}"
            self.history.append(rec)
            if self.log:
                self.log(msg)
        return loss

    def fit(self, num_epochs: Optional[int] = None, max_steps: Optional[int] = None) -> None:
        """Epochs [self.epoch, num_epochs); a checkpoint every ``save_freq`` epochs and a final one
        (train_nerf.py:272, :375-380).  ``max_steps`` stops early (after that many steps of this call)."""
        num_epochs = self.cfg.num_epochs if num_epochs is None else num_epochs
        done = 0
        while self.epoch < num_epochs:
            for _ in range(self.steps_per_epoch):
                self.train_step()
                done += 1
                if max_steps is not None and done >= max_steps:
                    return
            self.epoch += 1
            if self.epoch % self.cfg.save_freq == 0:
                self.save(self.epoch - 1)
        self.save(num_epochs - 1, is_final=True)

    # -- checkpoints (train_nerf.py:109-137, :219-260) --------------------------------------------------
    def checkpoint_dir(self) -> str:
        return os.path.join(self.cfg.basedir, self.cfg.expname, "checkpoints")

    def save(self, epoch: int, is_final: bool = False) -> Optional[str]:
        if self.rank != 0:
            return None
        args: Dict[str, Any] = dataclasses.asdict(self.cfg)
        common = dict(epoch=epoch, global_step=self.global_step, args=args, H=self.data.height, W=self.data.width,
                      focal=self.data.focal, model_coarse=self.model_coarse, model_fine=self.model_fine,
                      optimizer=self.optimizer)
        path = os.path.join(self.checkpoint_dir(), ckpt_io.checkpoint_name(epoch, self.global_step, is_final))
        ckpt_io.save_checkpoint(path, **common)
        if not is_final:
            ckpt_io.save_checkpoint(os.path.join(self.checkpoint_dir(), "latest.pth"), **common)
        if self.log:
            self.log(f"Checkpoint saved to {path}")
        return path

    def resume(self, path: str) -> None:
        """Continue from a checkpoint of either code base: weights, Adam moments and step counter,
        next epoch, global step.  A bare file name is looked up in this run's checkpoint directory."""
        if not os.path.exists(path):
            alt = os.path.join(self.checkpoint_dir(), path)
            if not os.path.exists(alt):
                raise FileNotFoundError(path)
            path = alt
        ck = ckpt_io.load_checkpoint(path, map_location=self.device)
        for model, key in ((self.model_coarse, "model_coarse_state_dict"), (self.model_fine, "model_fine_state_dict")):
            if model is None:
                continue
            if key not in ck:
                raise ValueError(f"{path} has no '{key}'")
            model.load_state_dict(ck[key])      # copies in place: the parameters stay views of the flat bucket
            model.invalidate_packed()
        self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        self.epoch = int(ck.get("epoch", -1)) + 1
        self.global_step = int(ck.get("global_step", 0))
