"""Train on an analytic scene (a shaded sphere) through the loader + trainer + renderer; held-out PSNR."""
import json, math, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import O
from nerf_amd.utils.png import write_png
from nerf_amd.datasets import BlenderDataset
from nerf_amd.trainer import TrainConfig, Trainer
from nerf_amd.render import evaluate

def sphere_image(H, W, focal, c2w):
    ro, rd = O.get_rays(H, W, focal, c2w)
    ro, rd = ro.reshape(-1, 3).double(), rd.reshape(-1, 3).double()
    b = (ro * rd).sum(-1); c = (ro * ro).sum(-1) - 1.0
    disc = b * b - c
    hit = disc > 0
    t = -b - torch.sqrt(disc.clamp_min(0))
    p = ro + rd * t[:, None]
    col = 0.5 + 0.5 * p                                   # colour = position on the unit sphere
    img = torch.where(hit[:, None], col, torch.zeros_like(col))
    a = hit.double()[:, None]
    return torch.cat([img, a], -1).reshape(H, W, 4)

def make_scene(root, n_train=12, H=64, W=64):
    ang = 0.6911112070083618
    focal = 0.5 * W / math.tan(0.5 * ang)
    for split, thetas in (("train", [360.0 * i / n_train for i in range(n_train)]), ("val", [15.0, 195.0])):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for i, th in enumerate(thetas):
            c2w = O.pose_spherical(th, -30.0 if i % 2 == 0 else -50.0, 4.0)
            img = (sphere_image(H, W, focal, c2w) * 255.0 + 0.5).clamp(0, 255).to(torch.uint8).numpy()
            write_png(os.path.join(root, split, f"r_{i}.png"), img)
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": c2w.double().tolist()})
        json.dump({"camera_angle_x": ang, "frames": frames}, open(os.path.join(root, f"transforms_{split}.json"), "w"))

root = tempfile.mkdtemp()
make_scene(root)
train, val = BlenderDataset(root, "train"), BlenderDataset(root, "val")
for prec in ("bf16", "fp16"):
    tr = Trainer(train, TrainConfig(basedir=root, expname="x" + prec, batch_size=1024, precision=prec, log_freq=500, use_hierarchical=True,
                                    use_viewdirs=True), log=print)
    kw = dict(chunk=65536)
    p0 = evaluate(tr.model_coarse, tr.model_fine, tr.pe, tr.pv, val, **kw)["psnr"]
    t0 = time.time()
    tr.fit(num_epochs=10**6, max_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
    torch.cuda.synchronize(); dt = time.time() - t0
    p1 = evaluate(tr.model_coarse, tr.model_fine, tr.pe, tr.pv, val, **kw)["psnr"]
    print(prec, "held-out PSNR before", p0, "after", p1, "train time %.1f s" % dt, flush=True)
