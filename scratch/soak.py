"""Determinism soak: the same seeded training run twice, and the same frame rendered repeatedly, must agree bit for bit."""
import sys, hashlib, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from dp_gpu_worker import run_steps
from util import O, amd_model
import nerf_amd
from nerf_amd.render import render_image
from nerf_amd.utils.synthetic import focal_from_fov, pose_spherical

def h(t): return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
t0 = time.time()
a, la = run_steps(1, 0, "same", n=1024, steps=steps)
print("run 1", h(a), la[-1], "%.1f s" % (time.time() - t0), flush=True)
b, lb = run_steps(1, 0, "same", n=1024, steps=steps)
print("run 2", h(b), lb[-1], flush=True)
print("training bit-identical:", torch.equal(a, b), "losses identical:", la == lb, "finite:", bool(torch.isfinite(a).all()), flush=True)
for prec in ("bf16", "fp16"):
    mc, mf = amd_model(O.init_params(0), prec), amd_model(O.init_params(1), prec)
    pe, pv = nerf_amd.PositionalEncoder(3, 10).cuda(), nerf_amd.PositionalEncoder(3, 4).cuda()
    c2w = pose_spherical(40.0, -30.0, 4.0).cuda()
    hs = set()
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
        img = render_image(mc, mf, pe, pv, 800, 800, focal_from_fov(800), c2w, chunk=65536 if i % 2 == 0 else 4096)
        hs.add(h(img))
    print(prec, "render hashes over repeats and both chunk sizes:", hs, flush=True)
