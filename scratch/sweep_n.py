import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from dp_gpu_worker import run_steps
for n in (1, 7, 100, 1000, 1023, 4096, 10000):
    flat, losses = run_steps(1, 0, "same", n=n, steps=5)
    print(n, losses[0], losses[-1], bool(torch.isfinite(flat).all()), flush=True)
