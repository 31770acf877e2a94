"""Two ranks on one GPU over gloo: repeated all_reduce of a device tensor behind ~1 ms of GPU work (no nerf_amd code)."""
import os, sys, time, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
def say(m): print(f"[{rank}] {time.time() % 1000:.2f} {m}", file=sys.stderr, flush=True)
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
a = torch.randn(4096, 4096, device="cuda")
y = torch.full((1192000,), float(rank + 1), device="cuda")
for i in range(8):
    for _ in range(3):
        b = a @ a
    y.add_(1.0)
    dist.all_reduce(y)
    say(f"all_reduce {i}")
torch.cuda.synchronize(); say(f"value {y[0].item()}")
dist.barrier(); say("barrier")
dist.destroy_process_group(); say("done")
