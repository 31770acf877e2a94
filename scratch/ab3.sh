#!/bin/bash
# alternate several builds of the library on one box: scratch/ab3.sh rounds a.so b.so c.so ...
set -e
n=$1; shift
for r in $(seq 1 $n); do
  for v in "$@"; do
    cp scratch/lib/$v nerf_amd/lib/libnerf_hip.so
    timeout -k 10 200 python bench.py --no-render --no-cpu-baseline --no-one-eighth --no-default-flags > gpurun_out/ab.log 2>&1
    python - "$v" <<PY
import json, sys
d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[1], "ms/step %.4f"%d["ms_per_step"], "4096: %.4f"%d["train_4096_rays_per_gpu"]["ms_per_step"], " ".join(f"{n.replace('mlp_','')}={k[n]['avg_us']:.0f}" for n in ("mlp_fwd_train","mlp_bwd_data","mlp_bwd_weights")))
PY
  done
done
