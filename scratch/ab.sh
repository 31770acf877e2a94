#!/bin/bash
# A/B two builds of the library on one box: scratch/ab.sh a.so b.so [rounds]
set -e
for r in $(seq 1 ${3:-2}); do
  for v in $1 $2; do
    cp scratch/lib/$v nerf_amd/lib/libnerf_hip.so
    timeout -k 10 200 python bench.py --no-render --no-cpu-baseline > gpurun_out/ab.log 2>&1
    python - "$v" <<PY
import json, sys
d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[1], "ms/step %.4f"%d["ms_per_step"], " ".join(f"{n.replace('mlp_','')}={k[n]['avg_us']:.0f}" for n in ("mlp_fwd_train","mlp_bwd_data","mlp_bwd_weights")))
PY
  done
done
