#!/bin/bash
# per-kernel times of several builds (results of ablation builds are wrong; timing only)
for v in "$@"; do
  cp scratch/lib/$v nerf_amd/lib/libnerf_hip.so
  timeout -k 10 200 python bench.py --no-render --no-cpu-baseline --no-one-eighth > gpurun_out/ab.log 2>&1
  python - "$v" <<PY
import json, sys
try:
    d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
    k=d["kernels"]
    print(sys.argv[1], "ms/step %.4f"%d["ms_per_step"], " ".join(f"{n.replace('mlp_','')}={k[n]['avg_us']:.0f}" for n in ("mlp_fwd_train","mlp_bwd_data","mlp_bwd_weights")))
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/ab.log").read()[-300:])
PY
done
