#!/bin/bash
# compare N builds of the library on one box (training step only): scratch/abn_train.sh rounds a.so b.so ...
set -e
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp scratch/lib/$v nerf_amd/lib/libnerf_hip.so
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-one-eighth --no-default-flags --big-batch 0 --no-render > gpurun_out/ab.log 2>gpurun_out/ab.err
    python - "$v" <<PY
import json, sys
d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[1], "ms/step %.4f"%d["ms_per_step"], " ".join(f"{n.replace('mlp_','')}={k[n]['avg_us']:.1f}/{k[n].get('avg_us_eager_events',0):.1f}" for n in ("mlp_fwd_train","mlp_bwd_data","mlp_bwd_weights")))
PY
  done
done
