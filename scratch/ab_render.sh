#!/bin/bash
# A/B two builds of the library on one box, training step and frame render: scratch/ab_render.sh a.so b.so [rounds]
set -e
for r in $(seq 1 ${3:-2}); do
  for v in $1 $2; do
    cp scratch/lib/$v nerf_amd/lib/libnerf_hip.so
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-one-eighth --no-default-flags --big-batch 0 --render-chunk 65536 > gpurun_out/ab.log 2>gpurun_out/ab.err
    python - "$v" <<PY
import json, sys
d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[1], "ms/step %.4f"%d["ms_per_step"], " ".join(f"{n.replace('mlp_','')}={k[n]['avg_us']:.1f}" for n in ("mlp_fwd_train","mlp_bwd_data","mlp_bwd_weights")),
      "render %.4f s/frame fwd16 %.1f TFLOP/s" % (d["render"][0]["s_per_frame"], d["render"][0]["kernels"]["mlp_fwd"]["TFLOP/s"]), "| timing:", d["roofline"]["timing"][:40])
PY
  done
done
