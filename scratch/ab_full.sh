#!/bin/bash
# per-kernel times and frame render of several builds: scratch/ab_full.sh a.so b.so ...
for v in "$@"; do
  cp scratch/lib/$v nerf_amd/lib/libnerf_hip.so
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-one-eighth --big-batch 0 --no-default-flags --repeats 2 --render-chunk 65536 > gpurun_out/ab.log 2>&1
  python - "$v" <<PY
import json, sys
try:
    d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
    k=d["kernels"]
    print(sys.argv[1], "ms/step %.4f"%d["ms_per_step"], " ".join(f"{n.replace('mlp_','')}={k[n]['avg_us']:.0f}" for n in ("mlp_fwd_train","mlp_bwd_data","mlp_bwd_weights")), "render s/frame", d["render"][0]["s_per_frame"], "fwd16 TF", d["render"][0]["kernels"]["mlp_fwd"]["TFLOP/s"])
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/ab.log").read()[-300:])
PY
done
