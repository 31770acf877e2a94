#!/bin/bash
# A/B of the training step with and without the merged launches, alternating on one box
for r in 1 2 3; do
  for v in "" "--no-joint"; do
    timeout -k 10 200 python bench.py --no-render --no-cpu-baseline --no-one-eighth --kernel-steps 3 $v > gpurun_out/ab.log 2>&1
    python - "joint$v" <<PY
import json, sys
d=json.loads(open("gpurun_out/ab.log").read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step", d["timed_regions"]["ms_per_step_each"], "4096:", d["train_4096_rays_per_gpu"]["ms_per_step_each"])
PY
  done
done
