#!/bin/bash
# kernel-trace of a short training-only bench; prints the MLP helper kernels' average durations
set -e
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace_small; rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-one-eighth --no-render --kernel-steps 2 > $O/out.json 2> $O/err.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$O/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if "nerf::" in n or "Cat" in n or "distribution" in n:
            print("%-70s calls %5s avg %8.1f us" % (n[:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
