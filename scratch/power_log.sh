#!/bin/bash
# samples power / clocks with rocm-smi while bench.py trains and renders
cd "$GRAFT_REPO_ROOT"
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "\(W\)|sclk" | tr '\n' ';'; echo; sleep 0.5; done ) > gpurun_out/power_log.txt &
LOGGER=$!
timeout -k 10 200 python bench.py --no-one-eighth --no-cpu-baseline --steps 4000 --warmup 10 > gpurun_out/b_power.json 2> gpurun_out/b_power.err
wait $LOGGER
cat gpurun_out/power_log.txt | cut -c1-300 | head -40
