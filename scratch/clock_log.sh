#!/bin/bash
# Shader / fabric / memory clocks while (1) the training step replays, (2) the bare copy schedule runs:
# scratch/clock_log.sh > gpurun_out/clock_log.txt
poll() {  # poll rocm-smi while the process $1 lives
  while kill -0 $1 2>/dev/null; do
    /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -E "sclk|fclk|mclk|socclk" | sed 's/ clock level//' | tr '\n' ' '; echo
    sleep 0.2
  done
}
echo "== training steps (graph replay, 1024 rays)"
timeout -k 10 200 python bench.py --no-cpu-baseline --no-one-eighth --no-default-flags --big-batch 0 --no-render --steps 4000 --repeats 1 --kernel-steps 2 > /dev/null 2>&1 &
poll $!
echo "== copy schedule only (dma_rate)"
( for i in $(seq 1 12); do timeout -k 5 60 ./scratch/bin/dma_rate > /dev/null 2>&1; done ) &
poll $!
echo "== idle"
/opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -E "sclk|fclk|mclk|socclk" | tr '\n' ' '; echo
