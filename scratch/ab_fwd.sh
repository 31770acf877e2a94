#!/bin/bash
# compare builds on the inference forward: scratch/ab_fwd.sh a.so b.so ...
set -e
for v in "$@"; do
  cp scratch/lib/$v nerf_amd/lib/libnerf_hip.so
  echo "== $v"; timeout -k 10 200 python scratch/time_fwd.py 2>&1 | grep -E "bf16 N=65536|bf16 N=4096"
done
