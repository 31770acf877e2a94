#!/bin/bash
# usage: ab_libs.sh script.py lib1 lib2 ...   (runs the script once per library build)
s=$1; shift
for l in "$@"; do
  cp scratch/$l nerf_amd/lib/libnerf_hip.so || exit 1
  echo "== $l"
  timeout -k 10 150 python -u $s 2>&1 | grep -v amdgpu.ids || exit 1
done
