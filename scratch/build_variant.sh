#!/bin/bash
# scratch/build_variant.sh <name> "<extra flags>" [files...]: compiles the given translation units (default: the chain
# kernels) with extra flags into build/var_<name>/ and links scratch/lib/<name>.so with the other objects of the build
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../nerf_amd/csrc"
O=../../build/obj; V=../../build/var_$NAME; mkdir -p $V ../../scratch/lib
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -mllvm -pragma-unroll-threshold=1000000 $FLAGS"
UNITS=${@:-mlp_fwd.tu0 mlp_fwd.tu1 mlp_fwd.tu2 mlp_bwd}
pids=()
for u in $UNITS; do
  case $u in
    mlp_fwd.tu*) $CXX -DNERF_TU=${u#mlp_fwd.tu} -c mlp_fwd.hip -o $V/$u.o & ;;
    *) $CXX -c $u.hip -o $V/$u.o & ;;
  esac
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for u in api rays composite sample_pdf adam loss mlp_pack mlp_fwd16 mlp_bwd mlp_fp32 mlp_p16 mlp_fwd.tu0 mlp_fwd.tu1 mlp_fwd.tu2; do
  if [ -f $V/$u.o ]; then OBJS="$OBJS $V/$u.o"; else OBJS="$OBJS $O/$u.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -o ../../scratch/lib/$NAME.so
echo "built scratch/lib/$NAME.so"
