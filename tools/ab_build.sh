#!/bin/bash
# Build a VARIANT of libmicformer_hip.so for A/B runs on one box: tools/ab_build.sh NAME "-DFLAG ..." [file.hip ...]
#   -> ab/libmicformer_NAME.so (only the listed sources are recompiled with the extra flags; the rest come from csrc/build/*.o)
# Use with MICF_LIB=$PWD/ab/libmicformer_NAME.so.  ab/ holds build artefacts only (*.so is git-ignored, but travels with gpurun).
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/ab/obj_$NAME
OBJS=""
for o in $ROOT/micformer_amd/csrc/build/*.o; do
  b=$(basename $o .o)
  use=$o
  for f in "$@"; do
    if [ "$(basename $f .hip)" = "$b" ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -fvisibility=default $FLAGS -c $ROOT/micformer_amd/csrc/$b.hip -o $ROOT/ab/obj_$NAME/$b.o &
      use=$ROOT/ab/obj_$NAME/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab/libmicformer_$NAME.so $OBJS
echo "built ab/libmicformer_$NAME.so"
