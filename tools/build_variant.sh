#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags, e.g. -DB3GS_SCAN_BATCH=4>" [file.hip ...]
# builds tools/ab/<name>.so: the in-tree objects with the named files recompiled under the extra flags (A/B runs: tools/ab.sh)
set -e
name=$1; extra=$2; shift 2
cd "$(dirname "$0")/../binocular3dgs_amd/csrc"
mkdir -p ../../tools/ab /tmp/ab_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-fast-math -mllvm -amdgpu-atomic-optimizer-strategy=None -Wall -Wno-unused-function"
objs=""
for f in api preprocess binning render optim loss lossfn densify knn; do
  if [[ " $* " == *" $f.hip "* ]]; then
    /opt/rocm/bin/hipcc $FLAGS $extra -c $f.hip -o /tmp/ab_$name/$f.o
    objs="$objs /tmp/ab_$name/$f.o"
  else
    objs="$objs $f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/$name.so $objs
echo built tools/ab/$name.so
