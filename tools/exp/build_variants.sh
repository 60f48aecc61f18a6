#!/bin/bash
# Experiment builds of libttk_hip.so: one translation unit recompiled with -D flags, linked with the product's other objects.
#   bash tools/exp/build_variants.sh <name> <source.hip> "<-D flags>" [<name> <source.hip> "<flags>" ...]
# -> tools/exp/_build/libttk_<name>.so   (select with TTK_LIB=...; never shipped, never loaded by the product)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neuralnet-tracker-traincode_amd/csrc
O=$R/tools/exp/_build
mkdir -p $O
make -C $C -j8 > /dev/null
FLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DTTK_EXPERIMENTS"  # (the A/B environment switches of the recompiled file exist in experiment builds only)
pids=()
while [ $# -ge 3 ]; do
  name=$1; src=$2; defs=$3; shift 3
  (
    base=$(basename $src .hip)
    /opt/rocm/bin/hipcc $FLAGS $defs -I$R/include -c $C/$src -o $O/${base}_$name.o
    objs=$(ls $C/build/*.o | grep -v "/$base.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $O/${base}_$name.o -o $O/libttk_$name.so
    echo "built $O/libttk_$name.so ($defs)"
  ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
