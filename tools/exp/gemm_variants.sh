#!/bin/bash
# Timing-only variants of pw_split_k (results are WRONG by construction).  usage: tools/exp/gemm_variants.sh N...
set -e
cd "$(dirname "$0")/../.."
C=neuralnet-tracker-traincode_amd/csrc
mkdir -p tools/exp/_build
OBJS=$(ls $C/build/*.o | grep -v pwconv_split.o)
build() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -w -I $C -DTTK_EXP=$1 -c $C/pwconv_split.hip -o tools/exp/_build/ps$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS tools/exp/_build/ps$1.o -o tools/exp/_build/libttk_exp$1.so
}
for n in "$@"; do build $n & done; wait
