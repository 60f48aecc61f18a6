#!/bin/bash
# Build variants of the fp16-split GEMM kernels: tools/exp/variants.sh name1 "flags1" name2 "flags2" ... builds
# tools/exp/_build/libttk_<name>.so from csrc/pwconv_f16.hip compiled with <flags> (e.g. "-DTTK_EXP=4", "-DTTK_RS=2 -DTTK_D=1");
# run with TTK_LIB=$PWD/tools/exp/_build/libttk_<name>.so python tools/bench_gemm.py
set -e
cd "$(dirname "$0")/../.."
CS=neuralnet-tracker-traincode_amd/csrc
make -s -C $CS -j8 >/dev/null
mkdir -p tools/exp/_build
rm -f tools/exp/_build/*.so tools/exp/_build/*.o
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 $flags -c $CS/pwconv_f16.hip -o tools/exp/_build/f16_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $CS/build/*.o | grep -v pwconv_f16.o) tools/exp/_build/f16_$name.o -o tools/exp/_build/libttk_$name.so &&
    rm tools/exp/_build/f16_$name.o ) &
done
wait
ls tools/exp/_build/
