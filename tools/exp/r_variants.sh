#!/bin/bash
# Timing-only builds of the row-block GEMMs (csrc/pwconv_r.hip, compile-time TTK_R_DBG bits - the bit table is in that file and in
# profiles/r04_rowblock_stamps.txt; results are wrong by construction).  Build here, run on the GPU box:
#   bash tools/exp/r_variants.sh build                       # -> tools/exp/_build/libttk_d<bits>.so, libttk_s<bits>.so (with stamps)
#   bash tools/exp/r_variants.sh run > gpurun_out/<dir>/r_variants.txt
# TTK_R_MERGED=0 / 2 in EXTRA selects the twelve-wave / the eight-wave kernel for both directions (default: 1, per direction).
BITS="0 1 2 4 8 19 23 51 83 147 211"
EXTRA=${EXTRA:-}
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  args=()
  for b in $BITS; do args+=(d$b pwconv_r.hip "-DTTK_R_DBG=$b $EXTRA" s$b pwconv_r.hip "-DTTK_R_STAMP -DTTK_R_DBG=$b $EXTRA"); done
  bash tools/exp/build_variants.sh "${args[@]}"
else
  for b in $BITS; do
    echo "== TTK_R_DBG=$b"
    TTK_LIB=$PWD/tools/exp/_build/libttk_d$b.so python tools/bench_gemm.py 512 10 2>&1 | grep -E "dw4_2|dw5_x|dw5_6|dw6 " | cut -c1-110
    TTK_LIB=$PWD/tools/exp/_build/libttk_s$b.so python tools/exp/r_stamps.py 512 2>&1 | grep -A3 "dw5_x\|dw6 " | grep -v "first wave"
  done
fi
