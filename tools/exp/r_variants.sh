#!/bin/bash
# Timing-only variants of the row-block GEMM (csrc/pwconv_r.hip, TTK_R_DBG bits; results are wrong by construction) on the wide layers:
#   bash tools/exp/r_variants.sh > gpurun_out/<dir>/r_variants.txt
for dbg in 0 1 2 3 4 8 16 17 19 23 31; do
  for r in 6 8; do
    echo "== TTK_R_DBG=$dbg TTK_R_RBLK=$r"
    TTK_R_DBG=$dbg TTK_R_RBLK=$r python tools/bench_gemm.py 512 10 2>&1 | grep -E "dw4_2|dw5_x|dw6 " | cut -c1-110
  done
done
echo "== old kernel"; TTK_GEMM_R=0 python tools/bench_gemm.py 512 10 2>&1 | grep -E "dw4_2|dw5_x|dw6 " | cut -c1-110
