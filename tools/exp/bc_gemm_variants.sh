#!/bin/bash
# Timing-only builds of the bf16-compute streamed GEMM (TTK_BC_GDBG bits, csrc/bc_gemm.hip) on the layer shapes (the L-kernel rows: 128 -> 256 and wider)
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== product"; python3 $R/tools/bench_bc.py 512 10 pw 2>&1 | grep -E "dw4_1|dw5_x|dw6|totals"
for v in $(ls $R/tools/exp/_build/libttk_g*.so 2>/dev/null); do
  echo "== $(basename $v)"; TTK_LIB=$v python3 $R/tools/bench_bc.py 512 10 pw 2>&1 | grep -E "dw4_1|dw5_x|dw6|totals"
done
