#!/bin/bash
# Timing-only builds of the bf16-compute weight gradient (TTK_BC_WDBG bits, csrc/bc_wgrad.hip); the wgrad column includes the fold launch
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== product"; python3 $R/tools/bench_bc.py 512 10 pw 2>&1 | grep -E "dw4_1|dw5_x|dw6" | cut -c1-28,112-150
for v in $(ls $R/tools/exp/_build/libttk_w[0-9]*.so 2>/dev/null); do
  echo "== $(basename $v)"; TTK_LIB=$v python3 $R/tools/bench_bc.py 512 10 pw 2>&1 | grep -E "dw4_1|dw5_x|dw6" | cut -c1-28,112-150
done
