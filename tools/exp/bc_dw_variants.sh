#!/bin/bash
# Timing-only builds of the bf16-compute depthwise backward (TTK_BC_DBG bits, csrc/bc_dw.hip) on the layer shapes:
#   (container) bash tools/exp/build_variants.sh bcd1 bc_dw.hip "-DTTK_BC_DBG=1" ...   then   (GPU box) bash tools/exp/bc_dw_variants.sh
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== product"; python3 $R/tools/bench_bc.py 512 10 dw
for v in $(ls $R/tools/exp/_build/libttk_bcd*.so 2>/dev/null); do
  echo "== $(basename $v)"; TTK_LIB=$v python3 $R/tools/bench_bc.py 512 10 dw
done
