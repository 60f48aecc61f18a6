#!/bin/bash
# A/B of experiment builds against the product on ONE box, interleaved twice (box-to-box spread of these micro-benchmarks is +-5 %):
#   bash tools/exp/ab_bc.sh <pw|dw|all> <grep pattern of the rows> libttk_<name>.so ...
R=${GRAFT_REPO_ROOT:-$PWD}
what=$1; pat=$2; shift 2
for rep in 1 2; do
  echo "== product (run $rep)"; python3 $R/tools/bench_bc.py 512 20 $what 2>&1 | grep -E "$pat"
  for v in "$@"; do
    echo "== $v (run $rep)"; TTK_LIB=$R/tools/exp/_build/$v python3 $R/tools/bench_bc.py 512 20 $what 2>&1 | grep -E "$pat"
  done
done
