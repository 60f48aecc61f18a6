#!/bin/bash
# same-box A/B of per-call kernel times: bash tools/exp/ab_percall.sh <kernel substring> <variant lib or "base"> ...   (two alternating rounds)
cd "$(dirname "$0")/../.."
pat=$1; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset TTK_LIB; else export TTK_LIB=tools/exp/_build/libttk_$v.so; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-copy-probe --per-call 2> /tmp/pc_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'crops/s', round(d['value']), end='  ')"
    grep "$pat" /tmp/pc_$v.txt | awk '{for(i=1;i<=NF;i++) if($i=="us") s+=$(i-1)} END {printf "sum(%s) = %.1f us\n", "'"$pat"'", s}'
  done
done
