#!/bin/bash
# same-box A/B of whole-step rates: bash tools/exp/ab_bench.sh <out dir> <variant> [<variant> ...]   (variant "base" = the product library)
O=$1; shift; mkdir -p $O
for rep in 1 2; do
  for v in "$@"; do
    lib=$PWD/neuralnet-tracker-traincode_amd/libttk_hip.so; [ $v != base ] && lib=$PWD/tools/exp/_build/libttk_$v.so
    TTK_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_ms_per_step']
print('$v', round(d['value']), 'crops/s', d['ms_per_step'], 'ms', {n: k[n] for n in k if n.startswith('dw_') or n.startswith('pw16m')})" >> $O/ab.txt
  done
done
cat $O/ab.txt
