R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2 3; do
  for v in product "$@"; do
    lib=$R/neuralnet-tracker-traincode_amd/libttk_hip.so; [ $v != product ] && lib=$R/tools/exp/_build/$v
    echo -n "$v: "; TTK_LIB=$lib python3 $R/bench.py --precision bf16-compute --no-cpu-baseline --no-copy-probe --steps 40 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
  done
done
