# usage: ab.sh nameA nameB  (libs in tools/exp/_build/libttk_<name>.so)
for i in 1 2 3; do for v in "$@"; do TTK_LIB=$PWD/tools/exp/_build/libttk_$v.so python bench.py --no-cpu-baseline --no-kernel-timing --no-graph --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3))"; done; done
