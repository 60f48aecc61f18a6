#!/bin/bash
# third leg: the same rounding of the raw conv outputs AFTER taking the per-channel mean out (what storing y - pivot would do)
out=gpurun_out/soak6; mkdir -p $out
run() { name=$1; shift; timeout 300 python tools/soak.py "$@" > $out/$name.txt 2>&1; grep -E "step  (100|200|300|400)|step  599" $out/$name.txt | awk -v n=$name '{printf "%s %s %s | ", n, $2, $4} END {print ""}'; }
for s in 0 1 2; do
  run yc_all_s$s --seed $s --round yc:0-12
done
run y_all_s2 --seed 2 --round y:0-12
