#!/bin/bash
# second leg of tools/exp/soak_ab.sh: WHICH raw conv outputs (rounded to the bf16 grid in the fp32 path) carry the gap
out=gpurun_out/soak6; mkdir -p $out
run() { name=$1; shift; timeout 300 python tools/soak.py "$@" > $out/$name.txt 2>&1; grep -E "step  (100|200|300|400)|step  599" $out/$name.txt | awk -v n=$name '{printf "%s %s %s | ", n, $2, $4} END {print ""}'; }
for s in 0 1; do
  run y_0_2_s$s --seed $s --round y:0-2
  run y_3_7_s$s --seed $s --round y:3-7
  run y_8_12_s$s --seed $s --round y:8-12
done
