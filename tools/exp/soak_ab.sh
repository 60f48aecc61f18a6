#!/bin/bash
# Round 6, VERDICT item 8: where the bf16-compute path's soak gap comes from.  Legs on ONE box: fp32 and bf16-compute with three weight initialisations
# each (run-to-run spread), and the fp32 path with tensors rounded to the bf16 grid after their kernels (tools/soak.py --round).
out=gpurun_out/soak6; mkdir -p $out
run() { name=$1; shift; timeout 300 python tools/soak.py "$@" > $out/$name.txt 2>&1; grep -E "step  (100|200|300|400)|step  599" $out/$name.txt | awk -v n=$name '{printf "%s %s %s | ", n, $2, $4} END {print ""}'; }
for s in 0 1 2; do run fp32_s$s --seed $s; run bc_s$s --precision bf16-compute --seed $s; done
for s in 0 1; do
  run g_early_s$s --seed $s --round g:0-2
  run g_all_s$s --seed $s --round g:0-13
  run g_late_s$s --seed $s --round g:3-13
  run y_all_s$s --seed $s --round y:0-12
  run gy_all_s$s --seed $s --round g:0-13,y:0-12
done
