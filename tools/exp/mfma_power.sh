#!/bin/bash
# rocm-smi samples beside tools/exp/mfma_power.hip's loops -> gpurun_out/power/mfma_*.txt
O=gpurun_out/power; mkdir -p $O tools/exp/_build
[ -x tools/exp/_build/mfma_power ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_power.hip -o tools/exp/_build/mfma_power 2>/dev/null
for cfg in "0 1" "1 1" "0 2" "1 2"; do
  set -- $cfg
  n=mfma_s$1_w$2
  timeout 60 tools/exp/_build/mfma_power $1 8 $2 > $O/$n.run.txt 2>&1 &
  pid=$!
  sleep 2
  : > $O/$n.smi.txt
  while kill -0 $pid 2>/dev/null; do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use" >> $O/$n.smi.txt; echo --- >> $O/$n.smi.txt; sleep 0.3; done
  wait $pid
  echo "== shape $1 ($( [ $1 = 0 ] && echo 16x16x32 || echo 32x32x16 )), $2 wave(s) per SIMD: $(tail -n 3 $O/$n.run.txt | awk '{print $5}' | tr '\n' ' ') TFLOP/s"
  python3 - $O/$n.smi.txt <<'P'
import re, sys
rows = []
for b in open(sys.argv[1]).read().split("---"):
    p, c, u = re.search(r"Power \(W\): ([\d.]+)", b), re.search(r"sclk clock level: \d+: \((\d+)Mhz", b), re.search(r"GPU use \(%\): (\d+)", b)
    if p and c and u and int(u.group(1)) >= 99: rows.append((float(p.group(1)), int(c.group(1))))
if rows:
    ps, cs = sorted(r[0] for r in rows), sorted(r[1] for r in rows)
    print(f"   {len(rows)} busy samples: power median {ps[len(ps) // 2]:.0f} W; sclk median {cs[len(cs) // 2]} MHz (min {cs[0]}, max {cs[-1]})")
P
done
