#!/bin/bash
# Power / clock telemetry (rocm-smi, read-only) while one kernel family loops: is the chip power-limited under the three-product fp16 GEMMs?
#   bash tools/exp/power_probe.sh   -> gpurun_out/power/*.txt
O=gpurun_out/power; mkdir -p $O
sample() { # name, command...
  n=$1; shift
  "$@" > $O/$n.run.txt 2>&1 &
  pid=$!
  sleep 6   # (python + torch start-up)
  : > $O/$n.smi.txt
  while kill -0 $pid 2>/dev/null; do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use" >> $O/$n.smi.txt; echo --- >> $O/$n.smi.txt; sleep 0.3; done
  wait $pid
  echo "== $n"; python3 - $O/$n.smi.txt <<'P'
import re, sys
t = open(sys.argv[1]).read().split("---")
rows = []
for b in t:
    p, c, u = re.search(r"Power \(W\): ([\d.]+)", b), re.search(r"sclk clock level: \d+: \((\d+)Mhz", b), re.search(r"GPU use \(%\): (\d+)", b)
    if p and c and u and int(u.group(1)) >= 99: rows.append((float(p.group(1)), int(c.group(1))))
if rows:
    ps, cs = sorted(r[0] for r in rows), sorted(r[1] for r in rows)
    print(f"   {len(rows)} busy samples: power median {ps[len(ps) // 2]:.0f} W (min {ps[0]:.0f}, max {ps[-1]:.0f}); sclk median {cs[len(cs) // 2]} MHz (min {cs[0]}, max {cs[-1]})")
P
}
sample idle sleep 8
LAYERS=dw5_x KINDS=fwd sample gemm_fwd_512 python tools/bench_gemm.py 512 200000
LAYERS=dw5_x KINDS=wgrad sample gemm_wgrad_512 python tools/bench_gemm.py 512 150000
LAYERS=dw5_x KINDS=dgrad sample gemm_dgrad_512 python tools/bench_gemm.py 512 150000
sample step python bench.py --no-legs --no-cpu-baseline --no-kernel-timing --steps 800 --warmup 5
sample step_bc python bench.py --precision bf16-compute --no-cpu-baseline --no-kernel-timing --steps 1000 --warmup 5
