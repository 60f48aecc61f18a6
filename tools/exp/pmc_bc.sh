#!/bin/bash
# Hardware counters of the bf16-compute kernels (csrc/bc_*.hip) inside the real step: two rocprofv3 --pmc passes (kernel trace only) over
#   python3 bench.py --precision bf16-compute --steps 2 --warmup 1 --no-graph ...
#   bash tools/exp/pmc_bc.sh <out dir under gpurun_out> [kernel-name prefix filter, default bc_]
OUT=${1:-gpurun_out/pmc_bc}
FILT=${2:-bc_}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--precision bf16-compute --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing --no-copy-probe"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p1 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p2 -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cd $R
python3 - $OUT $FILT <<'PY'
import csv, glob, sys, collections, re
out, filt = sys.argv[1], sys.argv[2]
for p in ("p1", "p2"):
    f = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(p, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("ttk::", "").replace("bc::", "")
        if not k.startswith(filt): continue
        agg[(k, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), c in sorted(agg.items()):
        print(f"{k[:44]:44s} grid {g:>8s} n={len(next(iter(c.values())))} " + "  ".join(f"{n.replace('SQ_','')}={sum(v)/len(v):.3g}" for n, v in sorted(c.items())))
PY
rm -rf $OUT/p1 $OUT/p2
