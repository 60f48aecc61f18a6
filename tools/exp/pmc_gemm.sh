#!/bin/bash
# Hardware counters of the pointwise GEMM kernels on the layer shapes (tools/bench_gemm.py), one rocprofv3 --pmc pass per counter set;
#   bash tools/exp/pmc_gemm.sh <out dir>     (set TTK_GEMM_R=0 for the 128x256 kernels)
OUT=${1:-gpurun_out/pmc_gemm}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p1 -- python3 $R/tools/bench_gemm.py 512 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p2 -- python3 $R/tools/bench_gemm.py 512 3 > /dev/null 2>&1
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
for p in ("p1", "p2"):
    f = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(p, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("ttk::", "")
        if not k.startswith("pw16"): continue
        agg[(k, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), c in sorted(agg.items()):
        print(f"{k[:60]:60s} grid {g:>8s} " + "  ".join(f"{n}={sum(v)/len(v):.3g}" for n, v in sorted(c.items())))
PY
