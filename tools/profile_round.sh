#!/bin/bash
# Regenerates the round's measurement artefacts on a GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <out dir under gpurun_out>
# rocprofv3 kernel statistics with the driver's command for the four legs (fp32 B = 512 / 256, resnet18, bf16-compute), the two PMC passes (FETCH_SIZE / WRITE_SIZE,
# separate runs, kernel trace only), and the unprofiled bench lines.  The program itself follows `--` (no wrappers).
set -u
OUT=${1:-gpurun_out/prof}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_mn -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-legs > $R/$OUT/bench_B512_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_256 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --batch 256 > $R/$OUT/bench_B256_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_rn -- python3 $R/bench.py --gpus 1 --steps 10 --warmup 3 --backbone resnet18 > $R/$OUT/bench_resnet18_B512_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_bc -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --precision bf16-compute > $R/$OUT/bench_B512_bf16_compute_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_f -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing --no-legs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_w -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing --no-legs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/$OUT/pmc_l2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing --no-legs > /dev/null 2>&1
BCARGS="--precision bf16-compute --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_bf -- python3 $R/bench.py $BCARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_bw -- python3 $R/bench.py $BCARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/$OUT/pmc_bl2 -- python3 $R/bench.py $BCARGS > /dev/null 2>&1
cd $R
for leg in mn:B512 256:B256 rn:resnet18_B512 bc:B512_bf16_compute; do
  d=${leg%%:*}; n=${leg##*:}
  python tools/rocpd_stats.py $(find $OUT/prof_$d -name "*.db" | head -1) $OUT/bench_${n}_kernel_stats.csv > /dev/null
done
python tools/pmc_summary.py $(find $OUT/pmc_f -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_w -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json $(find $OUT/pmc_l2 -name "*counter_collection.csv" | head -1) > /dev/null
python tools/pmc_summary.py $(find $OUT/pmc_bf -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_bw -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic_bf16_compute.json $(find $OUT/pmc_bl2 -name "*counter_collection.csv" | head -1) > /dev/null
rm -rf $OUT/pmc_bf $OUT/pmc_bw $OUT/pmc_bl2
rm -rf $OUT/prof_mn $OUT/prof_256 $OUT/prof_rn $OUT/prof_bc $OUT/pmc_f $OUT/pmc_w $OUT/pmc_l2
# (the summary carries the csrc_sha256 of this tree; bench.py also finds it by itself once it is copied to profiles/r0N_pmc_traffic.json)
python3 bench.py --steps 30 --warmup 5 --traffic-json $OUT/pmc_traffic.json > $OUT/bench_B512.json 2>/dev/null
python3 bench.py --batch 256 --steps 30 --warmup 5 > $OUT/bench_B256.json 2>/dev/null
python3 bench.py --backbone resnet18 --steps 20 --warmup 5 > $OUT/bench_resnet18_B512.json 2>/dev/null
python3 bench.py --precision bf16-compute --steps 30 --warmup 5 --traffic-json $OUT/pmc_traffic_bf16_compute.json > $OUT/bench_B512_bf16_compute.json 2>/dev/null
for f in bench_B512 bench_B256 bench_resnet18_B512 bench_B512_bf16_compute bench_B512_profiled; do cut -c1-170 $OUT/$f.json; echo; done
ls -la $OUT
