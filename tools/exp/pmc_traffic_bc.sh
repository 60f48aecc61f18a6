set -u
OUT=gpurun_out/r05pmcbc
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
BCARGS="--precision bf16-compute --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_bf -- python3 $R/bench.py $BCARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_bw -- python3 $R/bench.py $BCARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/$OUT/pmc_bl2 -- python3 $R/bench.py $BCARGS > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $(find $OUT/pmc_bf -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_bw -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic_bf16_compute.json $(find $OUT/pmc_bl2 -name "*counter_collection.csv" | head -1)
rm -rf $OUT/pmc_bf $OUT/pmc_bw $OUT/pmc_bl2
python3 bench.py --precision bf16-compute --steps 30 --warmup 5 --traffic-json $OUT/pmc_traffic_bf16_compute.json > $OUT/bench_B512_bf16_compute.json 2>/dev/null
