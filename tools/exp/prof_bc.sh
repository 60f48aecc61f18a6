R=${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/bcprof
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_bc -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --precision bf16-compute --no-cpu-baseline > $R/$OUT/bench_profiled.json 2>/dev/null
cd $R
python tools/rocpd_stats.py $(find $OUT/prof_bc -name "*.db" | head -1) $OUT/kernel_stats.csv > /dev/null
rm -rf $OUT/prof_bc
head -60 $OUT/kernel_stats.csv | cut -c1-200
