#!/bin/bash
TAG=${1:-r03u}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --workload sndenv --steps 200 --warmup 20 --no-cpu-baseline --launch eager --streams 1 --min-seconds 0.02 > $OUT/stats.log 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cut -c1-200 $f | head -12
cp $f $ROOT/gpurun_out/${TAG}_sndenv_kernel_stats.csv
