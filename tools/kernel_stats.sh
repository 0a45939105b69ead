#!/bin/bash
# rocprofv3 kernel statistics of one bench.py command line, summarised into gpurun_out/<TAG>_kernel_stats.csv.
#   gpurun -- 'bash tools/kernel_stats.sh r04_catke --closure catke --steps 30 --warmup 3'
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $REPO/bench.py --no-cpu-baseline --no-profile "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 $REPO/tools/rocpd_summary.py stats "$(find $OUT/stats -name '*.db' | head -1)" $REPO/gpurun_out/${TAG}_kernel_stats.csv
cp $OUT/bench.json $REPO/gpurun_out/${TAG}_bench_under_rocprof.json
head -${LINES_SHOWN:-24} $REPO/gpurun_out/${TAG}_kernel_stats.csv
