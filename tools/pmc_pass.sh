#!/bin/bash
# One rocprofv3 counter pass (kernel trace + the given counters, nothing else) of one bench.py command line, summarised per kernel:
#   gpurun -- 'bash tools/pmc_pass.sh TAG "SQ_INSTS_VALU SQ_WAVE_CYCLES ..." --data-free --size 1440 720 60'
set -e
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pass -o pass -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 $REPO/tools/rocpd_summary.py counters "$(find $OUT/pass -name '*.db' | head -1)" $REPO/gpurun_out/${TAG}_pmc.csv
