#!/bin/bash
# after tools/profile_round.sh TAG on the GPU box: rocpd databases -> the per-kernel CSV / JSON summaries kept under profiles/
set -e
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
db() { find $OUT/$1 -name "*.db" | head -1; }
python3 $REPO/tools/rocpd_summary.py stats "$(db stats)" $OUT/${TAG}_kernel_stats_1440x720x48.csv
python3 $REPO/tools/rocpd_summary.py counters "$(db fetch)" $OUT/${TAG}_pmc_fetch_size.csv
python3 $REPO/tools/rocpd_summary.py counters "$(db write)" $OUT/${TAG}_pmc_write_size.csv
python3 $REPO/tools/rocpd_summary.py counters "$(db sq)" $OUT/${TAG}_pmc_sq.csv
python3 $REPO/tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch_size.csv $OUT/${TAG}_pmc_write_size.csv $OUT/${TAG}_hbm_traffic_1440x720x48.json 1440 720 48
cp $OUT/bench_under_rocprof.json $OUT/${TAG}_bench_under_rocprof.json
head -12 $OUT/${TAG}_kernel_stats_1440x720x48.csv
