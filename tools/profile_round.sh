#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench command, HBM traffic (FETCH_SIZE and WRITE_SIZE in
# separate passes) and SQ issue counters.  Run on the GPU box:  gpurun -- 'bash tools/profile_round.sh r01b'
set -e
TAG=${1:-r01b}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "[1/4] kernel stats"; date
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $REPO/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "[2/4] FETCH_SIZE"; date
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "[3/4] WRITE_SIZE"; date
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err
echo "[4/4] SQ counters"; date
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY -d $OUT/sq -o sq -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_sq.json 2> $OUT/sq.err
date; ls -R $OUT | head -40
