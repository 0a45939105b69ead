#!/bin/bash
# VALU instruction mix of the kernels (rocprofv3 PMC, one pass): adds / muls / fmas / transcendental / int
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sq_mix
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT -d $OUT -o mix -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
ls $OUT
