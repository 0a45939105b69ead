#!/bin/bash
# One rocprofv3 SQ-counter pass of the bench with the current environment (GB25_* switches are inherited).
set -e
TAG=${1:-sq}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY -d $OUT -o sq -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $OUT -o sq2 -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench2.json 2> $OUT/err2.txt
ls $OUT
