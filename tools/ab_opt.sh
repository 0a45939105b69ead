#!/bin/bash
# A/B of one library option on the same box: bash tools/ab_opt.sh NAME V0 V1 [extra bench.py args]; alternating runs, 60 timed steps each
export PYTHONUNBUFFERED=1
NAME=$1; A=$2; B=$3; shift 3
for r in 1 2; do
  for v in $A $B; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 60 --warmup 10 --opt $NAME=$v "$@" 2>&1 | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$NAME=$v', round(d['value'],1), 'steps/s', round(d['ms_per_step'],3), 'ms', {k: round(x,3) for k,x in d['kernels_ms_per_launch'].items()})" | tee -a gpurun_out/ab_$NAME.log
  done
done
