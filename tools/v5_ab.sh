# A/B of the packed (v5) tendency kernels against the scalar ones (run on the GPU box)
export PYTHONUNBUFFERED=1
for cfg in "1 0" "5 0" "5 1" "5 3"; do
  set -- $cfg
  echo "TRACER_V3=$1 MOMENTUM_V5=$2" | tee -a gpurun_out/v5.log
  GB25_TRACER_V3=$1 GB25_MOMENTUM_V5=$2 timeout -k 10 120 python bench.py --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" | tee -a gpurun_out/v5.log
done
GB25_TRACER_V3=5 GB25_MOMENTUM_V5=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "config1 or protocol or ragged or lookahead or minimum or phase or smooth or momentum" 2>&1 | tail -5
