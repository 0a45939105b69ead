# throughput of BASELINE.json configs[0] and configs[1] (launch-latency-bound grids), run on the GPU box
# usage: small_configs.sh ["--opt name=value ..."]...
export PYTHONUNBUFFERED=1
[ $# -eq 0 ] && set -- ""
for cfg in "$@"; do
for size in "128 64 8" "360 180 24"; do
  echo "size=$size $cfg" | tee -a gpurun_out/small.log
  timeout -k 10 120 python bench.py $cfg --size $size --dt 600 --steps 400 --warmup 40 --no-profile --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value']), 'steps/s', round(d['ms_per_step'], 4), 'ms/step')" | tee -a gpurun_out/small.log
done
done
