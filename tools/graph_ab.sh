# A/B of HIP-graph replay against eager launches on the launch-bound configurations (run on the GPU box)
export PYTHONUNBUFFERED=1
for size in "128 64 8" "360 180 24"; do
  for ts in 1 0; do
  for g in 0 1; do
    echo "size=$size GB25_TWO_STREAMS=$ts GB25_GRAPH=$g" | tee -a gpurun_out/graph_ab.log
    GB25_TWO_STREAMS=$ts GB25_GRAPH=$g timeout -k 10 120 python bench.py --size $size --dt 600 --steps 200 --warmup 20 --no-profile --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')" | tee -a gpurun_out/graph_ab.log
  done
  done
done
