# two bench runs with the current build/environment (run on the GPU box); prints steps/s, ms/step and per-kernel ms
export PYTHONUNBUFFERED=1
for r in 1 2; do
  timeout -k 10 120 python bench.py --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_launch'].items()})" | tee -a gpurun_out/quick.log
done
