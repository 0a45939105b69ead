# quick A/B of library options: tools/opt_ab.sh "--opt name=value ..." ["..."]...   (run on the GPU box)
export PYTHONUNBUFFERED=1
for cfg in "$@"; do
  echo "== $cfg" | tee -a gpurun_out/opt_ab.log
  timeout -k 10 120 python bench.py --no-cpu-baseline $cfg 2>&1 | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_launch'].items()})" | tee -a gpurun_out/opt_ab.log
done
