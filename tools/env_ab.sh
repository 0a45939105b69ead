# quick A/B of environment switches: tools/env_ab.sh "VAR=val ..." ["VAR=val ..."]...   (run on the GPU box)
export PYTHONUNBUFFERED=1
for cfg in "$@"; do
  echo "== $cfg" | tee -a gpurun_out/env_ab.log
  env $cfg timeout -k 10 120 python bench.py --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_launch'].items()})" | tee -a gpurun_out/env_ab.log
done
