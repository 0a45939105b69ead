# A/B of the look-ahead variants (run on the GPU box)
export PYTHONUNBUFFERED=1
for a in 0 2 1; do
  echo "GB25_AB2_AHEAD=$a" | tee -a gpurun_out/v5.log
  GB25_AB2_AHEAD=$a timeout -k 10 120 python bench.py --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_launch'])" | tee -a gpurun_out/v5.log
done
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decomposition.py tests/test_gpu_multiprocess.py -q -m gpu -x 2>&1 | tail -5
