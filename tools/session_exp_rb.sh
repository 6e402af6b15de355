set -o pipefail
timeout -k 10 700 python -m pytest tests/test_gpu_deriv_marginal.py tests/test_gpu_differential.py tests/test_gpu_expect.py tests/test_gpu_golden.py -x -q 2>&1 | tail -4 || exit 1
for v in 0 1; do
  timeout -k 10 200 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/inl5_$v.json || exit 1
done
