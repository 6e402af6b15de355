set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_deriv_marginal.py tests/test_gpu_differential.py tests/test_gpu_golden.py tests/test_gpu_reference_cases.py -x -q 2>&1 | tail -4 || exit 1
for rep in 1 2; do for v in 2 10; do
  ARBPLF_UP_NODES=$v timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/ip_cfg3_${v}_$rep.json || exit 1
done; done
ARBPLF_UP_NODES=2 timeout -k 10 200 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/ip_cfg2_2_1.json || exit 1
ARBPLF_UP_NODES=10 timeout -k 10 200 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/ip_cfg2_10_1.json || exit 1
