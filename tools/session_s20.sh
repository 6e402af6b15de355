set -o pipefail
tools/step.sh s20_tests --timeout 900 -- python -m pytest tests/test_gpu_ll.py tests/test_gpu_golden.py tests/test_gpu_expect.py tests/test_gpu_deriv_marginal.py tests/test_gpu_reference_cases.py -x -q || exit 1
for c in 5 3 2; do python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --deriv-steps 0 2>/dev/null | grep '^{"metric' > gpurun_out/s20_cfg$c.json || exit 1; done
