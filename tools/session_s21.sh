set -o pipefail
tools/step.sh s21_tests --timeout 900 -- python -m pytest tests/test_gpu_ll.py tests/test_gpu_golden.py tests/test_gpu_differential.py tests/test_gpu_fullsize.py -x -q || exit 1
for o in 1 0; do python3 bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --deriv-steps 0 --engine-option 6=$o 2>/dev/null | grep '^{"metric' > gpurun_out/s21_cfg4_o$o.json || exit 1; done
