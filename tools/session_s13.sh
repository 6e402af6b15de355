set -o pipefail
tools/step.sh s13_tests --timeout 700 -- python -m pytest tests/test_gpu_ll.py tests/test_gpu_deriv_marginal.py tests/test_gpu_golden.py -x -q || exit 1
tools/step.sh s13_cfg4 --timeout 300 -- python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline || exit 1
tools/step.sh s13_cfg4_old --timeout 300 -- python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --engine-option 7=0 || exit 1
tools/step.sh s13_q4 --timeout 300 -- python tools/time_queries.py --config 4 || exit 1
