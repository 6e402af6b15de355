set -o pipefail
tools/step.sh s16_tests --timeout 900 -- python -m pytest tests/test_gpu_deriv_marginal.py tests/test_gpu_golden.py tests/test_gpu_differential.py tests/test_gpu_reference_cases.py tests/test_gpu_fit.py tests/test_gpu_shard.py -x -q || exit 1
tools/step.sh s16_q3 --timeout 300 -- python tools/time_queries.py --config 3 || exit 1
tools/step.sh s16_q2 --timeout 300 -- python tools/time_queries.py --config 2 || exit 1
