set -o pipefail
tools/step.sh s12_tests --timeout 600 -- python -m pytest tests/test_gpu_deriv_marginal.py tests/test_gpu_golden.py -x -q || exit 1
for c in 3 4 5 2; do tools/step.sh s12_q$c --timeout 300 -- python tools/time_queries.py --config $c || exit 1; done
