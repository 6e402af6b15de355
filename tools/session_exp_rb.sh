set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_deriv_marginal.py tests/test_gpu_differential.py -x -q 2>&1 | tail -3 || exit 1
python3 tools/time_queries.py --config 5 2>/dev/null | grep '^{' | cut -c1-300
python3 tools/time_queries.py --config 5 2>/dev/null | grep '^{' | cut -c1-300
