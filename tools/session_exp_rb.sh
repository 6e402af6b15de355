set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_deriv_marginal.py tests/test_gpu_differential.py -x -q 2>&1 | tail -4 || exit 1
python3 tools/time_queries.py --config 4 2>/dev/null | grep '^{' > gpurun_out/q4.json || exit 1
cat gpurun_out/q4.json
