set -o pipefail
tools/step.sh full_gpu_tests --timeout 1000 -- python -m pytest tests/ -x -q -m gpu --durations=25
