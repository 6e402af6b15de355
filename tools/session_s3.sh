set -o pipefail
tools/step.sh s3_tests --timeout 700 -- python -m pytest tests/test_gpu_fused_asm.py tests/test_gpu_ll.py tests/test_gpu_golden.py -x -q &&
tools/step.sh s3_bench10M --timeout 300 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 &&
tools/step.sh s3_bench10M_old --timeout 300 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --engine-option 6=0 &&
tools/step.sh s3_bench1p25M --timeout 200 -- python bench.py --steps 20 --warmup 3 --sites 1250000 --no-cpu-baseline --deriv-steps 0 &&
tools/step.sh s3_cfg2 --timeout 200 -- python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 &&
tools/step.sh s3_cfg2_old --timeout 200 -- python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --engine-option 6=0
