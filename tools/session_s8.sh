set -o pipefail
tools/step.sh s8_tests --timeout 400 -- python -m pytest tests/test_gpu_fused_asm.py -x -q || exit 1
for v in 1 5 2; do tools/step.sh s8_b10M_v$v --timeout 200 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --engine-option 6=$v || exit 1; done
for v in 1 5; do tools/step.sh s8_b1p25M_v$v --timeout 200 -- python bench.py --steps 20 --warmup 3 --sites 1250000 --no-cpu-baseline --deriv-steps 0 --engine-option 6=$v || exit 1; done
for v in 1 5; do tools/step.sh s8_cfg2_v$v --timeout 200 -- python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --engine-option 6=$v || exit 1; done
