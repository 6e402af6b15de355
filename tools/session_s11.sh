set -o pipefail
for v in 1 9 1 9; do tools/step.sh s11_b10M_v$v --timeout 200 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --engine-option 6=$v || exit 1; grep -o '"kernel_ms": [0-9.]*' gpurun_out/s11_b10M_v$v.log; done
