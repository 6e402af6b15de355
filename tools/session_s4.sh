set -o pipefail
tools/step.sh s4_tests --timeout 400 -- python -m pytest tests/test_gpu_fused_asm.py -x -q &&
for v in 1 2 3 4; do tools/step.sh s4_b10M_v$v --timeout 200 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --engine-option 6=$v || exit 1; done &&
for v in 1 2; do tools/step.sh s4_b1p25M_v$v --timeout 200 -- python bench.py --steps 20 --warmup 3 --sites 1250000 --no-cpu-baseline --deriv-steps 0 --engine-option 6=$v || exit 1; done &&
for v in 1 2; do tools/step.sh s4_cfg2_v$v --timeout 200 -- python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --engine-option 6=$v || exit 1; done &&
KREGEX='k_ll' EXTRA="" tools/sq_passes.sh s4_ll3 4000000 \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES"
