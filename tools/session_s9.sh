set -o pipefail
tools/step.sh s9_tests --timeout 400 -- python -m pytest tests/test_gpu_fused_asm.py tests/test_gpu_ll.py -x -q || exit 1
for v in 1 5; do tools/step.sh s9_b10M_v$v --timeout 200 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --engine-option 6=$v || exit 1; done
tools/step.sh s9_b1p25M --timeout 200 -- python bench.py --steps 20 --warmup 3 --sites 1250000 --no-cpu-baseline --deriv-steps 0 || exit 1
tools/step.sh s9_cfg2 --timeout 200 -- python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 || exit 1
PHYLY_AMD_LIB=$PWD/gpurun_exp/lib_EMPTY.so tools/step.sh s9_EMPTY --timeout 200 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 || exit 1
KREGEX='k_ll' tools/sq_passes.sh s9_ll3 4000000 \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY"
