set -o pipefail
tools/step.sh s5_tests --timeout 400 -- python -m pytest tests/test_gpu_fused_asm.py -x -q || exit 1
tools/step.sh s5_base --timeout 200 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 || exit 1
for n in NOMAT NOLDS NOMATNOLDS; do
  PHYLY_AMD_LIB=$PWD/gpurun_exp/lib_$n.so tools/step.sh s5_$n --timeout 200 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 || exit 1
done
PHYLY_AMD_LIB=$PWD/gpurun_exp/lib_NOMAT.so tools/step.sh s5_NOMAT_old --timeout 200 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --engine-option 6=0
