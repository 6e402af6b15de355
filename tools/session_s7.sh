set -o pipefail
for n in A1; do
  PHYLY_AMD_LIB=$PWD/gpurun_exp/lib_$n.so tools/step.sh s7_$n --timeout 200 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 || exit 1
done
