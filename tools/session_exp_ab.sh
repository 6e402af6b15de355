# A/B of engine builds on one box: every gpurun_exp/lib_*.so against the tree's library, alternating; ll leg of cfg3
timeout -k 10 300 python -m pytest tests/test_gpu_fused_asm.py -x -q 2>&1 | tail -2 || exit 1
for rep in 1 2; do
  for lib in tree gpurun_exp/lib_*.so; do
    v=$(basename $lib .so)
    if [ $lib = tree ]; then unset PHYLY_AMD_LIB; else export PHYLY_AMD_LIB=$lib; fi
    timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --deriv-steps 0 2>/dev/null | grep '^{"metric' > gpurun_out/ab_${v}_cfg3_$rep.json || exit 1
  done
done
