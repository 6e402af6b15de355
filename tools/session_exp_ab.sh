# A/B of engine builds on one box: every gpurun_exp/lib_*.so against the tree's library, alternating.
# AB_ARGS: bench.py arguments (default: the ll leg of config 3)
args=${AB_ARGS:---steps 30 --warmup 5 --no-cpu-baseline --deriv-steps 0}
for rep in 1 2; do
  for lib in tree gpurun_exp/lib_*.so; do
    v=$(basename $lib .so)
    if [ $lib = tree ]; then unset PHYLY_AMD_LIB; else export PHYLY_AMD_LIB=$lib; fi
    timeout -k 10 200 python bench.py $args 2>/dev/null | grep '^{"metric' > gpurun_out/ab_${v}_$rep.json || exit 1
  done
done
