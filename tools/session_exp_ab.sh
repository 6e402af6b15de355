# A/B of two engine builds on one box: gpurun_exp/lib_base.so against the tree's library, alternating, cfg3 then cfg2
for rep in 1 2; do
  for v in base tree; do
    if [ $v = base ]; then export PHYLY_AMD_LIB=gpurun_exp/lib_base.so; else unset PHYLY_AMD_LIB; fi
    timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --deriv-steps 0 2>/dev/null | grep '^{"metric' > gpurun_out/ab_${v}_cfg3_$rep.json || exit 1
    timeout -k 10 120 python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline --deriv-steps 0 2>/dev/null | grep '^{"metric' > gpurun_out/ab_${v}_cfg2_$rep.json || exit 1
  done
done
