for rep in 1 2; do for o in 1 9; do
  timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --deriv-steps 0 --engine-option 6=$o 2>/dev/null | grep '^{"metric' > gpurun_out/wu_${o}_$rep.json || exit 1
done; done
