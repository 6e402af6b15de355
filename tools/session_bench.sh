# bench lines of the four configs (with the CPU baseline), after the PMC traffic files of the tree are in profiles/
set -o pipefail
mkdir -p gpurun_out/r03
for c in 3 2 4 5; do
  python3 bench.py --config $c --steps 20 --warmup 3 > gpurun_out/r03/bench_cfg$c.log 2>&1 || { tail -5 gpurun_out/r03/bench_cfg$c.log; exit 1; }
  grep '^{"metric' gpurun_out/r03/bench_cfg$c.log > gpurun_out/r03/bench_cfg$c.json
done
