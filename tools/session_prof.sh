set -o pipefail
mkdir -p gpurun_out/r03
tools/profile_r03.sh 3 || exit 1
for s in 5000000 2500000 1250000; do
  python3 bench.py --sites $s --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 2>/dev/null | grep '^{"metric' > gpurun_out/r03/proxy_$s.json || exit 1
  python3 bench.py --sites $s --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --dist 2>/dev/null | grep '^{"metric' > gpurun_out/r03/proxy_${s}_dist_world1.json || exit 1
  python3 bench.py --sites $s --steps 20 --warmup 3 --no-cpu-baseline --deriv-steps 0 --dist --sync-allreduce 2>/dev/null | grep '^{"metric' > gpurun_out/r03/proxy_${s}_dist_world1_sync.json || exit 1
done
for c in 2 4 5; do tools/profile_r03.sh $c || exit 1; done
